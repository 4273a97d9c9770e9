// Host build of csrc/host_keys.h (the long-serial settlement of a group round) for the CPU tests.
#include "../../ct_mapreduce_amd/csrc/host_keys.h"

// Every rank's list arrives in wire form: words[] holds the lists one after the other, n_words[r] words for rank r.
// held[c] comes from the caller (what the all-reduce would deliver), candidates numbered rank by rank in list order.
// Returns the number of candidates (lost[c] filled), or -1 - r when rank r's list is malformed.
extern "C" int harness_host_keys_verdict(const uint64_t* words, const uint64_t* n_words, uint32_t world, const uint64_t* held,
                                         uint8_t* lost, uint32_t cap) {
  std::vector<ctmr::HostKeyCand> cands;
  uint64_t at = 0;
  for (uint32_t r = 0; r < world; r++) {
    if (!ctmr::host_keys_parse(words + at, n_words[r], r, cands)) return -1 - (int)r;
    at += n_words[r];
  }
  if (cands.size() > cap) return -1000;
  std::vector<uint64_t> h(held, held + cands.size());
  std::vector<uint8_t> l;
  ctmr::host_keys_verdict(cands, h, l);
  for (size_t c = 0; c < l.size(); c++) lost[c] = l[c];
  return (int)cands.size();
}

// the wire form of one member (for the round trip and for building malformed lists in the test)
extern "C" uint32_t harness_host_keys_append(uint64_t order, int32_t exp_hour, uint32_t canon, const uint8_t* member, uint32_t len,
                                             uint64_t* out, uint32_t cap) {
  std::vector<uint64_t> v;
  ctmr::host_keys_append(v, order, exp_hour, canon, std::string((const char*)member, len));
  if (v.size() > cap) return 0;
  for (size_t k = 0; k < v.size(); k++) out[k] = v[k];
  return (uint32_t)v.size();
}
