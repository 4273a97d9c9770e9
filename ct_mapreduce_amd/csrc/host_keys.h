// host_keys.h — members whose serial is longer than CTMR_MAX_SERIAL live in host-side sets per rank; a group round settles
// them between the ranks (engine/group.inc round_finish).  The pure parts — the wire form of a rank's list of additions and
// the verdict every rank derives from the gathered lists — are here, free of HIP and of the engine, so that the tests can
// run them on the CPU (tests/harness/host_keys_harness.cpp).
//
// What it stands for in the reference: ONE Redis set answers SADD for every ct-fetch process (storage/rediscache.go:57-65) —
// a member some rank held before the round is known to everybody; among the ranks that met a new member in the same round
// the lowest log index keeps WasUnknown.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

namespace ctmr {

struct HostKeyCand {
  uint64_t order;    // the entry's place in the round's log order
  uint32_t rank;     // the rank that added it
  int32_t exp_hour;  // the set: (expiry hour, canonical issuer index)
  uint32_t canon;
  std::string member;  // raw serial octets
  size_t at;           // index in that rank's own list of additions
};

// wire form, u64 words: order | exp_hour (low 32) + canon (high 32) | octets | the octets, zero padded to 8
inline void host_keys_append(std::vector<uint64_t>& out, uint64_t order, int32_t exp_hour, uint32_t canon, const std::string& member) {
  out.push_back(order);
  out.push_back((uint64_t)(uint32_t)exp_hour | (uint64_t)canon << 32);
  out.push_back(member.size());
  const size_t at = out.size();
  out.resize(at + (member.size() + 7) / 8, 0);
  if (!member.empty()) memcpy(&out[at], member.data(), member.size());
}

// one rank's list; false = malformed (a length that runs past the list, or an absurd one)
inline bool host_keys_parse(const uint64_t* p, uint64_t n_words, uint32_t rank, std::vector<HostKeyCand>& out) {
  size_t at = 0;
  uint64_t k = 0;
  while (k < n_words) {
    if (k + 3 > n_words) return false;
    const uint64_t len = p[k + 2], lw = (len + 7) / 8;
    if (len > (1u << 20) || k + 3 + lw > n_words) return false;
    out.push_back({p[k], rank, (int32_t)(uint32_t)p[k + 1], (uint32_t)(p[k + 1] >> 32), std::string((const char*)(p + k + 3), (size_t)len), at++});
    k += 3 + lw;
  }
  return true;
}

// held[c] != 0: some rank has held candidate c's member since BEFORE this round.  lost[c] = 1: candidate c does not keep
// WasUnknown and leaves its rank's set again.  Deterministic in its inputs: every rank computes the same verdict.
inline void host_keys_verdict(const std::vector<HostKeyCand>& cands, const std::vector<uint64_t>& held, std::vector<uint8_t>& lost) {
  const uint32_t nc = (uint32_t)cands.size();
  std::vector<uint32_t> by(nc);
  for (uint32_t c = 0; c < nc; c++) by[c] = c;
  auto same = [&](const HostKeyCand& a, const HostKeyCand& b) { return a.exp_hour == b.exp_hour && a.canon == b.canon && a.member == b.member; };
  std::sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) {
    const HostKeyCand &x = cands[a], &y = cands[b];
    if (x.exp_hour != y.exp_hour) return x.exp_hour < y.exp_hour;
    if (x.canon != y.canon) return x.canon < y.canon;
    if (x.member != y.member) return x.member < y.member;
    if (x.order != y.order) return x.order < y.order;
    return x.rank < y.rank;
  });
  lost.assign(nc, 0);
  for (uint32_t a = 0; a < nc;) {
    uint32_t b = a;
    bool before = false;
    for (; b < nc && same(cands[by[a]], cands[by[b]]); b++)
      if (held[by[b]]) before = true;
    for (uint32_t c = a; c < b; c++) lost[by[c]] = before || c != a;  // by[a] is the lowest (order, rank) of its member
    a = b;
  }
}

}  // namespace ctmr
