// Test-only host build of the PRODUCT's get-entries decoder (ct_mapreduce_amd/csrc/entry_decode.h), so that it
// can be fuzzed against the oracle on machines without a GPU.  Never linked into libctmr.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../ct_mapreduce_amd/csrc/entry_decode.h"

struct EntryOut {
  int32_t ok, entry_type;
  uint64_t timestamp, cert_lo, cert_hi, chain0_lo;
  uint32_t chain0_len, n_chain;
  uint64_t tbs_lo;
  uint32_t tbs_len;
};

extern "C" void harness_decode_entry(const uint8_t* blob, uint64_t blob_len, uint64_t l0, uint64_t l1, uint64_t x1,
                                     uint8_t fill, EntryOut* out) {
  // bytes past the blob are garbage the decoder must never depend on
  std::vector<uint8_t> buf((size_t)blob_len + 64, fill);
  memcpy(buf.data(), blob, blob_len);
  ctmr::HostBytes b{buf.data()};
  ctmr::EntryDec d;
  ctmr::decode_entry(b, l0, l1, x1, d);
  memset(out, 0, sizeof *out);
  out->ok = d.ok;
  if (!d.ok) return;
  out->entry_type = (int32_t)d.entry_type;
  out->timestamp = d.timestamp;
  out->cert_lo = d.cert_lo; out->cert_hi = d.cert_hi;
  out->chain0_lo = d.chain0_lo; out->chain0_len = d.chain0_len; out->n_chain = d.n_chain;
  out->tbs_lo = d.tbs_lo; out->tbs_len = d.tbs_len;
}

extern "C" uint64_t harness_quick_hash(const uint8_t* der, uint32_t len) {
  return ctmr::cert_quick_hash(ctmr::HostBytes{der}, 0, len);
}
