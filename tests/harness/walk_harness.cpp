// Test-only host build of the PRODUCT's device walk (ct_mapreduce_amd/csrc/der_walk.h), so that
// it can be fuzzed against the oracle on machines without a GPU.  Never linked into libctmr.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../ct_mapreduce_amd/csrc/der_walk.h"

struct PaddedReader {
  const uint8_t* p;
  uint32_t ld4(uint32_t pos) const {
    uint32_t v;
    memcpy(&v, p + pos, 4);
    return v;
  }
};

struct HarnessOut {
  int32_t ok;
  uint32_t serial_off, serial_len;
  int64_t not_before, not_after;
  uint32_t cn_off, cn_len;
  int32_t bc_valid, is_ca;
  uint32_t spki_off, spki_len;
};

extern "C" void harness_walk(const uint8_t* der, uint32_t len, uint8_t fill, HarnessOut* out) {
  // bytes past the certificate are garbage the walk must never depend on
  std::vector<uint8_t> buf((size_t)len + 64, fill);
  memcpy(buf.data(), der, len);
  PaddedReader r{buf.data()};
  ctmr::Walk w;
  const bool ok = ctmr::walk_cert(r, len, w);
  memset(out, 0, sizeof *out);
  out->ok = ok;
  if (!ok) return;
  out->serial_off = w.serial_off; out->serial_len = w.serial_len;
  out->not_before = w.not_before; out->not_after = w.not_after;
  out->cn_off = w.cn_off; out->cn_len = w.cn_len;
  out->bc_valid = w.bc_valid; out->is_ca = w.is_ca;
  out->spki_off = w.spki_off; out->spki_len = w.spki_len;
}
