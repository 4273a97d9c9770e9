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
  uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  ctmr::RawCert raw() const { return ctmr::RawCert{(const uint32_t*)p, 0}; }  // the out-of-line key checks, as on the device
  void touch(uint32_t, uint32_t) const {}
  void touch_tail(uint32_t, uint32_t) const {}
};

struct HarnessOut {
  int32_t ok;
  uint32_t serial_off, serial_len;
  int64_t not_before, not_after;
  uint32_t cn_off, cn_len;
  int32_t bc_valid, is_ca;
  uint32_t spki_off, spki_len;
  uint32_t serial_w[5];
  int32_t cn_match;
  int32_t nonfatal;
};

// spki: 1 = the walk also parses the public key (ctmr_set_strict_spki, the default), 0 = rounds 1-3 behaviour
static int g_spki = 1;
extern "C" void harness_set_spki(int on) { g_spki = on; }
// ext: 1 = strict_extensions (ctmr_set_strict_extensions; off by default)
static int g_ext = 0;
extern "C" void harness_set_ext(int on) { g_ext = on; }

// strings: 1 = strict_strings inside the walk (findings arrive as WALK_NF_STRING in HarnessOut.nonfatal)
static int g_strings = 0;
extern "C" void harness_set_strings(int on) { g_strings = on; }

extern "C" void harness_walk_f(const uint8_t* der, uint32_t len, uint8_t fill, const char* filter,
                               uint32_t flen, int use_filter, HarnessOut* out);

// Where the URIs of a cRLDistributionPoints value lie (der_walk.h crl_dps<COLLECT>: what kernels/meta*.h use):
// returns -1 = malformed, else the number of URIs; the first `cap` (offset, length) pairs go to out.
extern "C" int harness_crl_uris(const uint8_t* der, uint32_t len, uint32_t cv, uint32_t ev, uint32_t* out, uint32_t cap) {
  std::vector<uint8_t> buf((size_t)len + 64, 0xA5);
  memcpy(buf.data(), der, len);
  PaddedReader r{buf.data()};
  uint32_t uo[8] = {0}, ul[8] = {0}, nu = 0, nf = 0;
  bool ok = true;
  ctmr::crl_dps<true, false, false, 8u>(r, len, cv, ev, ok, uo, ul, nu, nf, false);
  if (!ok) return -1;
  for (uint32_t k = 0; k < nu && k < cap && k < 8; k++) { out[2 * k] = uo[k]; out[2 * k + 1] = ul[k]; }
  return (int)nu;
}

extern "C" void harness_walk(const uint8_t* der, uint32_t len, uint8_t fill, HarnessOut* out) {
  harness_walk_f(der, len, fill, nullptr, 0, 0, out);
}

extern "C" void harness_walk_f(const uint8_t* der, uint32_t len, uint8_t fill, const char* filter,
                               uint32_t flen, int use_filter, HarnessOut* out) {
  // strings.Split(filter, ",") laid out exactly as ctmr_set_filter does
  uint32_t piece_len[64], piece_word[64], words[1024] = {0}, np = 0, nw = 0;
  for (uint32_t s = 0;;) {
    uint32_t t = s;
    while (t < flen && filter[t] != ',') t++;
    piece_len[np] = t - s;
    piece_word[np] = nw;
    memcpy((uint8_t*)(words + nw), filter + s, t - s);
    nw += (t - s + 3) / 4;
    np++;
    if (t >= flen) break;
    s = t + 1;
  }
  ctmr::FilterView fv{np, piece_len, piece_word, words};
  // bytes past the certificate are garbage the walk must never depend on
  std::vector<uint8_t> buf((size_t)len + 64, fill);
  memcpy(buf.data(), der, len);
  PaddedReader r{buf.data()};
  ctmr::Walk w;
  const bool ok = ctmr::walk_cert(r, len, w, use_filter ? &fv : nullptr, g_spki != 0, g_strings != 0, g_ext != 0);
  memset(out, 0, sizeof *out);
  out->ok = ok;
  if (!ok) return;
  out->serial_off = w.serial_off; out->serial_len = w.serial_len;
  out->not_before = w.not_before; out->not_after = w.not_after;
  out->cn_off = w.cn_off; out->cn_len = w.cn_len;
  out->bc_valid = w.bc_valid; out->is_ca = w.is_ca;
  out->spki_off = w.spki_off; out->spki_len = w.spki_len;
  memcpy(out->serial_w, w.serial_w, 20);
  out->cn_match = w.cn_match;
  out->nonfatal = (int32_t)w.nonfatal;
}

// The walk over a BARE TBSCertificate (strict_leaf: what LogEntryFromLeaf parses of a precertificate entry's leaf)
extern "C" void harness_walk_tbs(const uint8_t* tbs, uint32_t len, uint8_t fill, HarnessOut* out) {
  std::vector<uint8_t> buf((size_t)len + 64, fill);
  memcpy(buf.data(), tbs, len);
  PaddedReader r{buf.data()};
  ctmr::Walk w;
  const bool ok = ctmr::walk_tbs(r, len, w, g_spki != 0, g_ext != 0);
  memset(out, 0, sizeof *out);
  out->ok = ok;
  if (!ok) return;
  out->serial_off = w.serial_off; out->serial_len = w.serial_len;
  out->not_before = w.not_before; out->not_after = w.not_after;
  out->cn_off = w.cn_off; out->cn_len = w.cn_len;
  out->bc_valid = w.bc_valid; out->is_ca = w.is_ca;
  out->spki_off = w.spki_off; out->spki_len = w.spki_len;
  memcpy(out->serial_w, w.serial_w, 20);
  out->cn_match = w.cn_match;
  out->nonfatal = (int32_t)w.nonfatal;
}

// strict_strings: the product's verdict on the character sets of the string values in both Names (der_walk.h
// walk_name → value_strings_ok) — 1 = fine, 0 = a finding, -1 = the certificate does not parse
extern "C" int harness_name_strings(const uint8_t* der, uint32_t len, uint8_t fill) {
  std::vector<uint8_t> buf((size_t)len + 64, fill);
  memcpy(buf.data(), der, len);
  PaddedReader r{buf.data()};
  ctmr::Walk w;
  if (!ctmr::walk_cert(r, len, w, nullptr, g_spki, true)) return -1;
  return (w.nonfatal & ctmr::WALK_NF_STRING) ? 0 : 1;
}

// What the walk READS: every ld4/ldg marks its four bytes.  Used by bench.py's "needed_bytes" accounting (the bytes and
// the distinct 128-byte HBM lines a certificate's walk covers when the certificate starts at byte `phase` of a line),
// against which the measured traffic of the map kernel is an over-fetch ratio.
struct CountingReader {
  const uint8_t* p;
  std::vector<uint8_t>* mark;
  uint32_t ld4(uint32_t pos) const {
    for (uint32_t k = 0; k < 4; k++)
      if (pos + k < mark->size()) (*mark)[pos + k] = 1;
    uint32_t v;
    memcpy(&v, p + pos, 4);
    return v;
  }
  uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  void touch(uint32_t, uint32_t) const {}
  void touch_tail(uint32_t, uint32_t) const {}
};

static int g_touched_strict = 0;  // harness_walk_touched under the reference profile (strict_strings + strict_extensions)
extern "C" void harness_touched_profile(int strict) { g_touched_strict = strict; }
extern "C" int harness_walk_touched(const uint8_t* der, uint32_t len, uint32_t phase, const char* filter, uint32_t flen,
                                    uint32_t* bytes, uint32_t* lines128) {
  uint32_t piece_len[64], piece_word[64], words[1024] = {0}, np = 0, nw = 0;
  for (uint32_t s = 0;;) {
    uint32_t t = s;
    while (t < flen && filter[t] != ',') t++;
    piece_len[np] = t - s;
    piece_word[np] = nw;
    memcpy((uint8_t*)(words + nw), filter + s, t - s);
    nw += (t - s + 3) / 4;
    np++;
    if (t >= flen) break;
    s = t + 1;
  }
  ctmr::FilterView fv{np, piece_len, piece_word, words};
  std::vector<uint8_t> buf((size_t)len + 64, 0);
  memcpy(buf.data(), der, len);
  std::vector<uint8_t> mark(len, 0);
  CountingReader r{buf.data(), &mark};
  ctmr::Walk w;
  const bool ok = ctmr::walk_cert(r, len, w, flen ? &fv : nullptr, true, g_touched_strict != 0, g_touched_strict != 0);
  uint32_t nb = 0, nl = 0;
  int64_t last_line = -1;
  for (uint32_t i = 0; i < len; i++)
    if (mark[i]) {
      nb++;
      const int64_t line = (int64_t)((phase + i) >> 7);
      if (line != last_line) { nl++; last_line = line; }
    }
  *bytes = nb;
  *lines128 = nl;
  return ok;
}

// The device's per-lane LDS window, simulated (round 6): same bookkeeping as kernels/readers.h WinReaderS — a window of
// `wbytes` bytes beginning `slack` bytes in front of the certificate, touch() refilling it from a 16-byte boundary when the
// hinted range is not inside, touch_tail() moving it to a dword boundary, reads outside it counted as misses — so that the
// walk's hints can be checked against a window geometry on the CPU: a per-lane refill on the GPU is sixteen uncoalesced
// loads and a round trip for the whole wave, a miss repeats the certificate with the exact reader.
template <bool HEAD>
struct WindowSimT {
  // HEAD: the first window begins behind the outer headers, which the walk reads from sixteen octets held apart (WinGeo::SKIP)
  static constexpr bool kHead = HEAD;
  uint32_t hd[4] = {0, 0, 0, 0};
  bool hd_ok = false;
  const uint8_t* p;
  uint32_t base_phase;  // (payload offset of the certificate) mod 16
  uint32_t wbytes;
  mutable int64_t grel;
  mutable uint32_t refills = 0, misses = 0, first_refill_pos = 0, first_miss_pos = 0;
  uint32_t size = 0xffffffffu;  // of the buffer behind p (set by the caller: damaged lengths ask for octets far outside)
  uint32_t at(uint32_t pos) const {
    uint32_t v = 0;
    if ((uint64_t)pos + 4u <= size) memcpy(&v, p + pos, 4);
    return v;
  }
  uint32_t ld4(uint32_t pos) const {
    const int64_t rel = (int64_t)pos - grel;
    if (rel < 0 || rel > (int64_t)wbytes - 4) {
      if (!misses) first_miss_pos = pos;
      misses++;
    }
    return at(pos);
  }
  uint32_t ld2c(uint32_t pos) const {
    const int64_t rel = (int64_t)pos - grel;
    if (rel < 0 || rel > (int64_t)wbytes - 2) {
      if (!misses) first_miss_pos = pos;
      misses++;
    }
    return at(pos) & 0xffffu;
  }
  uint32_t ldg(uint32_t pos) const { return at(pos); }  // the tail registers
  void touch(uint32_t pos, uint32_t need) const {
    if (need > wbytes - 16u) need = wbytes - 16u;
    const int64_t rel = (int64_t)pos - grel;
    if (rel < 0 || rel > (int64_t)(wbytes - need)) {
      if (!refills) first_refill_pos = pos;
      refills++;
      grel = (int64_t)pos - (int64_t)((base_phase + pos) & 15u);
    }
  }
  void touch_tail(uint32_t pos, uint32_t) const { grel = (int64_t)pos - (int64_t)((base_phase + pos) & 3u); }
  // the key tail registers (the RSA exponent at the end of the SubjectPublicKeyInfo arrives with touch_tail's burst)
  struct KeyTail { bool valid; uint32_t at; uint32_t w[4]; };
  KeyTail key_tail(uint32_t pos) const {
    KeyTail t{pos >= 12u && (uint64_t)pos + 4u <= size, pos - 12u, {0, 0, 0, 0}};
    if (t.valid) memcpy(t.w, p + pos - 12u, 16);
    return t;
  }
  // the wave-cooperative side of the device reader, for a wave of one lane: the subjectAltName walk is ext_san_coop's
  mutable uint32_t coop = 0, deferred = 0;
  static bool whole_wave() { return true; }
  static bool any_lane(bool x) { return x; }
  void coop_refill(uint32_t pos, bool want) const {
    if (want) { coop++; grel = (int64_t)pos - (int64_t)((base_phase + pos) & 15u); }
  }
  void coop_refill_lines(uint32_t pos, bool want) const {  // (base_phase is the offset within a 128-byte line here: see the caller)
    // (whole lines for windows that are a whole number of them, else 32-byte sectors: readers.h CTMR_SAN_ALIGN)
    if (want) { coop++; grel = (int64_t)pos - (int64_t)((base_phase + pos) & (wbytes % 128u == 0u ? 127u : 31u)); }
  }
  void coop_refill_lines_to(uint32_t pos, bool want, uint32_t) const { coop_refill_lines(pos, want); }
  bool holds(uint32_t pos, uint32_t need) const {
    const int64_t rel = (int64_t)pos - grel;
    return rel >= 0 && rel <= (int64_t)wbytes - (int64_t)need;
  }
  uint32_t ld2(uint32_t pos) const { return at(pos) & 0xffffu; }
  uint32_t wend() const { return (uint32_t)(grel + wbytes); }
  void defer_exact() const { deferred++; }
};
using WindowSim = WindowSimT<false>;
// returns ok; out = {per-lane refills, misses, position of the first refill, position of the first miss, cooperative
// refills of the subjectAltName walk, defer_exact calls}; phase = the certificate's offset within a 128-byte line
extern "C" int harness_walk_window(const uint8_t* der, uint32_t len, uint32_t phase, uint32_t wbytes, int strings, int ext,
                                   uint32_t* out) {
  std::vector<uint8_t> buf((size_t)len + 64, 0);
  memcpy(buf.data(), der, len);
  WindowSim r{{0, 0, 0, 0}, false, buf.data(), phase & 127u, wbytes, -(int64_t)(phase & 3u)};
  r.size = (uint32_t)buf.size();
  ctmr::Walk w;
  const bool ok = ctmr::walk_cert(r, len, w, nullptr, true, strings != 0, ext != 0);
  out[0] = r.refills; out[1] = r.misses; out[2] = r.first_refill_pos; out[3] = r.first_miss_pos; out[4] = r.coop; out[5] = r.deferred;
  return ok;
}
// … with the first window beginning `skip` octets into the certificate (dword-aligned down in the payload) and the outer
// headers read from the certificate's first sixteen octets (kernels/readers.h WinGeo::SKIP, der_walk.h HeadView)
extern "C" int harness_walk_window_skip(const uint8_t* der, uint32_t len, uint32_t phase, uint32_t wbytes, uint32_t skip, int strings,
                                        int ext, uint32_t* out) {
  std::vector<uint8_t> buf((size_t)len + 64, 0);
  memcpy(buf.data(), der, len);
  WindowSimT<true> r{{0, 0, 0, 0}, true, buf.data(), phase & 127u, wbytes, (int64_t)skip - (int64_t)((phase + skip) & 3u)};
  memcpy(r.hd, buf.data(), 16);
  r.size = (uint32_t)buf.size();
  ctmr::Walk w;
  const bool ok = ctmr::walk_cert(r, len, w, nullptr, true, strings != 0, ext != 0);
  out[0] = r.refills; out[1] = r.misses; out[2] = r.first_refill_pos; out[3] = r.first_miss_pos; out[4] = r.coop; out[5] = r.deferred;
  return ok;
}

// k_ec_resolve's loader + equation on the host: the point whose X starts at BIT `xbit` of buf (RightAlign as a bit offset)
extern "C" int harness_ec_point_bits(const uint8_t* buf, uint32_t len, uint64_t xbit, int curve) {
  std::vector<uint32_t> w((len + 3) / 4 + 40, 0xa5a5a5a5u);
  memcpy(w.data(), buf, len);
  auto run = [&](auto tag) {
    using C = decltype(tag);
    uint32_t x[C::NL], y[C::NL];
    ctmr::fe_load_bits<C>(w.data(), xbit, x);
    ctmr::fe_load_bits<C>(w.data(), xbit + 8ull * C::BYTES, y);
    return (int)ctmr::ec_equation<C>(x, y);
  };
  switch (curve) {
    case 1: return run(ctmr::CurveP256{});
    case 2: return run(ctmr::CurveP384{});
    case 3: return run(ctmr::CurveP521{});
    case 4: return run(ctmr::CurveP224{});
    default: return run(ctmr::CurveP192{});
  }
}
