// spki_key.h — the public key inside subjectPublicKeyInfo: what certificate-transparency-go's parsePublicKey makes of
// it (x509.ParseCertificate → parseCertificate → parsePublicKey; call sites cmd/ct-fetch/ct-fetch.go:202,221,452).
// Included by der_walk.h (it uses rd_hdr / ldc / int_check from there); one certificate per lane like the rest of the walk.
//
// Rounds 1–3 skipped the key bits by length; a certificate whose key does not parse never reaches Store in the reference
// (a fatal error drops the entry in every role), so the walk now restates parsePublicKey for the algorithms it knows —
// as recalled from CT-go v1.1.0, which is not on this machine (PARITY UNPINNED like the rest of the CT-go
// boundary; OpenSSL's X509_get_pubkey is the independent opinion the tests use):
//   asn1Data = PublicKey.RightAlign()            (a BIT STRING with pad bits is shifted right by the pad count)
//   rsaEncryption 1.2.840.113549.1.1.1 (and RSAES-OAEP …1.7, key part only)
//       parameters not exactly NULL (05 00)                      → finding (non-fatal)   [rsaEncryption only]
//       asn1Data = SEQUENCE { modulus INTEGER, publicExponent INTEGER (an `int`: ≤ 8 octets) }, nothing behind it
//                  (bytes behind the exponent INSIDE the SEQUENCE are ignored, as for every struct)  → else fatal
//       an INTEGER that is not minimally encoded                 → finding (only the lax re-parse accepts it)
//       modulus ≤ 0                                              → finding;  publicExponent ≤ 0 → fatal
//   id-dsa 1.2.840.10040.4.1
//       asn1Data = INTEGER y, nothing behind it; parameters = SEQUENCE { p, q, g INTEGER } (strict parse: minimal);
//       y, p, q, g > 0                                           → else fatal;  y not minimal → finding
//   id-ecPublicKey 1.2.840.10045.2.1
//       parameters = one OBJECT IDENTIFIER naming P-224 / P-256 / P-384 / P-521 (crypto/elliptic) or secp192r1
//       (CT-go's own; accepted with a finding)                   → else fatal
//       elliptic.Unmarshal(curve, asn1Data): 1 + 2·⌈bits/8⌉ octets, first 04, x < p, y < p, y² = x³ − 3x + b (mod p)
//                                                                → else fatal
//   any other algorithm (Ed25519 included — whether v1.1.0 knew it is not recoverable here): the key is not looked at.
// Findings go to Walk.nonfatal as WALK_NF_SPKI: the reference keeps such a certificate as an X509 entry and drops it
// as a precertificate or a Chain[0] issuer (der_walk.h Walk.nonfatal).
//
// Byte access: the algorithm, the BIT STRING header and the first key octets sit in the window the walk holds at that
// point; the exponent, a curve point, DSA parameters do not — they are read through key_reader_of(reader) (below).
#pragma once
#include "ec_curves.h"

namespace ctmr {

constexpr uint32_t WALK_NF_SPKI = 8u;  // parsePublicKey filed a non-fatal finding (see above)

// Byte access for the key: Reader::key_bytes() — a SMALL reader by value ("anywhere in the certificate": the window when it
// holds the bytes, else a plain global load; no miss bookkeeping) — for readers that have one; the others (global-memory and
// host readers, whose ld4 works anywhere) are used through a reference.  By value matters on the GPU: a view that held a
// reference to the map kernel's window reader (which every ld4 updates) kept that reader in scratch memory — 144 bytes of
// private segment and stores all over the walk (measured, round 4).
template <class R>
struct RefReader {
  const R& r;
  CTMR_HD uint32_t ld4(uint32_t pos) const { return r.ld4(pos); }
};
template <class R, class = void>
struct has_key_bytes : std::false_type {};
template <class R>
struct has_key_bytes<R, std::void_t<decltype(std::declval<const R&>().key_bytes())>> : std::true_type {};
template <class R>
CTMR_HD auto key_reader_of(const R& r) {
  if constexpr (has_key_bytes<R>::value) return r.key_bytes();
  else return RefReader<R>{r};
}
template <class R>
using KeyReaderOf = decltype(key_reader_of(std::declval<const R&>()));

// Where the curve equation runs.  Four modular products over 6..17 limbs per lane need 84 (P-256) to 170 VGPRs; inlined
// into the map kernel they took it from 121 to 245 VGPRs plus scratch, and as out-of-line calls they made it a non-leaf
// kernel whose hot path spilled (37.7 instead of 22.6 ms per 100 M certificates, measured, round 4).  So the walk of a MAP
// kernel does everything about an EC key except the equation (algorithm, named curve, length, the 04) and hands the point's
// place out (walk_cert<…, EC_DEFER = true> → Walk.ec_*): the entry leaves the map "key pending" and k_ec_resolve
// (kernels/reduce.h) — a kernel of its own, with its own register file, that exits at once when no entry is pending —
// checks the point before the entry is inserted.  Every other caller of the walk (issuer registration, the strict_leaf
// TBS check, the host builds) evaluates the equation on the spot through one out-of-line function per curve
// (Reader::raw(): the bytes in memory), so a kernel's register file is sized by the largest ONE of them.
// Readers without raw() (the test harness's byte-counting reader) take the same code inline.
#if defined(__HIPCC__)
#define CTMR_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define CTMR_HD_NOINLINE __attribute__((noinline))
#endif
struct RawCert {
  const uint32_t* words;  // 4-byte aligned base of the buffer the certificate lies in
  uint64_t base;          // the certificate's first octet
};
struct RawReader {
  RawCert c;
  CTMR_HD uint32_t ld4(uint32_t pos) const {
    const uint64_t a = c.base + pos, i = a >> 2;
    const uint32_t sh = 8u * ((uint32_t)a & 3u);
    const uint32_t lo = c.words[i], hi = c.words[i + 1];
    return sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
  }
};
template <class R, class = void>
struct has_raw : std::false_type {};
template <class R>
struct has_raw<R, std::void_t<decltype(std::declval<const R&>().raw())>> : std::true_type {};

// The right-aligned key octets as a reader in CERTIFICATE coordinates: octet at `pos` of the aligned string =
// B[pos-1] << (8-shift) | B[pos] >> shift with B[c0-1] = 0 (encoding/asn1 BitString.RightAlign; shift = pad count).
template <class K>
struct SpkiView {
  K r;  // key_reader_of(reader): by value
  uint32_t c0, shift;
  // sixteen octets fetched ahead (around the end of the key: an RSA key's publicExponent — Reader::key_tail), served from
  // registers.  (Part of this view, not a view over it: a reader reached through two levels of references stays in
  // scratch memory.)
  uint32_t at = 0u, e0 = 0u, e1 = 0u, e2 = 0u, e3 = 0u;
  bool pre = false;
  CTMR_HD uint32_t ld4(uint32_t pos) const {
    const uint32_t off = pos - at;
    if (pre & (off <= 12u)) {  // (64-bit shifts, not selects by `off`: the compiler turns those into a table in scratch memory)
      const unsigned long long a = (unsigned long long)e0 | ((unsigned long long)e1 << 32);
      const unsigned long long b = (unsigned long long)e2 | ((unsigned long long)e3 << 32);
      const uint32_t sh = 8u * (off & 7u);
      const unsigned long long lo64 = (off & 8u) ? b : a, hi64 = (off & 8u) ? 0ull : b;
      return (uint32_t)(sh ? ((lo64 >> sh) | (hi64 << (64u - sh))) : lo64);
    }
    if (shift == 0u) return r.ld4(pos);
    const uint32_t lo = r.ld4(pos - 1u), hi = r.ld4(pos + 3u);
    unsigned long long be = ((unsigned long long)__builtin_bswap32(lo) << 8) | (hi & 0xffu);  // B[pos-1] … B[pos+3]
    if (pos == c0) be &= 0xffffffffull;
    return __builtin_bswap32((uint32_t)(be >> shift));
  }
};

// INTEGER at p inside [p, end): tag, fit, checkInteger (empty → fatal, not minimal → finding).  sign: −1, 0, +1 of the
// value (a zero-padded, not minimal encoding is scanned for a non-zero octet: rare).
template <class V>
CTMR_HD void key_integer(const V& v, uint32_t L, uint32_t p, uint32_t end, bool& ok, uint32_t& nf, uint32_t& after,
                         int& sign, uint32_t& len) {
  uint32_t tag, cs, ce;
  rd_hdr(v, L, p, end, ok, tag, cs, ce);
  ok = ok & (tag == 0x02u);
  len = ce - cs;
  const uint32_t w = ldc(v, cs, L);
  const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu;
  const bool pad0 = (len > 1u) & (b0 == 0x00u) & ((b1 & 0x80u) == 0u);
  const bool padf = (len > 1u) & (b0 == 0xffu) & ((b1 & 0x80u) != 0u);
  ok = ok & (len != 0u);
  nf = (pad0 | padf) ? (nf | WALK_NF_SPKI) : nf;
  sign = (b0 & 0x80u) ? -1 : ((len == 1u) & (b0 == 0u)) ? 0 : 1;
  if (ok & pad0) {  // 00 00 …: zero unless some octet is not
    bool nz = false;
    for (uint32_t x = cs; !nz & (x < ce); x += 4u) {
      const uint32_t rem = ce - x;
      const uint32_t keep = rem >= 4u ? 0xffffffffu : ((1u << (8u * rem)) - 1u);
      nz = (ldc(v, x, L) & keep) != 0u;
    }
    sign = nz ? 1 : 0;
  }
  after = ce;
}

// ---------------------------------------------------------------- the curve equation, per lane, in registers
// Montgomery product b ← a·b·R⁻¹ mod p (CIOS, 32-bit limbs; inputs < p, result < p), in place: b is consumed by shifting
// it down one limb per step of the rolled outer loop, so one product is 2·NL multiply-adds of straight-line code and its
// live state is a, b and the NL+1 accumulator limbs (measured alone: 84 VGPRs for P-256, 124 for P-384).
// P-256 (every EC key of the public logs but a few): p = 2^256 − 2^224 + 2^192 + 2^96 − 1 and −p⁻¹ ≡ 1 (mod 2^32), so one
// reduction step t ← (t + t[0]·p) / 2^32 needs no multiplication: limb 0 cancels, and what is left of t[0]·p is ± t[0] at
// the limbs 3, 6, 7, 8 — seven additions with carry instead of eight 32 × 32 → 64-bit multiply-adds, which run at a quarter
// of the vector rate (round 5; the generic CIOS product spent half its multiplications there).
// top: limb 8 of t as the multiply-add row left it — up to 33 bits (t_prev + b_i·a ≤ p − 1 + 2^32·a can pass 2^288 when a
// is within ≈ 2^160 of p; the ninth limb then needs its carry, which rounds 5's `(uint32_t)top` dropped: ADVICE r05)
CTMR_HD void p256_redc_step(uint32_t (&t)[9], unsigned long long top) {
  const long long m = (long long)t[0];
  t[0] = t[1];
  t[1] = t[2];
  long long s = (long long)t[3] + m;
  t[2] = (uint32_t)s;
  s = (long long)t[4] + (s >> 32);
  t[3] = (uint32_t)s;
  s = (long long)t[5] + (s >> 32);
  t[4] = (uint32_t)s;
  s = (long long)t[6] + m + (s >> 32);
  t[5] = (uint32_t)s;
  s = (long long)t[7] - m + (s >> 32);  // (an arithmetic shift: the carry may be −1 here)
  t[6] = (uint32_t)s;
  s = (long long)top + m + (s >> 32);
  t[7] = (uint32_t)s;
  t[8] = (uint32_t)(s >> 32);
}
template <class C>
struct is_p256 : std::false_type {};
template <>
struct is_p256<CurveP256> : std::true_type {};

template <class C>
CTMR_HD void mont_mul(const uint32_t (&a)[C::NL], uint32_t (&b)[C::NL]) {
  constexpr int NL = C::NL;
  uint32_t t[NL + 1];
#pragma unroll
  for (int j = 0; j <= NL; j++) t[j] = 0u;
#pragma unroll 1
  for (int i = 0; i < NL; i++) {
    const uint32_t bi = b[0];
#pragma unroll
    for (int j = 0; j + 1 < NL; j++) b[j] = b[j + 1];
    unsigned long long c = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const unsigned long long s = (unsigned long long)a[j] * bi + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    unsigned long long top = (unsigned long long)t[NL] + c;
    if constexpr (is_p256<C>::value) {
      p256_redc_step(t, top);
    } else {
      const uint32_t m = t[0] * C::N0;
      unsigned long long s = (unsigned long long)m * C::P[0] + t[0];
      c = s >> 32;
#pragma unroll
      for (int j = 1; j < NL; j++) {
        s = (unsigned long long)m * C::P[j] + t[j] + c;
        t[j - 1] = (uint32_t)s;
        c = s >> 32;
      }
      top += c;
      t[NL - 1] = (uint32_t)top;
      t[NL] = (uint32_t)(top >> 32);
    }
  }
  // t < 2p: one conditional subtraction (the borrow chain first, then the subtraction itself: no second array)
  unsigned long long br = 0ull;
#pragma unroll
  for (int j = 0; j < NL; j++) br = (((unsigned long long)t[j] - C::P[j] - br) >> 32) & 1ull;
  const bool ge = (t[NL] != 0u) | (br == 0ull);
  br = 0ull;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const unsigned long long s = (unsigned long long)t[j] - (ge ? C::P[j] : 0u) - br;
    b[j] = (uint32_t)s;
    br = (s >> 32) & 1ull;
  }
}

template <class C>
CTMR_HD bool fe_lt_p(const uint32_t (&a)[C::NL]) {
  unsigned long long br = 0ull;
#pragma unroll
  for (int j = 0; j < C::NL; j++) br = (((unsigned long long)a[j] - C::P[j] - br) >> 32) & 1ull;
  return br != 0ull;
}

// One coordinate: BYTES big-endian octets at `pos` of the view → little-endian limbs.
template <class C, class V>
CTMR_HD void fe_load(const V& v, uint32_t L, uint32_t pos, uint32_t (&out)[C::NL]) {
  constexpr uint32_t top = C::BYTES & 3u;  // octets of the partial top limb (P-521: 2)
  const uint32_t end = pos + C::BYTES;
#pragma unroll
  for (int j = 0; j < C::NL; j++) {
    if (4u * (uint32_t)(j + 1) <= C::BYTES) {
      out[j] = __builtin_bswap32(ldc(v, end - 4u * (uint32_t)(j + 1), L));
    } else {
      out[j] = __builtin_bswap32(ldc(v, pos, L)) >> (8u * ((4u - top) & 3u));  // (never reached when top == 0)
    }
  }
}

// elliptic.Unmarshal's tests behind the length and the 04: x < p, y < p, y² = x³ − 3x + b.  Evaluated as
//   mont(mont(y,y), 1) == mont(mont(x,x) − 3R⁻¹, x) + bR⁻²   (both sides carry R⁻²; four products, one of them by 1).
// x and y are consumed.
template <class C>
CTMR_HD bool ec_equation(uint32_t (&x)[C::NL], uint32_t (&y)[C::NL]) {
  constexpr int NL = C::NL;
  const bool in_range = fe_lt_p<C>(x) & fe_lt_p<C>(y);
  uint32_t u[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) u[j] = x[j];
  mont_mul<C>(x, u);  // u = x²R⁻¹
  {                   // u −= 3R⁻¹ (mod p)
    unsigned long long br = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const unsigned long long s = (unsigned long long)u[j] - C::C3[j] - br;
      u[j] = (uint32_t)s;
      br = (s >> 32) & 1ull;
    }
    unsigned long long c = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const unsigned long long s = (unsigned long long)u[j] + (br ? C::P[j] : 0u) + c;
      u[j] = (uint32_t)s;
      c = s >> 32;
    }
  }
  mont_mul<C>(u, x);  // x = (x² − 3)·x·R⁻²
  {                   // x += bR⁻² (mod p)
    unsigned long long c = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const unsigned long long s = (unsigned long long)x[j] + C::CB[j] + c;
      x[j] = (uint32_t)s;
      c = s >> 32;
    }
    unsigned long long br = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) br = (((unsigned long long)x[j] - C::P[j] - br) >> 32) & 1ull;
    const bool ge = (c != 0ull) | (br == 0ull);
    br = 0ull;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const unsigned long long s = (unsigned long long)x[j] - (ge ? C::P[j] : 0u) - br;
      x[j] = (uint32_t)s;
      br = (s >> 32) & 1ull;
    }
  }
#pragma unroll
  for (int j = 0; j < NL; j++) u[j] = y[j];
  mont_mul<C>(y, u);  // u = y²R⁻¹
  if constexpr (is_p256<C>::value) {  // u·1·R⁻¹: eight reduction steps, no multiplication at all
    uint32_t t[9];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = u[j];
    t[8] = 0u;
#pragma unroll 1
    for (int i = 0; i < 8; i++) p256_redc_step(t, t[8]);
    unsigned long long br = 0ull;  // t < p + 1 ≤ 2p: one conditional subtraction
#pragma unroll
    for (int j = 0; j < 8; j++) br = (((unsigned long long)t[j] - C::P[j] - br) >> 32) & 1ull;
    const bool ge = (t[8] != 0u) | (br == 0ull);
    br = 0ull;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const unsigned long long d = (unsigned long long)t[j] - (ge ? C::P[j] : 0u) - br;
      u[j] = (uint32_t)d;
      br = (d >> 32) & 1ull;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NL; j++) y[j] = j == 0 ? 1u : 0u;
    mont_mul<C>(y, u);  // u = y²R⁻²
  }
  uint32_t diff = 0u;
#pragma unroll
  for (int j = 0; j < NL; j++) diff |= x[j] ^ u[j];
  return in_range & (diff == 0u);
}

template <class C, class V>
CTMR_HD bool ec_on_curve(const V& v, uint32_t L, uint32_t pos) {
  uint32_t x[C::NL], y[C::NL];
  fe_load<C>(v, L, pos, x);
  fe_load<C>(v, L, pos + C::BYTES, y);
  return ec_equation<C>(x, y);
}

// One coordinate straight out of memory, for k_ec_resolve: the C::BYTES octets that START at bit `bit` of the buffer
// `words` (bit = 8·octet position − the BIT STRING's pad count: BitString.RightAlign as a bit offset; bits count from the
// most significant bit of octet 0), as little-endian limbs.  NL + 1 aligned dwords, loaded unconditionally and together —
// one memory latency — and funnel-shifted into place.  (Through a byte reader every 4-octet read is a branch and a wait
// of its own: ≈ 70 dependent round trips per point, 80 ms per 47 M points, measured.)
template <class C>
CTMR_HD void fe_load_bits(const uint32_t* words, unsigned long long bit, uint32_t (&out)[C::NL]) {
  constexpr int NL = C::NL;
  constexpr uint32_t lead = 32u * (uint32_t)NL - 8u * C::BYTES;  // unused high bits of the top limb (P-521: 16)
  const unsigned long long start = bit - lead;                  // the limbs as NL whole big-endian words from here
  const unsigned long long d0 = start >> 5;
  const uint32_t r = (uint32_t)start & 31u;
  uint32_t w[NL + 1];
#pragma unroll
  for (int k = 0; k <= NL; k++) w[k] = __builtin_bswap32(words[d0 + (unsigned long long)k]);
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const uint32_t v = r ? ((w[k] << r) | (w[k + 1] >> (32u - r))) : w[k];
    out[NL - 1 - k] = (k == 0 && lead) ? (v & (0xffffffffu >> lead)) : v;
  }
}

enum : uint32_t { PK_OTHER = 0, PK_RSA = 1, PK_RSA_OAEP = 2, PK_DSA = 3, PK_EC = 4 };

// getPublicKeyAlgorithmFromOID on the OID's octets (a minimal encoding is the only one parseObjectIdentifier accepts,
// so equal arcs = equal octets).  The three reads lie in the window the walk holds (the SPKI starts there).
template <class R>
CTMR_HD uint32_t spki_algorithm(const R& r, uint32_t L, const AlgView& a) {
  const uint32_t n = a.oid_e - a.oid_c;
  const uint32_t w0 = ldc(r, a.oid_c, L), w1 = ldc(r, a.oid_c + 4u, L), w2 = ldc(r, a.oid_c + 8u, L);
  const bool rsa_arc = (n == 9u) & (w0 == 0x8648862au) & (w1 == 0x01010df7u);
  const bool x9 = (n == 7u) & (w0 == 0xce48862au);
  uint32_t alg = PK_OTHER;
  alg = (rsa_arc & ((w2 & 0xffu) == 0x01u)) ? PK_RSA : alg;
  alg = (rsa_arc & ((w2 & 0xffu) == 0x07u)) ? PK_RSA_OAEP : alg;
  alg = (x9 & ((w1 & 0xffffffu) == 0x010438u)) ? PK_DSA : alg;
  alg = (x9 & ((w1 & 0xffffffu) == 0x01023du)) ? PK_EC : alg;
  return alg;
}

// What phase 1 leaves for phase 2 (the walk fills its next window in between).  An RSA key's publicExponent — its TLV
// starts where the modulus ends, ~0.3 KB behind everything the window holds — is read in phase 2, out of the sixteen
// octets around the key's end that a windowed reader fetches together with that next window (no round trip of its own).
struct KeyPending {
  uint32_t alg;        // PK_*; PK_OTHER also stands for "nothing left to check"
  uint32_t c0, ek, shift;
  uint32_t e_pos;      // RSA: where the publicExponent's TLV starts; seq_end in e_end
  uint32_t e_end;
  uint32_t k_at, k0, k1, k2, k3;  // sixteen octets of the certificate from k_at on, fetched with the walk's tail (key_tail_of)
  bool pre;            // … are valid
  bool fast;           // RSA: the key had the common shape up to the modulus (spki_key_begin): the exponent may take the short way too
};

// Readers that fetch the octets around the key's end together with their next window (kernels/readers.h WinReaderS::
// touch_tail) hand them over here; for the others the exponent is read where it lies.
template <class R, class = void>
struct has_key_tail : std::false_type {};
template <class R>
struct has_key_tail<R, std::void_t<decltype(std::declval<const R&>().key_tail(0u))>> : std::true_type {};

template <class R>
CTMR_HD void key_tail_of(const R& r, uint32_t q, KeyPending& kp);  // (defined behind KeyPending's users)

// Phase 1 — behind the SubjectPublicKeyInfo's own checks, with the window on its first octets.  [ck, ek) = the BIT
// STRING's content (ck = the pad octet, already validated by bit_string_check).
template <class R>
CTMR_HD void spki_key_begin(const R& r, uint32_t L, const AlgView& a, uint32_t ck, uint32_t ek, bool& ok, uint32_t& nf,
                            KeyPending& kp) {
  kp.alg = PK_OTHER;
  kp.pre = false;
  kp.fast = false;
  kp.c0 = ck + 1u;
  kp.ek = ek;
  kp.shift = 0u;
  kp.e_pos = kp.e_end = 0u;
  kp.k_at = kp.k0 = kp.k1 = kp.k2 = kp.k3 = 0u;
  if (!ok) return;
  const uint32_t alg = spki_algorithm(r, L, a);
  if (alg == PK_OTHER) return;
  kp.alg = alg;
  kp.shift = ldc(r, ck, L) & 0xffu;  // ≤ 7 (bit_string_check)
  if ((alg == PK_RSA) | (alg == PK_RSA_OAEP)) {
    if (alg == PK_RSA) {  // bytes.Equal(Parameters.FullBytes, asn1.NullBytes)
      const bool null_par = (a.par_e - a.par_p == 2u) & ((ldc(r, a.par_p, L) & 0xffffu) == 0x0005u);
      nf = null_par ? nf : (nf | WALK_NF_SPKI);
    }
    // The shape every RSA key of 2048 bits and more has — 30 82 ll ll | 02 82 nn nn | modulus, minimal and positive — is
    // recognised from three reads of the window (the general parse below costs ≈ 100 vector instructions more per
    // certificate, and on a power-limited part instructions are clock: DESIGN.md §7).  Anything else takes the general parse.
    const uint32_t h0 = ldc(r, kp.c0, L), h1 = ldc(r, kp.c0 + 4u, L), h2 = ld2(r, kp.c0 + 8u, L);  // (two octets: SPKI_HEAD_NEED)
    const uint32_t seqlen = __builtin_bswap32(h0) & 0xffffu, nlen = __builtin_bswap32(h1) & 0xffffu;
    const uint32_t b0 = h2 & 0xffu, b1 = (h2 >> 8) & 0xffu;
    const bool shape = (kp.shift == 0u) & ((h0 & 0xffffu) == 0x8230u) & (seqlen >= 256u) & (kp.c0 + 4u + seqlen == ek) &
                       ((h1 & 0xffffu) == 0x8202u) & (nlen >= 256u) & (nlen + 4u < seqlen) &
                       (((b0 - 1u) < 0x7fu) | ((b0 == 0u) & (b1 >= 0x80u)));
    kp.fast = shape;
    if (shape) {
      kp.e_pos = kp.c0 + 8u + nlen;
      kp.e_end = ek;
    } else {
      const SpkiView<KeyReaderOf<R>> v{key_reader_of(r), kp.c0, kp.shift};
      uint32_t ts, ss, se, n_end, n_len;
      int n_sign;
      rd_hdr(v, L, kp.c0, ek, ok, ts, ss, se);
      ok = ok & (ts == 0x30u) & (se == ek);  // "x509: trailing data after RSA public key"
      key_integer(v, L, ss, se, ok, nf, n_end, n_sign, n_len);
      nf = (n_sign <= 0) ? (nf | WALK_NF_SPKI) : nf;  // "x509: RSA modulus is not a positive number"
      kp.e_pos = n_end;
      kp.e_end = se;
    }
  }
}

template <class R>
CTMR_HD void key_tail_of(const R& r, uint32_t q, KeyPending& kp) {
  if constexpr (has_key_tail<R>::value) {
    const auto t = r.key_tail(q);
    kp.pre = t.valid & (kp.shift == 0u);
    kp.k_at = t.at; kp.k0 = t.w[0]; kp.k1 = t.w[1]; kp.k2 = t.w[2]; kp.k3 = t.w[3];
  }
}

template <class C, class V>
CTMR_HD bool ec_point_ok(const V& v, uint32_t L, uint32_t c0, uint32_t ek) {
  if (ek - c0 != 1u + 2u * C::BYTES) return false;
  if ((ldc(v, c0, L) & 0xffu) != 0x04u) return false;
  return ec_on_curve<C>(v, L, c0 + 1u);
}

// DSA: asn1Data = INTEGER y, nothing behind it; Parameters.FullBytes [par_p, par_e) = SEQUENCE { p, q, g } by the strict
// parser; all four positive.  Returns ok; *nf_out = 1 when y is only accepted by the lax re-parse.
template <class V, class PV>
CTMR_HD bool dsa_key_body(const V& v, const PV& pv, uint32_t L, uint32_t c0, uint32_t ek, uint32_t par_p, uint32_t par_e,
                          uint32_t* nf_out) {
  bool ok = true;
  uint32_t nf = 0u, nfp = 0u, after, len;
  int sign;
  key_integer(v, L, c0, ek, ok, nf, after, sign, len);
  ok = ok & (after == ek) & (sign > 0);  // "trailing data after DSA public key"; "zero or negative DSA parameter"
  uint32_t tp, ps, pe;
  ok = ok & (par_e != par_p);            // absent parameters: Unmarshal of nothing fails
  rd_hdr(pv, L, par_p, par_e, ok, tp, ps, pe);
  ok = ok & (tp == 0x30u);
  uint32_t q = ps;
  for (int k = 0; k < 3; k++) {
    key_integer(pv, L, q, pe, ok, nfp, after, sign, len);
    ok = ok & (sign > 0);
    q = after;
  }
  *nf_out = nf;
  return ok & (nfp == 0u);               // no lax re-parse for the parameters
}

// One out-of-line function per curve: the register file a kernel needs is the largest ONE of them, not their union.
template <class C>
CTMR_HD_NOINLINE bool ec_key_far(RawCert rc, uint32_t L, uint32_t c0, uint32_t ek, uint32_t shift) {
  const RawReader rr{rc};
  const SpkiView<RawReader> v{rr, c0, shift};
  return ec_point_ok<C>(v, L, c0, ek);
}
// elliptic.Unmarshal for the curve the parameters name (curve = 1..5: P-256, P-384, P-521, P-224, secp192r1)
template <class R>
CTMR_HD bool ec_key_check(const R& r, const SpkiView<KeyReaderOf<R>>& v, uint32_t L, uint32_t curve, uint32_t c0, uint32_t ek) {
  if constexpr (has_raw<R>::value) {
    const RawCert rc = r.raw();
    switch (curve) {
      case 1u: return ec_key_far<CurveP256>(rc, L, c0, ek, v.shift);
      case 2u: return ec_key_far<CurveP384>(rc, L, c0, ek, v.shift);
      case 3u: return ec_key_far<CurveP521>(rc, L, c0, ek, v.shift);
      case 4u: return ec_key_far<CurveP224>(rc, L, c0, ek, v.shift);
      default: return ec_key_far<CurveP192>(rc, L, c0, ek, v.shift);
    }
  } else {
    switch (curve) {
      case 1u: return ec_point_ok<CurveP256>(v, L, c0, ek);
      case 2u: return ec_point_ok<CurveP384>(v, L, c0, ek);
      case 3u: return ec_point_ok<CurveP521>(v, L, c0, ek);
      case 4u: return ec_point_ok<CurveP224>(v, L, c0, ek);
      default: return ec_point_ok<CurveP192>(v, L, c0, ek);
    }
  }
}
CTMR_HD_NOINLINE bool dsa_key_far(RawCert rc, uint32_t L, uint32_t c0, uint32_t ek, uint32_t shift, uint32_t par_p,
                                  uint32_t par_e, uint32_t* nf_out) {
  const RawReader rr{rc};
  const SpkiView<RawReader> v{rr, c0, shift}, pv{rr, 0u, 0u};
  return dsa_key_body(v, pv, L, c0, ek, par_p, par_e, nf_out);
}

// What a deferring walk reports about an EC key it accepted so far: curve (1..5 as in ec_key_check; 0 = nothing pending),
// certificate offset of the point's X coordinate and the BIT STRING's pad count.
struct EcPending {
  uint32_t curve, pos, shift;
};

// Phase 2 — anywhere behind phase 1 (the walk calls it behind its next window fill).  EC_DEFER: see the top of the file.
template <bool EC_DEFER, class R>
CTMR_HD void spki_key_finish(const R& r, uint32_t L, const AlgView& a, const KeyPending& kp, bool& ok, uint32_t& nf,
                             EcPending& ecp) {
  ecp.curve = ecp.pos = ecp.shift = 0u;
  if (!ok | (kp.alg == PK_OTHER)) return;
  const SpkiView<KeyReaderOf<R>> v{key_reader_of(r), kp.c0, kp.shift};
  if ((kp.alg == PK_RSA) | (kp.alg == PK_RSA_OAEP)) {
    const SpkiView<KeyReaderOf<R>> ev{key_reader_of(r), kp.c0, kp.shift, kp.k_at, kp.k0, kp.k1, kp.k2, kp.k3, kp.pre};
    if (kp.fast) {  // the exponent as the last element: 02 len e…, 1..8 octets, minimal and positive — else the general parse
      const uint32_t t = ldc(ev, kp.e_pos, L);
      const uint32_t len = (t >> 8) & 0xffu, e0 = (t >> 16) & 0xffu, e1 = t >> 24;
      if (((t & 0xffu) == 0x02u) & ((len - 1u) < 8u) & (kp.e_pos + 2u + len == kp.e_end) &
          (((e0 - 1u) < 0x7fu) | ((e0 == 0u) & (len > 1u) & (e1 >= 0x80u))))
        return;
    }
    uint32_t e_after, e_len;
    int e_sign;
    key_integer(ev, L, kp.e_pos, kp.e_end, ok, nf, e_after, e_sign, e_len);
    ok = ok & (e_len <= 8u) & (e_sign > 0);  // parseInt64: "integer too large"; "RSA public exponent is not a positive number"
  } else if (kp.alg == PK_DSA) {
    uint32_t nfd = 0u;
    bool good;
    if constexpr (has_raw<R>::value && !EC_DEFER) {
      good = dsa_key_far(r.raw(), L, kp.c0, kp.ek, kp.shift, a.par_p, a.par_e, &nfd);
    } else {  // (a map kernel: no calls — the DSA walk is a few header reads, cheap in registers)
      const SpkiView<KeyReaderOf<R>> pv{key_reader_of(r), 0u, 0u};
      good = dsa_key_body(v, pv, L, kp.c0, kp.ek, a.par_p, a.par_e, &nfd);
    }
    ok = ok & good;
    nf |= nfd;
  } else {  // PK_EC
    // asn1.Unmarshal(Parameters.FullBytes, &namedCurveOID): one OBJECT IDENTIFIER; namedCurveFromOID
    const SpkiView<KeyReaderOf<R>> pv{key_reader_of(r), 0u, 0u};
    const uint32_t n = a.par_e - a.par_c;
    const uint32_t w0 = ldc(pv, a.par_c, L), w1 = ldc(pv, a.par_c + 4u, L);
    const bool is_oid = (a.par_e != a.par_p) & (a.par_tag == 0x06u);
    const bool ansi = is_oid & (n == 8u) & (w0 == 0xce48862au) & ((w1 & 0xffffffu) == 0x01033du);  // 1.2.840.10045.3.1.x
    const bool secg = is_oid & (n == 5u) & (w0 == 0x0004812bu);                                   // 1.3.132.0.x
    const uint32_t a8 = w1 >> 24, s5 = w1 & 0xffu;
    uint32_t curve = 0u;
    curve = (ansi & (a8 == 0x07u)) ? 1u : curve;
    curve = (secg & (s5 == 0x22u)) ? 2u : curve;
    curve = (secg & (s5 == 0x23u)) ? 3u : curve;
    curve = (secg & (s5 == 0x21u)) ? 4u : curve;
    curve = (ansi & (a8 == 0x01u)) ? 5u : curve;
    nf = curve == 5u ? (nf | WALK_NF_SPKI) : nf;  // "insecure curve (secp192r1) specified"
    // elliptic.Unmarshal: the length, the 04 …
    const uint32_t bytes = curve == 1u ? 32u : curve == 2u ? 48u : curve == 3u ? 66u : curve == 4u ? 28u : 24u;
    ok = ok & (curve != 0u) & (kp.ek - kp.c0 == 1u + 2u * bytes) & ((ldc(v, kp.c0, L) & 0xffu) == 0x04u);
    // … and the point: x < p, y < p, on the curve
    if constexpr (EC_DEFER) {
      if (ok) {
        ecp.curve = curve;
        ecp.pos = kp.c0 + 1u;
        ecp.shift = kp.shift;
      }
    } else {
      ok = ok && ec_key_check(r, v, L, curve, kp.c0, kp.ek);
    }
  }
}

}  // namespace ctmr
