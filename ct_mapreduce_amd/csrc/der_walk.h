// der_walk.h — the TBSCertificate field walk of the map kernel.
//
// One function, walk_cert<Reader>(), written for one-certificate-per-lane execution on
// CDNA4: every byte access goes through Reader::ld4(pos), a 4-byte little-endian window at an
// arbitrary byte position (LDS tile or global memory — kernels/readers.h supplies both), so one
// TLV header (tag, length byte, up to two long-form length bytes) costs a single load.
//
// It replaces, for the fields the reference path consumes, the call
// x509.ParseCertificate at cmd/ct-fetch/ct-fetch.go:202,221 and storage.NewSerial
// (storage/types.go:165-178).  The accept/reject profile is DESIGN.md §3.
//
// CTMR_HD lets tests/harness compile this exact code for the host, to fuzz it without a GPU; the
// shipped library instantiates it in device code (plus one host instantiation for the exact
// slow path of serials longer than CTMR_MAX_SERIAL, ctmr_engine.hip).
#pragma once
#include <stdint.h>

#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#define CTMR_HD __host__ __device__ __forceinline__
#else
#define CTMR_HD inline
#endif

namespace ctmr {

struct Walk {
  uint32_t serial_off, serial_len;
  int64_t not_before, not_after;
  uint32_t cn_off, cn_len;
  uint32_t spki_off, spki_len;
  bool bc_valid, is_ca;
  // findings certificate-transparency-go reports as x509.NonFatalErrors (WALK_NF_*): the certificate is handed out
  // all the same.  The reference keeps it when it is an X509 entry (ct.LogEntryFromLeaf drops an entry only on
  // x509.IsFatal errors, cmd/ct-fetch/ct-fetch.go:452-459) and drops it when it is a precertificate or a Chain[0]
  // issuer (any err: :202-209, :221-225) — map_one / k_issuer_ids apply that.
  uint32_t nonfatal;
  // captured while the bytes are at hand (a windowed reader may have moved on afterwards):
  uint32_t serial_w[5];  // first min(20, serial_len) serial octets, little-endian words, zero padded
  bool cn_match;         // some issuerCNFilter piece is a byte prefix of the CommonName
  // where IssuerMetadata.Accumulate's inputs lie (storage/issuermetadata.go:92-138), packed off | len << 16
  // (meta_pack): the issuer Name TLV, and the OCTET STRING content of extension 2.5.29.31
  uint32_t meta_issuer, meta_crl;
  uint32_t issuer_name, subject_name;  // where the two Name TLVs start
  // walk_cert<…, EC_DEFER>: an EC key accepted except for the curve equation (spki_key.h) — ec_curve 1..5 (0: nothing
  // pending), the certificate offset of the point's X coordinate, the BIT STRING's pad count.  The caller owes the check.
  uint32_t ec_curve, ec_pos, ec_shift;
};

constexpr uint32_t WALK_NF_NEGATIVE_SERIAL = 1u;  // "x509: negative serial number"
constexpr uint32_t WALK_NF_LAX_INTEGER = 2u;      // an INTEGER only CT-go's lax asn1 re-parse accepts (not minimal)
constexpr uint32_t WALK_NF_STRING = 4u;           // strict_strings only: a Name value breaks its string type's character set
constexpr uint32_t WALK_EXT_ON = 1u, WALK_EXT_NF = 2u;  // walk_cert's ext_mode: strict_extensions; non-fatal findings inside bodies matter to the caller
constexpr uint32_t WALK_NF_EXT = 16u;             // strict_extensions only: what CT-go files as non-fatal inside an extension body
                                                  // (8u is WALK_NF_SPKI, spki_key.h)

constexpr uint32_t META_NONE = 0u;           // no such element
constexpr uint32_t META_HOST = 0xffffffffu;  // does not fit 16+16 bits, or the extension occurs twice: host parse
CTMR_HD uint32_t meta_pack(uint32_t off, uint32_t len) {
  return ((off > 0xfffeu) | (len > 0xfffeu)) ? META_HOST : (off | (len << 16));
}

// strings.Split(*ctconfig.IssuerCNFilter, ",") — pieces NOT trimmed (ct-fetch.go:57-59)
struct FilterView {
  uint32_t n_pieces;
  const uint32_t* piece_len;   // bytes
  const uint32_t* piece_word;  // index of the piece's first word in words[]
  const uint32_t* words;       // piece bytes, zero padded to 4-byte multiples
};

// Readers whose ld4 is memory-safe at ANY position (the window-only reader clamps into its LDS window) declare
// `static constexpr bool kNoClamp = true`; for all others a read position is clamped to the certificate length
// first.  Either way a read at or past L returns bytes the checks around it never accept.
template <class R>
constexpr auto reader_no_clamp(int) -> decltype(R::kNoClamp, bool()) { return R::kNoClamp; }
template <class R>
constexpr bool reader_no_clamp(...) { return false; }
template <class R>
CTMR_HD uint32_t ldc(const R& r, uint32_t p, uint32_t L) {
  if constexpr (reader_no_clamp<R>(0)) return r.ld4(p);
  else return r.ld4(p < L ? p : L);
}

// Readers whose first window begins BEHIND the outer headers (kernels/readers.h WinGeo::SKIP) hold the certificate's first
// sixteen octets in registers; the walk reads the Certificate / TBSCertificate headers through this view of them.  A read
// that does not lie in them (a high tag number, a length of three or four octets, a certificate shorter than its headers
// say) is remembered, and the caller hands the certificate to the exact reader.
template <class R>
constexpr auto reader_has_head(int) -> decltype(R::kHead, bool()) { return R::kHead; }
template <class R>
constexpr bool reader_has_head(...) { return false; }
struct HeadView {
  static constexpr bool kNoClamp = true;
  uint32_t w0, w1, w2, w3;
  mutable bool out;
  CTMR_HD uint32_t ld4(uint32_t pos) const {
    out = out | (pos > 12u);
    const uint32_t i = pos >> 2;
    const uint32_t lo = i == 0u ? w0 : i == 1u ? w1 : i == 2u ? w2 : w3;
    const uint32_t hi = i == 0u ? w1 : i == 1u ? w2 : i == 2u ? w3 : 0u;
    const uint32_t sh = 8u * (pos & 3u);
    return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
  }
};

// the two octets at p (readers with a clamped two-octet read of their own use it: a window's last two bytes are in reach)
template <class R, class = void>
struct has_ld2c : std::false_type {};
template <class R>
struct has_ld2c<R, std::void_t<decltype(std::declval<const R&>().ld2c(0u))>> : std::true_type {};
template <class R>
CTMR_HD uint32_t ld2(const R& r, uint32_t p, uint32_t L) {
  if constexpr (has_ld2c<R>::value) return r.ld2c(p);
  else return ldc(r, p, L) & 0xffffu;
}

// certIsFilteredOut filter (3): strings.HasPrefix(Issuer.CommonName, piece) for some piece
// (ct-fetch.go:57-69).  Pieces are wave-uniform, the CN bytes per lane.
template <class R>
CTMR_HD bool cn_prefix_match(const R& r, uint32_t L, uint32_t cn_off, uint32_t cn_len,
                             const FilterView& f) {
  for (uint32_t j = 0; j < f.n_pieces; j++) {
    const uint32_t pl = f.piece_len[j];
    if (pl > cn_len) continue;
    const uint32_t* pw = f.words + f.piece_word[j];
    bool eq = true;
    for (uint32_t k = 0; (k < pl) & eq; k += 4) {
      const uint32_t rem = pl - k;
      const uint32_t mask = rem >= 4 ? 0xffffffffu : (0xffffffffu >> (8 * (4 - rem)));
      const uint32_t at = cn_off + k;
      eq = ((ldc(r, at, L) ^ pw[k >> 2]) & mask) == 0;
    }
    if (eq) return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// Control-flow style: SIMD lanes walk different certificates, so the walk never returns early.
// Every check is AND-ed into `ok`; after a failed check the remaining steps run on whatever
// values they have (every load address is clamped to the certificate, every loop also tests
// `ok`), and only the final `ok` is observable.  This keeps the exec-mask bookkeeping down to
// the loops and the few optional elements.

// TLV header at p: tag (the identifier octet), content start, content end.  Go encoding/asn1 parseTagAndLength rules:
// definite minimal lengths < 2^31; the high-tag-number form is allowed (minimal, < 2^31) — it can only ever match an
// ANY position, every expected tag on this path is a low one.  FIT = true: header and contents must lie inside
// [p, end) (Go: "data truncated").  FIT = false: only the header must (an OPTIONAL field that does not match is
// skipped, but parseTagAndLength has run on it; an EXPLICIT wrapper's own length is never checked).
template <bool FIT = true, class R>
CTMR_HD void rd_hdr(const R& r, uint32_t L, uint32_t p, uint32_t end, bool& ok, uint32_t& tag,
                    uint32_t& cs, uint32_t& ce) {
  uint32_t w = ldc(r, p, L);
  tag = w & 0xffu;
  uint32_t lp = p;  // the length octet is at lp + 1
  bool tag_ok = true;
  if ((tag & 0x1fu) == 0x1fu) {  // high-tag-number form (rare): parseBase128Int, <= 5 octets, first != 0x80, value in [31, 2^31)
    const uint32_t x = ldc(r, p + 1u, L), y = ldc(r, p + 5u, L) & 0xffu;
    uint32_t k = 5u;
    unsigned long long v = 0ull;
#pragma unroll
    for (int i = 4; i >= 0; i--) {  // k = index of the first octet without the continuation bit
      const uint32_t bt = i < 4 ? (x >> (8 * i)) & 0xffu : y;
      k = (bt & 0x80u) ? k : (uint32_t)i;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint32_t bt = i < 4 ? (x >> (8 * i)) & 0xffu : y;
      if ((uint32_t)i <= k) v = (v << 7) | (bt & 0x7fu);
    }
    tag_ok = (k < 5u) & ((x & 0xffu) != 0x80u) & (v <= 0x7fffffffull) & (v >= 0x1full);
    lp = p + 1u + (k < 5u ? k : 0u);
    w = ldc(r, lp, L);
  }
  const uint32_t b = (w >> 8) & 0xffu;
  const uint32_t n = b & 0x7fu;
  const bool lng = b >= 0x80u;
  const uint32_t len2 = __builtin_bswap32(w) & 0xffffu;  // the two octets behind the length octet, big endian
  const uint32_t len1 = len2 >> 8;
  // short form | 0x81 vv (vv >= 0x80) | 0x82 hh ll (hh != 0): minimal, no leading zero.  For the long forms with
  // one or two length octets "minimal" is one comparison: value >= 0x40 << n (0x80, 0x100).
  uint32_t len = lng ? (n == 1u ? len1 : len2) : b;
  const uint32_t hl = (lp - p) + (lng ? 2u + n : 2u);
  bool good = !lng | (((n - 1u) <= 1u) & (len >= (0x40u << (n & 3u))));
  if (lng & (n > 2u)) {  // > 64 KiB contents: rare
    const uint32_t q = lp + 2u;
    const uint32_t x = ldc(r, q, L);  // the n length bytes, big endian
    const uint32_t be = __builtin_bswap32(x);
    len = n == 3u ? (be >> 8) : be;
    good = (n <= 4u) & ((x & 0xffu) != 0u) & (len <= 0x7fffffffu);
  }
  good = good & tag_ok;
  const uint32_t c = p + hl;  // p <= 2^31, hl <= 11: no wrap
  cs = c;
  ce = c + len;               // len < 2^31: no wrap; ce >= c >= p, so ONE comparison bounds header and contents
  ok = ok & good & ((FIT ? ce : cs) <= end);
}

// rd_hdr for the places where nearly every header is a low tag number with a short-form length (the bodies of the
// extensions strict_extensions looks into): those two octets decode in a handful of instructions; anything else takes
// rd_hdr's path — behind a branch no lane of a wave of ordinary certificates takes.  Same results as rd_hdr.
template <bool FIT = true, class R>
CTMR_HD void rd_hdr_q(const R& r, uint32_t L, uint32_t p, uint32_t end, bool& ok, uint32_t& tag, uint32_t& cs, uint32_t& ce) {
  const uint32_t w = ldc(r, p, L);
  if (((w & 0x1fu) != 0x1fu) & ((w & 0x8000u) == 0u)) {
    tag = w & 0xffu;
    cs = p + 2u;
    ce = cs + ((w >> 8) & 0x7fu);
    ok = ok & ((FIT ? ce : cs) <= end);
  } else {
    rd_hdr<FIT>(r, L, p, end, ok, tag, cs, ce);
  }
}

CTMR_HD bool digits4(uint32_t w) {  // four ASCII digits?  (SWAR: every byte ^ 0x30 must be <= 9)
  const uint32_t t = w ^ 0x30303030u;
  return ((((t & 0x7f7f7f7fu) + 0x76767676u) | t) & 0x80808080u) == 0u;
}

CTMR_HD uint32_t d2(uint32_t w, int sh) {  // two digits at bit offset sh of w → value
  return ((w >> sh) & 0xfu) * 10u + ((w >> (sh + 8)) & 0xfu);
}

CTMR_HD int64_t days_from_civil(int32_t y, uint32_t m, uint32_t d) {
  y -= m <= 2;
  const int32_t era = (y >= 0 ? y : y - 399) / 400;
  const uint32_t yoe = (uint32_t)(y - era * 400);
  const uint32_t doy = (153u * (m > 2 ? m - 3 : m + 9) + 2u) / 5u + d - 1u;
  const uint32_t doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
  return (int64_t)era * 146097 + (int64_t)doe - 719468;
}

// UTCTime "YYMMDDHHMM[SS]" + zone (tag 0x17) or GeneralizedTime "YYYYMMDDHHMMSS" + zone (tag 0x18) at content offset
// c; zone = 'Z' or a NON-ZERO numeric offset ±hhmm with mm <= 59.  Go asn1 parseUTCTime / parseGeneralizedTime:
// time.Parse with the layouts "0601021504Z0700" (tried first), "060102150405Z0700", "20060102150405Z0700", then the
// result must serialise back to the input — so ±0000 is out (offset 0 prints as "Z"), mm = 60..99 is out (prints as
// the next hour), fractions are out; hh is any two digits (the Go 1.13 toolchain the reference pins does not
// range-check it, and Format prints it back).  UTCTime years: yy < 50 → 20yy, else 19yy (time.Parse's 69 pivot
// plus parseUTCTime's "year >= 2050 → −100").
template <class R>
CTMR_HD void rd_time(const R& r, uint32_t L, uint32_t c, uint32_t tag, uint32_t len, bool& ok,
                     int64_t& out) {
  const uint32_t cc = c < L ? c : L;
  uint32_t w0 = r.ld4(cc), w1 = r.ld4(cc + 4), w2 = r.ld4(cc + 8), w3 = r.ld4(cc + 12), w4 = 0u;
  if (len > 16u) w4 = r.ld4(cc + 16);  // only a numeric zone reaches this far: rare
  const bool gt = tag == 0x18u;
  bool good = gt | (tag == 0x17u);
  uint32_t century = 0;
  uint32_t dl = len;  // length from the two-digit year on
  if (gt) {
    good = good & digits4(w0);
    century = d2(w0, 0);
    // drop the century: shift the 20-byte window down by two bytes → "YYMMDDHHMMSS" + zone
    w0 = (w0 >> 16) | (w1 << 16);
    w1 = (w1 >> 16) | (w2 << 16);
    w2 = (w2 >> 16) | (w3 << 16);
    w3 = (w3 >> 16) | (w4 << 16);
    w4 = w4 >> 16;
    dl = len - 2u;
  }
  // seconds are there iff the 11th character is a digit (the layout without seconds is tried first and fails on it)
  const uint32_t c10 = (w2 >> 16) & 0xffu;
  const bool secs = (c10 - 0x30u) <= 9u;
  good = good & (secs | !gt);
  const uint32_t zw = secs ? w3 : ((w2 >> 16) | (w3 << 16));     // zone characters 0..3
  const uint32_t z4 = secs ? (w4 & 0xffu) : ((w3 >> 16) & 0xffu);  // zone character 4
  const uint32_t zl = dl - (secs ? 12u : 10u);                     // wraps for short strings: neither 1 nor 5
  if (!secs) w2 = (w2 & 0xffffu) | 0x30300000u;                    // seconds = "00"
  good = good & digits4(w0) & digits4(w1) & digits4(w2);
  const uint32_t z0 = zw & 0xffu;
  int32_t zoff = 0;
  bool zone_ok = (zl == 1u) & (z0 == (uint32_t)'Z');
  if (zl == 5u) {  // rare
    const uint32_t dg = (zw >> 8) | (z4 << 24);  // "hhmm"
    const uint32_t zh = d2(dg, 0), zm = d2(dg, 16);
    zone_ok = ((z0 == (uint32_t)'+') | (z0 == (uint32_t)'-')) & digits4(dg) & (zm <= 59u) & ((zh | zm) != 0u);
    zoff = (int32_t)(zh * 3600u + zm * 60u);
    zoff = z0 == (uint32_t)'-' ? -zoff : zoff;
  }
  good = good & zone_ok;
  const uint32_t yy = d2(w0, 0);
  const int32_t year = gt ? (int32_t)(century * 100u + yy) : (int32_t)(yy < 50u ? 2000u + yy : 1900u + yy);
  const uint32_t mon = d2(w0, 16), day = d2(w1, 0), hh = d2(w1, 16), mm = d2(w2, 0), ss = d2(w2, 16);
  uint32_t dim = 31u - ((0xA50u >> (mon & 15u)) & 1u);  // 30-day months: Apr Jun Sep Nov
  const bool leap = (year % 4 == 0) & ((year % 100 != 0) | (year % 400 == 0));
  dim = mon == 2u ? (leap ? 29u : 28u) : dim;
  good = good & (mon >= 1u) & (mon <= 12u) & (day >= 1u) & (day <= dim) & (hh <= 23u) & (mm <= 59u) & (ss <= 59u);
  ok = ok & good;
  out = days_from_civil(year, mon, day) * 86400 + (int64_t)(hh * 3600u + mm * 60u + ss) - (int64_t)zoff;
}

// Go asn1 checkInteger on content [c, c+len): non-empty (else fatal) and minimally encoded (else a finding only
// CT-go's lax re-parse tolerates: WALK_NF_LAX_INTEGER).  neg: the sign bit of the first content octet.
template <class R>
CTMR_HD void int_check(const R& r, uint32_t L, uint32_t c, uint32_t len, bool& ok, uint32_t& nf, bool& neg) {
  const uint32_t w = ldc(r, c, L);
  const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu;
  const bool pad0 = (b0 == 0x00u) & ((b1 & 0x80u) == 0u);
  const bool padf = (b0 == 0xffu) & ((b1 & 0x80u) != 0u);
  ok = ok & (len != 0u);
  nf = ((len > 1u) & (pad0 | padf)) ? (nf | WALK_NF_LAX_INTEGER) : nf;
  neg = (b0 & 0x80u) != 0u;
}

// An `int` field (Version, MaxPathLen): parseInt32 = checkInteger, at most 8 octets, the value fits int32.  A
// minimal encoding fits iff it has at most 4 octets; a non-minimal one (lax) is decoded.
template <class R>
CTMR_HD void int32_check(const R& r, uint32_t L, uint32_t c, uint32_t len, bool& ok, uint32_t& nf) {
  const uint32_t w = ldc(r, c, L);
  const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu;
  const bool pad0 = (b0 == 0x00u) & ((b1 & 0x80u) == 0u);
  const bool padf = (b0 == 0xffu) & ((b1 & 0x80u) != 0u);
  const bool nonmin = (len > 1u) & (pad0 | padf);
  bool good = (len != 0u) & (len <= (nonmin ? 8u : 4u));
  if (nonmin & good) {  // rare
    const uint32_t w1 = ldc(r, c + 4u, L);
    const unsigned long long be = ((unsigned long long)__builtin_bswap32(w) << 32) | __builtin_bswap32(w1);
    const long long v = (long long)be >> (8u * (8u - len));  // the first len octets, sign-extended
    good = v == (long long)(int32_t)v;
    nf |= WALK_NF_LAX_INTEGER;
  }
  ok = ok & good;
}

// Go asn1 parseBitString on content [c, c+len): non-empty, pad <= 7, no pad bits in an empty string, pad bits zero
// (only then is the last octet read — it may be far away).
template <class R>
CTMR_HD void bit_string_check(const R& r, uint32_t L, uint32_t c, uint32_t len, bool& ok) {
  const uint32_t pad = ldc(r, c, L) & 0xffu;
  ok = ok & (len != 0u) & (pad <= 7u) & ((len != 1u) | (pad == 0u));
  if (ok & (pad != 0u)) {
    const uint32_t lastp = c + len - 1u;
    const uint32_t last = ldc(r, lastp, L) & 0xffu;
    ok = (last & ((1u << (pad & 7u)) - 1u)) == 0u;
  }
}

// pkix.AlgorithmIdentifier ::= SEQUENCE { algorithm OBJECT IDENTIFIER, parameters ANY OPTIONAL } at p, inside [p, end):
// the OID must be there, non-empty and end on an octet without the continuation bit (parseObjectIdentifier, as far as
// it is modelled); parameters, when present, must be one well-formed TLV that fits; anything behind is ignored.
// Go asn1 parseObjectIdentifier on content [c, e): a run of base-128 integers (parseBase128Int) — each at most 5 octets,
// its first octet not 0x80 ("integer is not minimally encoded"), its value at most 2^31 − 1 ("base 128 integer too
// large": a five-octet integer whose first octet carries more than three payload bits), the last one complete.  OpenSSL
// rejects the first and the last of these as well ("invalid object encoding"); rounds 1–3 only looked at the last octet.
template <class R>
CTMR_HD bool oid_arcs_exact(const R& r, uint32_t L, uint32_t c, uint32_t e) {  // octet by octet (rare: see oid_arcs_ok)
  bool good = e != c;
  uint32_t run = 0u, first = 0u;  // octets of the integer under way, and its first octet
  for (uint32_t p = c; good & (p < e); p += 4u) {
    const uint32_t w = ldc(r, p, L);
    const uint32_t nb = e - p < 4u ? e - p : 4u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      if (k < nb) {
        const uint32_t b = (w >> (8u * k)) & 0xffu;
        first = run == 0u ? b : first;
        good = good & !((run == 0u) & (b == 0x80u));
        run++;
        good = good & (run <= 5u);
        if (!(b & 0x80u)) {
          good = good & !((run == 5u) & ((first & 0x78u) != 0u));
          run = 0u;
        }
      }
    }
  }
  return good & (run == 0u);
}
// Four octets per step: no integer starts with 0x80 (an octet equal to 0x80 whose predecessor does not continue), the
// last octet ends one, and — as long as the whole OID holds at most three continuation octets, which every OID of the
// X.509 world does — no integer can reach five octets, so the size rules hold by counting.  Otherwise: the exact loop.
template <class R>
CTMR_HD bool oid_arcs_ok(const R& r, uint32_t L, uint32_t c, uint32_t e) {
#ifdef CTMR_EXP_NO_OIDARCS  // measurement build: rounds 1-3 looked at the last octet only
  const uint32_t lastp = e - 1u;
  return (e != c) & ((ldc(r, lastp, L) & 0x80u) == 0u);
#else
  bool good = e != c;
  uint32_t conts = 0u, prev = 0u;  // continuation octets so far; 0x80 when the octet before this word continues
  for (uint32_t p = c; p < e; p += 4u) {
    const uint32_t nb = e - p < 4u ? e - p : 4u;
    const uint32_t keep = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
    const uint32_t w = ldc(r, p, L);
    const uint32_t cb = w & 0x80808080u & keep;
    const uint32_t t = (w ^ 0x80808080u) | ~keep;  // zero octet ⇔ the octet is 0x80
    const uint32_t is80 = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
    const uint32_t starts = ~((cb << 8) | prev);   // bit 7 of octet k: the octet before it does not continue
    good = good & ((is80 & starts) == 0u);
    conts += (uint32_t)__builtin_popcount(cb);
    prev = (cb >> (8u * (nb - 1u))) & 0x80u;
  }
  good = good & (prev == 0u);
  if (good & (conts > 3u)) good = oid_arcs_exact(r, L, c, e);
  return good;
#endif
}

// What alg_id found: the algorithm OID's content octets and the parameters element (absent: par_e == par_p).
struct AlgView {
  uint32_t oid_c, oid_e;
  uint32_t par_p, par_tag, par_c, par_e;
};

template <class R>
CTMR_HD void alg_id(const R& r, uint32_t L, uint32_t p, uint32_t end, bool& ok, uint32_t& after, AlgView& av) {
  uint32_t tag, cs, ce, to, co, eo;
  rd_hdr(r, L, p, end, ok, tag, cs, ce);
  rd_hdr(r, L, cs, ce, ok, to, co, eo);
  ok = ok & (tag == 0x30u) & (to == 0x06u);
  {  // the algorithm OID's arcs: the families every certificate of the Web PKI uses are recognised by two compares (their
     // octets are a valid encoding), anything else takes the general test — which costs the map 1.5 ms per 100 M
     // certificates when all three AlgorithmIdentifiers of every certificate go through it (measured, round 4)
    const uint32_t n = eo - co, w0 = ldc(r, co, L), w1 = ldc(r, co + 4u, L), w2 = ldc(r, co + 8u, L);
    const bool pkcs1 = (n == 9u) & (w0 == 0x8648862au) & (w1 == 0x01010df7u) & ((w2 & 0x80u) == 0u);     // 1.2.840.113549.1.1.x
    const bool x962 = (w0 == 0xce48862au) & (((n == 8u) & ((w1 & 0x80ffffffu) == 0x0003043du)) |             // ecdsa-with-SHA2 …10045.4.3.x
                                             ((n == 7u) & ((w1 & 0x00ffffffu) == 0x0001023du)));              // id-ecPublicKey …10045.2.1
    ok = ok && (pkcs1 | x962 || oid_arcs_ok(r, L, co, eo));
  }
  uint32_t tp = 0u, cp = eo, ep = eo;
  if (ok & (eo < ce)) rd_hdr(r, L, eo, ce, ok, tp, cp, ep);
  av = AlgView{co, eo, eo, tp, cp, ok ? ep : eo};  // (by reference: a pointer that may be null puts the view in scratch memory)
  after = ce;
}
template <class R>
CTMR_HD void alg_id(const R& r, uint32_t L, uint32_t p, uint32_t end, bool& ok, uint32_t& after) {
  AlgView unused;
  alg_id(r, L, p, end, ok, after, unused);
}

CTMR_HD bool string_tag(uint32_t t) {
  // UTF8String, NumericString, PrintableString, T61String, IA5String
  return (t == 0x0cu) | (t == 0x12u) | (t == 0x13u) | (t == 0x14u) | (t == 0x16u);
}

// The three reads behind the TBSCertificate (signatureAlgorithm header, signatureValue header, its pad octet) lie
// ~1 KB past everything else the walk touches: they go through Reader::ldg(pos) — "known to be far away" — so that
// a windowed reader can serve ld4 from its window alone.
template <class R>
struct TailView {
  const R& r;
  CTMR_HD uint32_t ld4(uint32_t pos) const { return r.ldg(pos); }
};

}  // namespace ctmr
#include "spki_key.h"  // the key inside subjectPublicKeyInfo (parsePublicKey)
namespace ctmr {

// strict_strings (opt-in): the character-set rules Go's encoding/asn1 applies when it unmarshals an
// AttributeTypeAndValue's `Value interface{}` of a universal, primitive string type — a violation is a parse error of the
// stdlib; what certificate-transparency-go's lax fork makes of it is NOT verifiable here, so the finding is filed as
// non-fatal (WALK_NF_STRING) behind a switch that is off by default:
//   PrintableString (0x13)  isPrintable with '*' and '&' allowed: A-Z a-z 0-9 space ' ( ) + , - . / : = ? * &
//   NumericString   (0x12)  0-9 and space
//   IA5String       (0x16)  every octet < 0x80
//   UTF8String      (0x0c)  utf8.Valid: no overlong forms, no surrogates, nothing above U+10FFFF, no truncated sequence
// T61String is taken as it is.  walk_name<…, STRINGS = true> checks each value where it meets it, while the window holds it
// (round 4; a second pass over the Name re-read its headers and, on a long subject, found the window moved on).
// Four octets of a PrintableString (0x13) or NumericString (0x12) at once, all of them below 0x80 (the caller has looked):
// every predicate leaves its verdict in bit 7 of each byte — `byte >= c` is bit 7 of byte + (0x80 − c), `byte == c` is
// bit 7 of 0x80 − (byte ^ c); neither carries into the neighbouring byte while all octets are 7-bit.  (Round 4: one
// 64-bit mask lookup per octet before — a third of what strict_strings cost.)
CTMR_HD bool string_word_ok(uint32_t tag, uint32_t w) {
  const uint32_t H = 0x80808080u;
  const auto ge = [](uint32_t x, uint32_t c) { return x + (0x80u - c) * 0x01010101u; };
  const auto eq = [](uint32_t x, uint32_t c) { return 0x80808080u - (x ^ (c * 0x01010101u)); };
  uint32_t okb;
  if (tag == 0x13u) {
    const uint32_t lw = w | 0x20202020u;  // A-Z onto a-z; nothing else lands in a-z
    // a-z A-Z | & ' ( ) * + , - . / 0-9 : (0x26..0x3a) | space | = | ?
    okb = (ge(lw, 0x61u) & ~ge(lw, 0x7bu)) | (ge(w, 0x26u) & ~ge(w, 0x3bu)) | eq(w, 0x20u) | eq(w, 0x3du) | eq(w, 0x3fu);
  } else {
    okb = (ge(w, 0x30u) & ~ge(w, 0x3au)) | eq(w, 0x20u);  // 0-9 | space
  }
  return (okb & H) == H;
}

// the value [cv, ev) of universal type tv against its character set
// Round 6: sixteen octets per step — four independent window reads, one OR — decide what nearly every value needs decided:
// "all octets below 0x80" (every IA5String and every UTF8String of ASCII text is done with that; round 4 took four octets
// per step through the UTF-8 automaton's fast exit: 28 instructions per four octets, 450 per wave of the headline corpus).
// Only a UTF8String that does hold an octet >= 0x80 goes through the automaton, only Printable / NumericStrings through
// their per-word predicate.
template <class R>
CTMR_HD bool utf8_exact_ok(R& r, uint32_t L, uint32_t cv, uint32_t ev) {
  bool good = true;
  uint32_t need = 0u, lo = 0x80u, hi = 0xbfu;  // continuation octets still owed, and the range of the next one
  for (uint32_t p = cv; good & (p < ev); p += 4u) {
    if (((p - cv) & 63u) == 0u) r.touch(p, ev - p < 64u ? ev - p : 64u);
    const uint32_t w = ldc(r, p, L), nb = ev - p < 4u ? ev - p : 4u;
    const uint32_t keep = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
    if ((need == 0u) & ((w & keep & 0x80808080u) == 0u)) continue;  // four ASCII octets between sequences
    for (uint32_t k = 0; k < nb; k++) {
      const uint32_t b = (w >> (8u * k)) & 0xffu;
      if (need == 0u) {
        if (b < 0x80u) {
        } else if ((b >= 0xc2u) & (b <= 0xdfu)) {
          need = 1u;
        } else if ((b >= 0xe0u) & (b <= 0xefu)) {
          need = 2u;
          lo = b == 0xe0u ? 0xa0u : 0x80u;
          hi = b == 0xedu ? 0x9fu : 0xbfu;
        } else if ((b >= 0xf0u) & (b <= 0xf4u)) {
          need = 3u;
          lo = b == 0xf0u ? 0x90u : 0x80u;
          hi = b == 0xf4u ? 0x8fu : 0xbfu;
        } else {
          good = false;
        }
      } else {
        good = good & (b >= lo) & (b <= hi);
        lo = 0x80u;
        hi = 0xbfu;
        need--;
      }
    }
  }
  return good & (need == 0u);
}
template <class R>
CTMR_HD bool value_strings_ok(R& r, uint32_t L, uint32_t tv, uint32_t cv, uint32_t ev) {
  const bool seven = (tv == 0x13u) | (tv == 0x12u) | (tv == 0x16u), utf8 = tv == 0x0cu;
  if (!(seven | utf8)) return true;
  bool good = true;
  uint32_t high = 0u;  // the OR of every octet of the value (bit 7 of each byte lane: some octet >= 0x80)
  for (uint32_t p = cv; p < ev; p += 16u) {
    if (((p - cv) & 63u) == 0u) r.touch(p, ev - p < 64u ? ev - p : 64u);
    const uint32_t n = ev - p;  // octets left, >= 1
    uint32_t w[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      const uint32_t at = p + 4u * k;
      const uint32_t left = n > 4u * k ? n - 4u * k : 0u;  // octets of this word that belong to the value
      const uint32_t keep = left >= 4u ? 0xffffffffu : ((1u << (8u * left)) - 1u);
      w[k] = (left ? ldc(r, at, L) & keep : 0u) | (0x30303030u & ~keep);  // octets behind the value count as '0': in every set
    }
    high |= (w[0] | w[1]) | (w[2] | w[3]);
    if ((tv == 0x13u) | (tv == 0x12u)) {
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) good = good & string_word_ok(tv, w[k] & 0x7f7f7f7fu);
    }
  }
  const bool ascii = (high & 0x80808080u) == 0u;
  if (seven) return good & ascii;                 // the three 7-bit sets
  return ascii || utf8_exact_ok(r, L, cv, ev);    // utf8.Valid: ASCII is; anything else takes the automaton
}

// pkix.RDNSequence at q (asn1.RawValue in the tbsCertificate, then asn1.Unmarshal into pkix.RDNSequence): SEQUENCE OF
// SET OF SEQUENCE { type OID, value ANY }; bytes behind the value inside an AttributeTypeAndValue are ignored.
// CN = true: also finds the last attribute with OID 2.5.4.3 whose value is a string type (the types Go decodes to a
// `string`; pkix.Name.FillFromRDNSequence).  One flattened loop: each iteration decodes either a SET (RDN) header or
// one AttributeTypeAndValue.  Returns the end of the Name.
// An AttributeTypeAndValue's `Value interface{}`: Go's encoding/asn1 decodes a universal, primitive value by its tag, and
// besides the string types (strict_strings) it knows INTEGER (parseInt64: non-empty, at most 8 octets; not minimal →
// only the lax re-parse accepts it), BIT STRING (parseBitString), OBJECT IDENTIFIER (parseObjectIdentifier), UTCTime and
// GeneralizedTime (the validity's own rules) — a value of one of these types that does not parse fails the Name, and with
// it x509.ParseCertificate.  Everything else (OCTET STRING, BOOLEAN, constructed and non-universal values) is taken as it
// is.  value_plain(tag): not one of the five.  Round 4 (DESIGN.md §3.1 listed it as not checked before).
CTMR_HD bool value_plain(uint32_t tag) {
#ifdef CTMR_EXP_NO_NAMEVAL  // measurement build
  return true;
#endif
  return ((tag < 32u) & (((0x0180004cu >> (tag & 31u)) & 1u) != 0u)) == 0;
}
template <class R>
CTMR_HD void name_value_check(const R& r, uint32_t L, uint32_t tag, uint32_t c, uint32_t e, bool& ok, uint32_t& nf) {
  if (tag == 0x02u) {
    bool neg;
    int_check(r, L, c, e - c, ok, nf, neg);
    ok = ok & (e - c <= 8u);
  } else if (tag == 0x03u) {
    bit_string_check(r, L, c, e - c, ok);
  } else if (tag == 0x06u) {
    ok = ok && oid_arcs_ok(r, L, c, e);
  } else {
    int64_t t;
    rd_time(r, L, c, tag, e - c, ok, t);
  }
}

// touch() at a point of the walk that every lane of a wave passes: readers that can refill together do (touch_coop)
template <class R, class = void>
struct has_touch_coop : std::false_type {};
template <class R>
struct has_touch_coop<R, std::void_t<decltype(std::declval<R&>().touch_coop(0u, 0u))>> : std::true_type {};
template <class R>
CTMR_HD void touch_all(R& r, uint32_t pos, uint32_t need) {
  if constexpr (has_touch_coop<R>::value) r.touch_coop(pos, need);
  else r.touch(pos, need);
}

// What the walk reads from the start of a SubjectPublicKeyInfo before the next window is asked for: its header, the
// AlgorithmIdentifier of an RSA key (15 octets), the BIT STRING's header and pad octet, the RSAPublicKey and modulus headers
// and the modulus' first two octets — 34 octets for every RSA key of 2048 bits and more.  (Round 6: the hints used to say
// 48; a 224-byte window holds a synthetic certificate's front only when they say what is read.)
constexpr uint32_t SPKI_HEAD_NEED = 34u;
template <bool CN, bool STRINGS, class R>
CTMR_HD uint32_t walk_name(R& r, uint32_t L, uint32_t q, uint32_t tbs_end, bool& ok, uint32_t& cn_off, uint32_t& cn_len,
                           uint32_t& nf, bool strings) {
  uint32_t tag, cs, ce;
  rd_hdr(r, L, q, tbs_end, ok, tag, cs, ce);
  ok = ok & (tag == 0x30u);
  const uint32_t s_end = ce;
  // a long Name (OV/EV subjects) that runs past the window: refill ONCE, here, where the window then covers the whole
  // Name and the SubjectPublicKeyInfo header behind it — instead of somewhere in the middle and again at the key
  if constexpr (!CN) touch_all(r, cs, ok ? (ce - cs) + SPKI_HEAD_NEED : 0u);
  // (No such hint for the issuer: its Name lies in the first window.  A build that had one — same form, need = the Name's
  //  length — issued a lane-by-lane refill in EVERY wave of the headline corpus, 14 loads and a round trip, fast profile
  //  23.2 instead of 20.9 ms; the window simulator of the CPU harness sees no lane lacking anything there, the listing
  //  shows the threshold computed as written.  Not understood; profiles/r06/issuer_hint_*, EXPERIMENTS.md.)
  uint32_t a = cs, a_end = cs;
  while (ok & (a < s_end)) {
    uint32_t t1, c1, e1;
    r.touch(a, 32);
    if (a == a_end) {  // next RDN
      // Fast form: the 12 bytes at a hold SET hdr, AttributeTypeAndValue hdr, a 3-byte OID with its hdr, and the
      // value hdr — when every length is short form (the usual 2.5.4.x attribute).  One independent 12-byte
      // read replaces five dependent header reads; each header read is an LDS round trip on the critical path.
      const uint32_t w0 = ldc(r, a, L), w1 = ldc(r, a + 4u, L), w2 = ldc(r, a + 8u, L);
      // three one-octet arcs (2.5.4.x), short value length, low value tag — and a value Go does not look into (value_plain)
      const uint32_t tvf = (w2 >> 8) & 0xffu;
      const bool fast = ((w0 & 0x80ff80ffu) == 0x00300031u) & ((w1 & 0x8080ffffu) == 0x00000306u) &
                        ((w2 & 0x800080u) == 0u) & ((w2 & 0x1f00u) != 0x1f00u) & value_plain(tvf);
      if (fast) {
        const uint32_t set_end = a + 2u + ((w0 >> 8) & 0xffu), atv_end = a + 4u + (w0 >> 24);
        const uint32_t tv = tvf, cv = a + 11u, ev = cv + ((w2 >> 16) & 0xffu);
        ok = ok & (set_end <= s_end) & (atv_end <= set_end) & (ev <= atv_end);
        if constexpr (STRINGS)
          if (strings & ok) nf = value_strings_ok(r, L, tv, cv, ev) ? nf : (nf | WALK_NF_STRING);
        if constexpr (CN) {
          const bool is_cn = (((w1 >> 16) | ((w2 & 0xffu) << 16)) == 0x030455u) & string_tag(tv);
          cn_off = is_cn ? cv : cn_off;
          cn_len = is_cn ? ev - cv : cn_len;
        }
        a = atv_end;
        a_end = set_end;
      } else {
        rd_hdr(r, L, a, s_end, ok, t1, c1, e1);
        ok = ok & (t1 == 0x31u);
        a = c1;
        a_end = e1;
      }
    } else {
      uint32_t to, co, eo, tv, cv, ev;
      rd_hdr(r, L, a, a_end, ok, t1, c1, e1);      // AttributeTypeAndValue
      rd_hdr(r, L, c1, e1, ok, to, co, eo);        // type OID
      const uint32_t oidw = ldc(r, co, L);
      rd_hdr(r, L, eo, e1, ok, tv, cv, ev);        // value
      ok = ok & (t1 == 0x30u) & (to == 0x06u);
      ok = ok && oid_arcs_ok(r, L, co, eo);
      if (ok & !value_plain(tv)) name_value_check(r, L, tv, cv, ev, ok, nf);
      if constexpr (STRINGS)
        if (strings & ok) nf = value_strings_ok(r, L, tv, cv, ev) ? nf : (nf | WALK_NF_STRING);
      if constexpr (CN) {
        const bool is_cn = (eo - co == 3u) & ((oidw & 0xffffffu) == 0x030455u) & string_tag(tv);
        cn_off = is_cn ? cv : cn_off;
        cn_len = is_cn ? ev - cv : cn_len;
      }
      a = e1;
    }
  }
  return ce;
}

// The RelativeDistinguishedNames [cs, s_end) without a Name's outer header: what a distribution point's
// nameRelativeToCRLIssuer holds behind its IMPLICIT [1] tag (crl_dps below; strict_extensions only).  The same rules as
// walk_name's loop, without its 12-byte fast form — kept apart from it so that the map kernels' code for the two Names is
// the code that was measured (a shared loop changed the default kernel's register allocation, round 5).
template <bool STRINGS, class R>
CTMR_HD void walk_rdns(R& r, uint32_t L, uint32_t cs, uint32_t s_end, bool& ok, uint32_t& nf, bool strings) {
  uint32_t a = cs;
  while (ok & (a < s_end)) {
    uint32_t t1, c1, e1;
    r.touch(a, 32);
    rd_hdr(r, L, a, s_end, ok, t1, c1, e1);          // RelativeDistinguishedName: a universal SET
    ok = ok & (t1 == 0x31u);
    uint32_t b = c1;
    while (ok & (b < e1)) {
      uint32_t t2, c2, e2, to, co, eo, tv, cv, ev;
      r.touch(b, 32);
      rd_hdr(r, L, b, e1, ok, t2, c2, e2);           // AttributeTypeAndValue
      rd_hdr(r, L, c2, e2, ok, to, co, eo);          // type OID
      rd_hdr(r, L, eo, e2, ok, tv, cv, ev);          // value: must be there and fit; anything behind it is ignored
      ok = ok & (t2 == 0x30u) & (to == 0x06u);
      ok = ok && oid_arcs_ok(r, L, co, eo);
      if (ok & !value_plain(tv)) name_value_check(r, L, tv, cv, ev, ok, nf);
      if constexpr (STRINGS)
        if (strings & ok) nf = value_strings_ok(r, L, tv, cv, ev) ? nf : (nf | WALK_NF_STRING);
      b = e2;
    }
    a = e1;
  }
}

// strict_extensions (opt-in): the BODIES of the extensions Go 1.13's parseCertificate unmarshals with
// plain struct rules — a malformed one is an error of x509.ParseCertificate there (which of them CT-go's fork downgrades
// is not verifiable here: hence a switch, off by default):
//   1 keyUsage 2.5.29.15            one BIT STRING (parseBitString) and nothing behind it
//   2 subjectKeyIdentifier .14      one OCTET STRING and nothing behind it
//   3 extKeyUsage .37               SEQUENCE OF OBJECT IDENTIFIER (every element an OID with valid arcs), nothing behind it
//   4 authorityKeyIdentifier .35    SEQUENCE { [0] IMPLICIT OCTET STRING OPTIONAL, … }: a first element with another tag is
//                                   skipped (its header must parse), whatever follows is ignored; nothing behind the SEQUENCE
//   5 certificatePolicies .32       SEQUENCE OF SEQUENCE { OID, … ignored }
//   6 authorityInfoAccess 1.3.6.1.5.5.7.1.1   SEQUENCE OF SEQUENCE { OID, any TLV that fits, … ignored }, not empty (CT-go)
//     subjectInfoAccess 1.3.6.1.5.5.7.1.11    the same (CT-go's fork only; round 6)
// subjectAltName, nameConstraints and cRLDistributionPoints are NOT modelled (URI parsing, nested optional tags): they stay
// on DESIGN.md's "not checked" list.  The bodies lie right behind their extension headers: the window that holds the
// header usually holds them too.
CTMR_HD uint32_t ext_kind(uint32_t oid_len, uint32_t w0, uint32_t w1) {  // w0, w1: the OID's first eight octets
  if ((oid_len == 3u) & ((w0 & 0xffffu) == 0x1d55u)) {
    const uint32_t arc = (w0 >> 16) & 0xffu;
    return arc == 15u ? 1u : arc == 14u ? 2u : arc == 37u ? 3u : arc == 35u ? 4u : arc == 32u ? 5u :
           arc == 17u ? 7u : arc == 30u ? 8u : arc == 31u ? 9u : 0u;  // 7 subjectAltName, 8 nameConstraints, 9 cRLDistributionPoints
  }
  if ((oid_len == 10u) & (w0 == 0x0401062bu) & (w1 == 0x0279d601u)) return 10u;  // 1.3.6.1.4.1.11129.2.4.x: the caller looks at x
  if ((oid_len == 8u) & (w0 == 0x0501062bu) & ((w1 & 0x00ffffffu) == 0x00010705u)) {  // id-pe 1.3.6.1.5.5.7.1.x
    const uint32_t x = w1 >> 24;
    // .1 authorityInfoAccess, .11 subjectInfoAccess (CT-go's fork parses it like the former: round 6), and RFC 3779's
    // .7 sbgp-ipAddrBlock → 12, .8 sbgp-autonomousSysNum → 13 (CT-go only, non-fatal findings: ext_rpki_ok)
    return ((x == 1u) | (x == 11u)) ? 6u : x == 7u ? 12u : x == 8u ? 13u : 0u;
  }
  return 0u;
}
// oid_arcs_ok with the id-pkix family recognised by two compares (1.3.6.1.5.5.7.x.y: eight octets, all below 0x80 — every arc
// one octet, a valid encoding): the extKeyUsage purposes and the accessMethods of nearly every certificate
template <class R>
CTMR_HD bool oid_arcs_ok_pkix(const R& r, uint32_t L, uint32_t c, uint32_t e) {
  const uint32_t w0 = ldc(r, c, L), w1 = ldc(r, c + 4u, L);
  const bool pkix = (e - c == 8u) & (w0 == 0x0501062bu) & ((w1 & 0x8080ffffu) == 0x00000705u);
  return pkix || oid_arcs_ok(r, L, c, e);
}
template <class R>
CTMR_HD void ext_body_check(R& r, uint32_t L, uint32_t kind, uint32_t cv, uint32_t ev, bool& ok) {
  uint32_t t, c, ce;
  r.touch(cv, ev - cv < 200u ? ev - cv : 200u);
  rd_hdr_q(r, L, cv, ev, ok, t, c, ce);
  ok = ok & (ce == ev);  // "x509: trailing data after X.509 …"
  if (kind == 1u) {
    ok = ok & (t == 0x03u);
    bit_string_check(r, L, c, ce - c, ok);
  } else if (kind == 2u) {
    ok = ok & (t == 0x04u);
  } else {
    ok = ok & (t == 0x30u);
    if (kind == 4u) {
      if (ok & (c < ce)) {
        uint32_t tf, cf, ef;
        rd_hdr_q<false>(r, L, c, ce, ok, tf, cf, ef);
        ok = ok & ((tf != 0x80u) | (ef <= ce));  // the keyIdentifier itself must fit; another element is skipped unseen
      }
    } else {
      uint32_t p = c;
      if (kind == 6u) ok = ok & (c < ce);  // CT-go: "x509: empty AuthorityInfoAccess / SubjectInfoAccess extension" (recalled; round 6)
      while (ok & (p < ce)) {  // SEQUENCE OF
        uint32_t te, x, xe;
        r.touch(p, 32);
        rd_hdr_q(r, L, p, ce, ok, te, x, xe);
        if (kind == 3u) {
          ok = ok & (te == 0x06u);
          ok = ok && oid_arcs_ok_pkix(r, L, x, xe);
        } else {
          uint32_t to, co, eo;
          ok = ok & (te == 0x30u);
          rd_hdr_q(r, L, x, xe, ok, to, co, eo);
          ok = ok & (to == 0x06u);
          ok = ok && oid_arcs_ok_pkix(r, L, co, eo);
          if (kind == 6u) {  // accessLocation: asn1.RawValue, not optional
            uint32_t tl, cl, el;
            r.touch(eo, 8);
            rd_hdr_q(r, L, eo, xe, ok, tl, cl, el);
          }
        }
        p = xe;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// strict_extensions, round 5: the three extensions whose VALUE Go parses with code of its own, and CT-go's embedded SCT
// list.  Recalled from Go 1.13's crypto/x509, net/url, net and golang.org/x/crypto/cryptobyte (go.mod:24) and from
// certificate-transparency-go v1.1.0's fork of crypto/x509 — not verifiable on this machine:
//   subjectAltName 2.5.29.17 (parseSANExtension → forEachSAN): asn1.Unmarshal(value, &seq RawValue) — one element that
//     fits, nothing behind it ("trailing data"), a universal constructed SEQUENCE ("bad SAN sequence"); every element of it
//     a RawValue that fits, dispatched on v.Tag ALONE (the class is not looked at): 1 rfc822Name and 2 dNSName are kept as
//     they are (this toolchain checks no character set when it parses); 6 URI: url.Parse must succeed and, when the URL
//     has a host, domainToReverseLabels(host) — else fatal; 7 iPAddress: 4 or 16 octets, any other length is what CT-go
//     files as a NON-FATAL finding (the standard library fails); every other name form is ignored.
//   cRLDistributionPoints 2.5.29.31: asn1.Unmarshal(value, &[]distributionPoint), nothing behind it —
//     distributionPoint { DistributionPoint distributionPointName `optional,tag:0`; Reason BitString `optional,tag:1`;
//     CRLIssuer RawValue `optional,tag:2` }, distributionPointName { FullName []RawValue `optional,tag:0`; RelativeName
//     pkix.RDNSequence `optional,tag:1` } by encoding/asn1's struct rules (crl_dps below).
//   nameConstraints 2.5.29.30 (parseNameConstraintsExtension, cryptobyte): SEQUENCE { [0] permitted, [1] excluded } filling
//     the value, in that order, nothing else; not both absent or empty; every subtree SEQUENCE { base, … }: [2] dNSName IA5
//     and, without one leading '.', domainToReverseLabels; [7] iPAddress 8 or 32 octets, the second half a contiguous mask;
//     [1] rfc822Name IA5 and parseRFC2821Mailbox when it holds an '@', else as a domain; [6] URI IA5, not net.ParseIP, as a
//     domain; other forms ignored.
//   embedded SCT list 1.3.6.1.4.1.11129.2.4.2 (CT-go only): an OCTET STRING holding the TLS vector
//     SignedCertificateTimestampList — every failure an nfe.AddError: NON-FATAL.
//
// Where the work runs.  The structural part (headers, tag dispatch, lengths) is a few instructions per element and runs
// wherever the walk runs.  The parts that read an element's CONTENTS octet by octet — a URI, a name constraint, a
// nameRelativeToCRLIssuer, an SCT list — are rare in certificates of the public logs and would put their loops, and their
// registers, into the map kernel's window walk: a reader whose ld4 is served by its LDS window alone (WinReaderS) is told
// defer_exact() instead, and the kernel repeats that one certificate with the exact global-memory reader, as it does for a
// walk that left the window (kernels/reduce.h).
template <class R, class = void>
struct has_defer_exact : std::false_type {};
template <class R>
struct has_defer_exact<R, std::void_t<decltype(std::declval<R&>().defer_exact())>> : std::true_type {};

// octet i of the element [base, …) — the content readers below address their string from 0
template <class R>
struct Octets {
  const R& r;
  uint32_t L, base;
  CTMR_HD uint32_t operator[](uint32_t i) const { return ldc(r, base + i, L) & 0xffu; }
};

// x509.domainToReverseLabels(s).ok, fed octet by octet: no empty label (no trailing dot, no ".."), every rune in 33..126 (an
// octet >= 0x80 is, or decodes to, a rune above 126).  No octets at all: ok.
struct LabelCheck {
  uint32_t n = 0u, prev = 0u;
  bool bad = false;
  CTMR_HD void feed(uint32_t b) {
    bad = bad | (b < 33u) | (b > 126u) | ((n != 0u) & (prev == 0x2eu) & (b == 0x2eu));
    prev = b;
    n++;
  }
  // (one leading dot passes: the loop cuts labels off the END and never records the empty one in front — the release the
  //  reference builds with predates the fix that appends it)
  CTMR_HD bool ok() const { return (n == 0u) | (!bad & (prev != 0x2eu)); }
};
CTMR_HD bool is_hex_octet(uint32_t c) { return ((c - 0x30u) <= 9u) | (((c | 0x20u) - 0x61u) <= 5u); }
CTMR_HD uint32_t unhex_octet(uint32_t c) { return c <= 0x39u ? c - 0x30u : (c | 0x20u) - 0x61u + 10u; }
CTMR_HD bool is_alnum_octet(uint32_t c) { return ((c - 0x30u) <= 9u) | (((c | 0x20u) - 0x61u) <= 25u); }
// 1 << (c - 32) for the octets 32..95 / 96..127 that a set holds
CTMR_HD bool in_set(uint32_t c, unsigned long long lo_32_95, uint32_t hi_96_127) {
  return (((c - 32u) < 64u) & (((lo_32_95 >> ((c - 32u) & 63u)) & 1ull) != 0ull)) |
         (((c - 96u) < 32u) & (((hi_96_127 >> ((c - 96u) & 31u)) & 1u) != 0u));
}
// net/url shouldEscape(c, encodeHost): letters and digits pass, and  ! $ & ' ( ) * + , ; = : [ ] < > " - _ . ~
CTMR_HD bool url_host_should_escape(uint32_t c) {
  // 32..95:  ! " $ & ' ( ) * + , - . : ; < = > [ ] _      96..127: ~
  constexpr unsigned long long LO = (1ull << 1) | (1ull << 2) | (1ull << 4) | (1ull << 6) | (1ull << 7) | (1ull << 8) | (1ull << 9) |
                                    (1ull << 10) | (1ull << 11) | (1ull << 12) | (1ull << 13) | (1ull << 14) | (1ull << 26) |
                                    (1ull << 27) | (1ull << 28) | (1ull << 29) | (1ull << 30) | (1ull << 59) | (1ull << 61) | (1ull << 63);
  return !(is_alnum_octet(c) | in_set(c, LO, 1u << 30));
}
// net/url unescape(s[lo, hi), mode) as far as it can fail; mode 0 = path / fragment / userinfo (escapes only), 1 = host,
// 2 = zone.  lc (host, zone): the unescaped octets go to the label check.
template <class B>
CTMR_HD bool url_unescape_ok(const B& s, uint32_t lo, uint32_t hi, uint32_t mode, LabelCheck& lc) {
  bool good = true;
  uint32_t i = lo;
  while (good & (i < hi)) {
    const uint32_t c = s[i];
    if (c == 0x25u) {  // '%'
      const uint32_t h1 = i + 1u < hi ? s[i + 1u] : 0u, h2 = i + 2u < hi ? s[i + 2u] : 0u;
      good = (i + 2u < hi) & is_hex_octet(h1) & is_hex_octet(h2);
      const bool is25 = (h1 == 0x32u) & (h2 == 0x35u);
      const uint32_t v = (unhex_octet(h1) << 4) | unhex_octet(h2);
      if (mode == 1u) good = good & ((unhex_octet(h1) >= 8u) | is25);
      if (mode == 2u) good = good & (is25 | (v == 0x20u) | !url_host_should_escape(v));
      if (mode) lc.feed(v);
      i += 3u;
    } else {
      if (mode) {
        good = good & ((c == 0x2bu) | (c >= 0x80u) | !url_host_should_escape(c));
        lc.feed(c);
      }
      i++;
    }
  }
  return good;
}
template <class B>
CTMR_HD uint32_t octet_index(const B& s, uint32_t lo, uint32_t hi, uint32_t c) {  // first c in [lo, hi), else hi
  uint32_t i = lo;
  while ((i < hi) && (s[i] != c)) i++;
  return i;
}
template <class B>
CTMR_HD uint32_t octet_last_index(const B& s, uint32_t lo, uint32_t hi, uint32_t c) {  // last c in [lo, hi), else hi
  uint32_t at = hi;
  for (uint32_t i = lo; i < hi; i++) at = s[i] == c ? i : at;
  return at;
}
template <class B>
CTMR_HD bool url_port_ok(const B& s, uint32_t lo, uint32_t hi) {  // validOptionalPort
  bool good = (lo == hi) || (s[lo] == 0x3au);
  for (uint32_t i = lo + 1u; good & (i < hi); i++) good = (s[i] - 0x30u) <= 9u;
  return good;
}
// url.Parse(s[0, n)) succeeds and the URL's host, when it has one, passes domainToReverseLabels (parseSANExtension,
// nameTypeURI).  Go 1.13's net/url: the control-character test, getscheme, the opaque form, "first path segment in URL cannot
// contain colon", parseAuthority / parseHost (IP-literal brackets, the RFC 6874 zone, the port), validUserinfo and the
// escapes of userinfo, path and fragment; the query is not looked at.
template <class B>
CTMR_HD bool san_uri_ok(const B& s, uint32_t n) {
  const uint32_t un = octet_index(s, 0u, n, 0x23u);  // '#': u = s[:un], fragment behind it
  bool good = true;
  for (uint32_t i = 0; good & (i < un); i++) good = (s[i] >= 0x20u) & (s[i] != 0x7fu);
  LabelCheck host, none;
  const bool star = (un == 1u) && (s[0] == 0x2au);
  if (good & !star) {
    uint32_t rest = 0u;
    bool scheme = false;
    {  // getscheme
      uint32_t i = 0u;
      bool more = true;
      while (more & (i < un)) {
        const uint32_t c = s[i];
        const bool letter = ((c | 0x20u) - 0x61u) <= 25u;
        const bool mark = ((c - 0x30u) <= 9u) | (c == 0x2bu) | (c == 0x2du) | (c == 0x2eu);
        if (letter | (mark & (i != 0u))) {
          i++;
        } else {
          if (c == 0x3au) {
            good = i != 0u;  // "missing protocol scheme"
            scheme = i != 0u;
            rest = i != 0u ? i + 1u : 0u;
          }
          more = false;
        }
      }
    }
    uint32_t re = octet_index(s, rest, un, 0x3fu);  // the query goes, whatever its form
    const bool rooted = (rest < re) && (s[rest] == 0x2fu);
    bool opaque = false;
    if (good & !rooted) {
      opaque = scheme;
      if (!scheme) {
        const uint32_t colon = octet_index(s, rest, re, 0x3au), slash = octet_index(s, rest, re, 0x2fu);
        good = !((colon < re) & ((slash == re) | (colon < slash)));
      }
    }
    if (good & !opaque) {
      const bool two = (re - rest >= 2u) && (s[rest] == 0x2fu) && (s[rest + 1u] == 0x2fu);
      const bool three = two && (re - rest >= 3u) && (s[rest + 2u] == 0x2fu);
      if (two & (scheme | !three)) {  // an authority
        const uint32_t a0 = rest + 2u, a1 = octet_index(s, a0, re, 0x2fu);
        const uint32_t at = octet_last_index(s, a0, a1, 0x40u);  // '@'
        const uint32_t h0 = at < a1 ? at + 1u : a0;
        // parseHost
        if ((h0 < a1) && (s[h0] == 0x5bu)) {  // '[': an IP literal
          const uint32_t rb = octet_last_index(s, h0, a1, 0x5du);
          good = (rb < a1) && url_port_ok(s, rb + 1u, a1);
          if (good) {
            uint32_t z = h0;  // "%25" in front of the bracket: a zone
            bool found = false;
            while (!found & (z + 3u <= rb)) {
              found = (s[z] == 0x25u) && (s[z + 1u] == 0x32u) && (s[z + 2u] == 0x35u);
              z += found ? 0u : 1u;
            }
            if (found) good = url_unescape_ok(s, h0, z, 1u, host) && url_unescape_ok(s, z, rb, 2u, host) && url_unescape_ok(s, rb, a1, 1u, host);
            else good = url_unescape_ok(s, h0, a1, 1u, host);
          }
        } else {
          const uint32_t cl = octet_last_index(s, h0, a1, 0x3au);
          good = (cl == a1) || url_port_ok(s, cl, a1);
          good = good && url_unescape_ok(s, h0, a1, 1u, host);
        }
        if (good & (at < a1)) {  // validUserinfo, then the escapes of user and password
          // 32..95: ! $ & ' ( ) * + , - . : ; = @ _     96..127: ~     (and '%', letters, digits)
          constexpr unsigned long long UI = (1ull << 1) | (1ull << 4) | (1ull << 5) | (1ull << 6) | (1ull << 7) | (1ull << 8) | (1ull << 9) |
                                            (1ull << 10) | (1ull << 11) | (1ull << 12) | (1ull << 13) | (1ull << 14) | (1ull << 26) |
                                            (1ull << 27) | (1ull << 29) | (1ull << 32) | (1ull << 63);
          for (uint32_t i = a0; good & (i < at); i++) good = is_alnum_octet(s[i]) | in_set(s[i], UI, 1u << 30);
          good = good && url_unescape_ok(s, a0, at, 0u, none);
        }
        rest = a1;
      }
      good = good && url_unescape_ok(s, rest, re, 0u, none);  // setPath
    }
  }
  if (good & (un < n)) good = url_unescape_ok(s, un + 1u, n, 0u, none);  // the fragment
  return good && host.ok();
}

// subjectAltName: [cv, ev) = the extension's value.  nf: an iPAddress of another length than 4 or 16 (CT-go: non-fatal).
template <class R>
CTMR_HD void ext_san_check(R& r, uint32_t L, uint32_t cv, uint32_t ev, bool& ok, uint32_t& nf) {
  uint32_t t, c, ce;
  r.touch(cv, 48);
  rd_hdr(r, L, cv, ev, ok, t, c, ce);
  ok = ok & (ce == ev) & (t == 0x30u);  // one element, nothing behind it, a universal constructed SEQUENCE
  uint32_t p = c;
  while (ok & (p < ev)) {  // GeneralNames: every element a RawValue that fits; dispatch on the tag NUMBER alone
    r.touch(p, 16);
    const uint32_t w = ldc(r, p, L);
    const uint32_t tg = w & 0xffu, lb = (w >> 8) & 0xffu;
    uint32_t x, xe;
    if (((tg & 0x1fu) != 0x1fu) & (lb < 0x80u)) {  // short form (every dNSName of the Web PKI): no second read
      x = p + 2u;
      xe = x + lb;
      ok = ok & (xe <= ev);
    } else {
      uint32_t t2;
      rd_hdr(r, L, p, ev, ok, t2, x, xe);
    }
    const uint32_t tn = tg & 0x1fu;  // (0x1f: the high-tag-number form — a number >= 31, no case below)
    nf = (ok & (tn == 7u) & (xe - x != 4u) & (xe - x != 16u)) ? (nf | WALK_NF_EXT) : nf;
    if (ok & (tn == 6u)) {
      if constexpr (has_defer_exact<R>::value) r.defer_exact();
      else ok = san_uri_ok(Octets<R>{r, L, x}, xe - x);
    }
    p = xe;
  }
}

// The same walk for readers with a wave-cooperative refill (the map kernels' LDS windows), called BEHIND the extension loop,
// where the lanes of the wave are together again: a subjectAltName is 0.5–1 KB, two to four windows — met inside the loop,
// every lane refilled its own window whenever IT ran out (≈ 14 refill events of 16 loads per wave, measured: VMEM_RD 42 →
// 265 instructions per wave, round 5's first build).  Here all lanes still walking refill together, 16 lanes per
// certificate, once per round; between refills a lane hops from header to header inside its window.  [cv, ev) = the
// value (ev = 0: this certificate has no subjectAltName).
template <class R, class = void>
struct has_coop_refill : std::false_type {};
template <class R>
struct has_coop_refill<R, std::void_t<decltype(std::declval<R&>().coop_refill(0u, false))>> : std::true_type {};
template <class R>
CTMR_HD void ext_san_coop(R& r, uint32_t L, uint32_t cv, uint32_t ev, bool& ok, uint32_t& nf) {
  if (!R::whole_wave()) {  // the batch's last wave: lane by lane
    if (cv < ev) ext_san_check(r, L, cv, ev, ok, nf);
    return;
  }
  // Round 6.  (a) A lane goes on in the window it has — the extension window usually still holds the head of the
  // subjectAltName — and takes part in a refill only when the next header lies outside (round 5 refilled every lane at the
  // start of every round).  (b) A refill brings the two 128-byte LINES from the one the next header lies in
  // (coop_refill_lines): the windows of successive rounds do not overlap, so every line of the subjectAltName crosses the
  // fabric once (round 5: 256 bytes from a 16-byte boundary = three lines, the third one again in the next round — 2 047
  // bytes of traffic per 1 523-byte certificate).  (c) The elements of a window are hopped SPECULATIVELY as short-form
  // TLVs — two octets read, p += 2 + length, tags and length octets OR-ed into two masks: five vector instructions per
  // element — and only a window in which a URI, an iPAddress, a high tag number or a long-form length turned up (the
  // masks say so; the hops behind such an element are garbage, and harmless: p only grows and stops at the window's or
  // the value's end) is walked again, element by element, by round 5's loop.
  bool act = ok & (cv < ev), first = true, again = false;
  uint32_t p = cv;
  for (;;) {
    const bool want = act & (p < ev);
    if (!R::any_lane(want)) break;  // (wave-uniform)
    const bool need = want & (again | !r.holds(p, 2u));
    if (R::any_lane(need)) r.coop_refill_lines_to(p, need, ev);  // (what is left of the value, not a whole window)
    again = false;
    if (want) {
      if (first) {  // one element filling the value: a universal constructed SEQUENCE
        if (r.holds(p, 12u)) {
          uint32_t t, c, ce;
          rd_hdr(r, L, cv, ev, ok, t, c, ce);
          ok = ok & (ce == ev) & (t == 0x30u);
          p = c;
          first = false;
        } else {
          again = true;  // the header straddles the window's end: the lines from p, next round
        }
      }
      if (!first & ok) {
        const uint32_t p0 = p, we = r.wend() - 1u, stop = ev < we ? ev : we;
        uint32_t seen = 0u, lens = 0u;
        while (p < stop) {
          const uint32_t w = r.ld2(p);
          seen |= 1u << (w & 31u);
          lens |= w;
          p += 2u + (w >> 8);
        }
        if (((seen & 0x800000c0u) | (lens & 0x8000u)) != 0u) {  // tag numbers 6, 7, 31, or a length octet >= 0x80
          p = p0;
          while (ok & (p < ev) && r.holds(p, 16u)) {
            const uint32_t w = r.ld4(p);
            const uint32_t tg = w & 0xffu, lb = (w >> 8) & 0xffu;
            uint32_t x, xe;
            if (((tg & 0x1fu) != 0x1fu) & (lb < 0x80u)) {
              x = p + 2u;
              xe = x + lb;
              ok = ok & (xe <= ev);
            } else {
              uint32_t t2;
              rd_hdr(r, L, p, ev, ok, t2, x, xe);
            }
            const uint32_t tn = tg & 0x1fu;
            nf = (ok & (tn == 7u) & (xe - x != 4u) & (xe - x != 16u)) ? (nf | WALK_NF_EXT) : nf;
            if (ok & (tn == 6u)) {  // a URI's contents: the exact reader's business (see above) — or a reader that reaches anywhere
              if constexpr (has_defer_exact<R>::value) r.defer_exact();
              else ok = san_uri_ok(Octets<R>{r, L, x}, xe - x);
            }
            p = xe;
          }
          again = ok & (p < ev);  // stopped short of the window's end (a header needs up to 16 octets): the lines from p
        } else {
          ok = ok & (p <= ev);    // the last element must fit ("data truncated")
        }
      }
      act = ok;
    }
  }
}

// cRLDistributionPoints: [cv, ev) = the extension's value, []distributionPoint by encoding/asn1's struct rules — fields in
// order, each preceded by a header that must parse unless the contents are used up, a field of another tag skipped, a
// matching field must fit, what follows the last field ignored.
//   COLLECT  the FullName elements with tag NUMBER 6 go to uo/ul (certificate offsets / lengths; nu counts all of them):
//            parseCertificate's CRLDistributionPoints, what IssuerMetadata.Accumulate reads (kernels/meta*.h)
//   DEEP     a nameRelativeToCRLIssuer's contents are parsed as well (RelativeDistinguishedNames: walk_rdns) — the
//            certificate check; the metadata side only needs to know where the URIs are
template <bool COLLECT, bool DEEP, bool STRINGS, uint32_t MAXU, class R>
CTMR_HD void crl_dps(R& r, uint32_t L, uint32_t cv, uint32_t ev, bool& ok, uint32_t (&uo)[MAXU], uint32_t (&ul)[MAXU],
                     uint32_t& nu, uint32_t& nf, bool strings) {
  uint32_t t, p, pe;
  rd_hdr_q(r, L, cv, ev, ok, t, p, pe);
  ok = ok & (t == 0x30u) & (pe == ev);
  while (ok & (p < ev)) {
    uint32_t td, off, end;
    rd_hdr_q(r, L, p, ev, ok, td, off, end);
    ok = ok & (td == 0x30u);
    uint32_t tf = 0u, fc = 0u, fe = 0u;
    if (ok & (off < end)) rd_hdr_q<false>(r, L, off, end, ok, tf, fc, fe);
    if (ok & (off < end) & (tf == 0xa0u)) {  // DistributionPoint distributionPointName `optional,tag:0`
      ok = ok & (fe <= end);
      uint32_t n = fc, tg = 0u, gc = 0u, ge = 0u;
      const uint32_t n_end = fe;
      if (ok & (n < n_end)) rd_hdr_q<false>(r, L, n, n_end, ok, tg, gc, ge);
      if (ok & (n < n_end) & (tg == 0xa0u)) {  // FullName []asn1.RawValue `optional,tag:0`
        ok = ok & (ge <= n_end);
        uint32_t q = gc;
        while (ok & (q < ge)) {
          uint32_t tn, u, ue;
          rd_hdr_q(r, L, q, ge, ok, tn, u, ue);
          if constexpr (COLLECT) {
            if (ok & ((tn & 0x1fu) == 6u)) {
#pragma unroll
              for (uint32_t k = 0; k < MAXU; k++) {  // register arrays: no dynamic indexing
                uo[k] = k == nu ? u : uo[k];
                ul[k] = k == nu ? ue - u : ul[k];
              }
              nu++;
            }
          }
          q = ue;
        }
        n = ge;
        tg = 0u;
        if (ok & (n < n_end)) rd_hdr_q<false>(r, L, n, n_end, ok, tg, gc, ge);
      }
      if (ok & (n < n_end) & (tg == 0xa1u)) {  // RelativeName pkix.RDNSequence `optional,tag:1`
        ok = ok & (ge <= n_end);
        if constexpr (DEEP) {
          if (ok) {
            if constexpr (has_defer_exact<R>::value) {
              r.defer_exact();
            } else {
              uint32_t nfr = 0u;
              walk_rdns<STRINGS>(r, L, gc, ge, ok, nfr, strings);
              nf |= (nfr & WALK_NF_LAX_INTEGER) ? WALK_NF_EXT : 0u;
              nf |= nfr & WALK_NF_STRING;
            }
          }
        }
      }
      off = fe;
      tf = 0u;
      if (ok & (off < end)) rd_hdr_q<false>(r, L, off, end, ok, tf, fc, fe);
    }
    if (ok & (off < end) & (tf == 0x81u)) {  // Reason asn1.BitString `optional,tag:1`
      ok = ok & (fe <= end);
      bit_string_check(r, L, fc, fe - fc, ok);
      off = fe;
      tf = 0u;
      if (ok & (off < end)) rd_hdr_q<false>(r, L, off, end, ok, tf, fc, fe);
    }
    if (ok & (off < end) & ((tf == 0x82u) | (tf == 0xa2u))) ok = fe <= end;  // CRLIssuer asn1.RawValue `optional,tag:2`
    p = end;
  }
}

// nameConstraints, read the way golang.org/x/crypto/cryptobyte reads it (a tag is the whole identifier octet, the
// high-tag-number form is refused, an optional element is recognised by its first octet alone).  Contents only: callers
// whose reader defers have done so.
template <class B>
CTMR_HD bool nc_ip_mask_ok(const B& s, uint32_t lo, uint32_t hi) {  // isValidIPMask: ones, then zeros
  bool good = true, zero = false;
  for (uint32_t i = lo; good & (i < hi); i++) {
    const uint32_t b = s[i], inv = ~b & 0xffu;
    good = zero ? (b == 0u) : ((inv & (inv + 1u)) == 0u);
    zero = zero | (b != 0xffu);
  }
  return good;
}
template <class B>
CTMR_HD bool labels_ok(const B& s, uint32_t lo, uint32_t hi) {
  LabelCheck lc;
  for (uint32_t i = lo; i < hi; i++) lc.feed(s[i]);
  return lc.ok();
}
template <class B>
CTMR_HD bool go_dtoi(const B& s, uint32_t& i, uint32_t hi, uint32_t& v) {  // net.dtoi: digits, value below 0xFFFFFF, at least one
  const uint32_t i0 = i;
  bool good = true;
  v = 0u;
  while (good & (i < hi) && ((s[i] - 0x30u) <= 9u)) {
    v = v * 10u + (s[i] - 0x30u);
    good = v < 0xFFFFFFu;
    i += good ? 1u : 0u;
  }
  return good & (i != i0);
}
template <class B>
CTMR_HD bool go_parse_ipv4(const B& s, uint32_t lo, uint32_t hi) {  // net.parseIPv4 of go1.13 (leading zeros are fine)
  bool good = true;
  uint32_t i = lo;
  for (uint32_t k = 0; good & (k < 4u); k++) {
    good = i < hi;
    if (good & (k > 0u)) {
      good = s[i] == 0x2eu;
      i++;
    }
    uint32_t v = 0u;
    good = good && go_dtoi(s, i, hi, v) && (v <= 0xffu);
  }
  return good & (i == hi);
}
template <class B>
CTMR_HD bool go_parse_ipv6(const B& s, uint32_t lo, uint32_t hi) {  // net.parseIPv6 of go1.13 (no zone)
  int ellipsis = -1;
  uint32_t i = lo, at = 0u;  // at: octets of the address filled so far
  if ((hi - lo >= 2u) && (s[lo] == 0x3au) && (s[lo + 1u] == 0x3au)) {
    ellipsis = 0;
    i = lo + 2u;
    if (i == hi) return true;
  }
  bool good = true, more = true;
  while (good & more & (at < 16u)) {
    uint32_t v = 0u, c = i;  // xtoi
    while (good & (c < hi) && is_hex_octet(s[c])) {
      v = v * 16u + unhex_octet(s[c]);
      good = v < 0xFFFFFFu;
      c += good ? 1u : 0u;
    }
    good = good & (c != i) & (v <= 0xffffu);
    if (!good) break;
    if ((c < hi) && (s[c] == 0x2eu)) {  // a trailing dotted quad
      good = !((ellipsis < 0) & (at != 12u)) & (at + 4u <= 16u) && go_parse_ipv4(s, i, hi);
      i = hi;
      at += 4u;
      more = false;
    } else {
      at += 2u;
      i = c;
      if (i == hi) {
        more = false;
      } else {
        good = (s[i] == 0x3au) & (hi - i != 1u);
        i++;
        if (good && (s[i] == 0x3au)) {
          good = ellipsis < 0;
          ellipsis = (int)at;
          i++;
          more = i != hi;
        }
      }
    }
  }
  return good & (i == hi) & (at < 16u ? ellipsis >= 0 : ellipsis < 0);
}
template <class B>
CTMR_HD bool go_parse_ip(const B& s, uint32_t lo, uint32_t hi) {  // net.ParseIP(s) != nil
  for (uint32_t i = lo; i < hi; i++) {
    if (s[i] == 0x2eu) return go_parse_ipv4(s, lo, hi);
    if (s[i] == 0x3au) return go_parse_ipv6(s, lo, hi);
  }
  return false;
}
template <class B>
CTMR_HD bool go_mailbox_ok(const B& s, uint32_t lo, uint32_t hi) {  // x509.parseRFC2821Mailbox(s).ok
  if (lo == hi) return false;
  uint32_t i = lo;
  bool good = true;
  if (s[lo] == 0x22u) {  // quoted-string
    i = lo + 1u;
    bool open = true;
    while (good & open) {
      good = i < hi;
      if (!good) break;
      const uint32_t c = s[i++];
      if (c == 0x22u) {
        open = false;
      } else if (c == 0x5cu) {  // quoted-pair
        const uint32_t e = i < hi ? s[i] : 0u;
        good = (i < hi) & ((e == 11u) | (e == 12u) | ((e - 1u) <= 8u) | ((e - 14u) <= 113u));
        i++;
      } else {  // qtext (with the space RFC 3696's example needs)
        good = (c == 11u) | (c == 12u) | (c == 32u) | (c == 33u) | (c == 127u) | ((c - 1u) <= 7u) | ((c - 14u) <= 17u) |
               ((c - 35u) <= 56u) | ((c - 93u) <= 33u);
      }
    }
  } else {  // Atom ("." Atom)*, backslash escapes accepted outside quotes as well
    // 32..95: ! # $ % & ' * + - . / = ? ^ _     96..127: ` { | } ~
    constexpr unsigned long long AT = (1ull << 1) | (1ull << 3) | (1ull << 4) | (1ull << 5) | (1ull << 6) | (1ull << 7) | (1ull << 10) |
                                      (1ull << 11) | (1ull << 13) | (1ull << 14) | (1ull << 15) | (1ull << 29) | (1ull << 31) |
                                      (1ull << 62) | (1ull << 63);
    constexpr uint32_t AT_HI = (1u << 0) | (1u << 27) | (1u << 28) | (1u << 29) | (1u << 30);
    uint32_t nlocal = 0u, first = 0u, last = 0u;
    bool dots = false, more = true;
    while (good & more & (i < hi)) {
      const uint32_t c = s[i];
      if (c == 0x5cu) {
        i++;
        good = i < hi;
      } else if (!(is_alnum_octet(c) | in_set(c, AT, AT_HI))) {
        more = false;
      }
      if (good & more) {
        const uint32_t b = s[i++];  // after a backslash: the escaped octet, whatever it is
        first = nlocal == 0u ? b : first;
        dots = dots | ((nlocal != 0u) & (last == 0x2eu) & (b == 0x2eu));
        last = b;
        nlocal++;
      }
    }
    good = good & (nlocal != 0u) & (first != 0x2eu) & (last != 0x2eu) & !dots;
  }
  good = good && (i < hi) && (s[i] == 0x40u);
  return good && labels_ok(s, i + 1u, hi);
}
template <class R>
CTMR_HD void nc_subtrees(const R& r, uint32_t L, uint32_t p, uint32_t end, bool& ok) {
  while (ok & (p < end)) {
    uint32_t t, c, ce, tv, v, ve;
    rd_hdr(r, L, p, end, ok, t, c, ce);           // ReadASN1(&seq, SEQUENCE)
    ok = ok & (t == 0x30u);
    rd_hdr(r, L, c, ce, ok, tv, v, ve);           // seq.ReadAnyASN1(&value, &tag); minimum / maximum are not read
    ok = ok & ((tv & 0x1fu) != 0x1fu);
    if (ok) {
      const Octets<R> s{r, L, 0u};
      bool ia5 = true;
      if ((tv == 0x82u) | (tv == 0x81u) | (tv == 0x86u))
        for (uint32_t i = v; ia5 & (i < ve); i++) ia5 = s[i] < 0x80u;
      const uint32_t v1 = ((v < ve) && (s[v] == 0x2eu)) ? v + 1u : v;  // one leading '.' is the constraint's own syntax
      if (tv == 0x82u) {
        ok = ia5 && labels_ok(s, v1, ve);
      } else if (tv == 0x87u) {
        const uint32_t n = ve - v;
        ok = ((n == 8u) | (n == 32u)) && nc_ip_mask_ok(s, v + n / 2u, ve);
      } else if (tv == 0x81u) {
        ok = ia5 && (octet_index(s, v, ve, 0x40u) < ve ? go_mailbox_ok(s, v, ve) : labels_ok(s, v1, ve));
      } else if (tv == 0x86u) {
        ok = ia5 && !go_parse_ip(s, v, ve) && labels_ok(s, v1, ve);
      }
    }
    p = ce;
  }
}
template <class R>
CTMR_HD void ext_nc_check(const R& r, uint32_t L, uint32_t cv, uint32_t ev, bool& ok) {
  uint32_t t, p, pe;
  ok = ok & (ev - cv >= 2u);
  rd_hdr(r, L, cv, ev, ok, t, p, pe);             // outer.ReadASN1(&toplevel, SEQUENCE) && outer.Empty()
  ok = ok & (t == 0x30u) & (pe == ev);
  bool have_p = false, have_e = false;
  uint32_t ps = 0u, pn = 0u, es = 0u, en = 0u, tf, fc, fe;
  if (ok & (p < ev) && ((ldc(r, p, L) & 0xffu) == 0xa0u)) {
    ok = ok & (ev - p >= 2u);
    rd_hdr(r, L, p, ev, ok, tf, fc, fe);
    have_p = true; ps = fc; pn = fe; p = fe;
  }
  if (ok & (p < ev) && ((ldc(r, p, L) & 0xffu) == 0xa1u)) {
    ok = ok & (ev - p >= 2u);
    rd_hdr(r, L, p, ev, ok, tf, fc, fe);
    have_e = true; es = fc; en = fe; p = fe;
  }
  ok = ok & (p == ev);                             // toplevel.Empty()
  ok = ok & (have_p | have_e) & ((pn != ps) | (en != es));  // "x509: empty name constraints extension"
  if (have_p) nc_subtrees(r, L, ps, pn, ok);
  if (have_e) nc_subtrees(r, L, es, en, ok);
}

// CT-go's embedded SCT list: OCTET STRING { opaque list<1..2^16-1> of opaque sct<1..2^16-1> }, nothing left over anywhere
template <class R>
CTMR_HD bool ext_sct_ok(const R& r, uint32_t L, uint32_t cv, uint32_t ev) {
  bool good = true;
  uint32_t t, c, ce;
  rd_hdr(r, L, cv, ev, good, t, c, ce);
  good = good & (t == 0x04u) & (ce == ev) & (ev - c >= 2u);
  const uint32_t ll = __builtin_bswap32(ldc(r, c, L)) >> 16;
  uint32_t p = c + 2u;
  good = good && (ll >= 1u) && (p + ll == ev);
  while (good & (p < ev)) {
    const uint32_t sl = __builtin_bswap32(ldc(r, p, L)) >> 16;
    good = (ev - p >= 2u) && (sl >= 1u) && (p + 2u + sl <= ev);
    p += 2u + sl;
  }
  return good;
}

// RFC 3779 (round 6; certificate-transparency-go ONLY: x509/rpki.go, go.mod:10 — recalled, not verifiable here).
// parseRPKIAddrBlocks / parseRPKIASIdentifiers decode the value with strict asn1.Unmarshal calls and file EVERY failure as a
// non-fatal error: an X509 entry keeps its certificate, a precertificate and a Chain[0] issuer are dropped.
//   sbgp-ipAddrBlock 1.3.6.1.5.5.7.1.7: []ipAddressFamily { AddressFamily []byte; Choice asn1.RawValue } filling the value;
//     AddressFamily 2 or 3 octets; Choice == 05 00 (inherit), or else []asn1.RawValue whose elements are, by tag NUMBER
//     alone, 3 → asn1.BitString (universal, primitive, parseBitString) or 16 → struct { Min, Max asn1.BitString }; any other
//     tag number is a finding.
//   sbgp-autonomousSysNum 1.3.6.1.5.5.7.1.8: struct { ASNum RawValue `optional,tag:0`; RDI RawValue `optional,tag:1` }
//     filling the value; a present choice's CONTENTS are 05 00 or one []asn1.RawValue filling them whose elements are, by
//     tag number, 2 → int (universal primitive INTEGER, minimal, at most 8 octets) or 16 → struct { Min, Max int }.
// (The test suite's checker restates the same rules independently.)  These extensions do not occur in the public logs'
// certificates: a window-only reader hands such a certificate to the exact one.
template <class R>
CTMR_HD bool rpki_bit_string(const R& r, uint32_t L, uint32_t p, uint32_t end, uint32_t& after) {
  bool good = true;
  uint32_t t, c, ce;
  rd_hdr(r, L, p, end, good, t, c, ce);
  good = good & (t == 0x03u);
  if (good) bit_string_check(r, L, c, ce - c, good);
  after = ce;
  return good;
}
template <class R>
CTMR_HD bool rpki_int(const R& r, uint32_t L, uint32_t p, uint32_t end, uint32_t& after) {  // `int`: parseInt64, strict
  bool good = true, neg;
  uint32_t t, c, ce, lax = 0u;
  rd_hdr(r, L, p, end, good, t, c, ce);
  good = good & (t == 0x02u);
  if (good) int_check(r, L, c, ce - c, good, lax, neg);
  after = ce;
  return good & (lax == 0u) & (ce - c <= 8u);
}
// every element of [p, end) is a TLV that fits (a []asn1.RawValue unmarshals)
template <class R>
CTMR_HD bool rpki_list_fits(const R& r, uint32_t L, uint32_t p, uint32_t end) {
  bool good = true;
  while (good & (p < end)) {
    uint32_t t, c, ce;
    rd_hdr(r, L, p, end, good, t, c, ce);
    p = ce;
  }
  return good;
}
template <class R>
CTMR_HD bool ext_ipaddr_ok(const R& r, uint32_t L, uint32_t cv, uint32_t ev) {
  bool good = true;
  uint32_t t, c0, ce0;
  rd_hdr(r, L, cv, ev, good, t, c0, ce0);
  good = good & (t == 0x30u) & (ce0 == ev);
  for (uint32_t q = c0; good & (q < ev);) {  // asn1.Unmarshal(data, &addrBlocks) as a whole
    uint32_t tf, x, xe, ta, a, ae, tc, h, he;
    rd_hdr(r, L, q, ev, good, tf, x, xe);
    good = good & (tf == 0x30u);
    rd_hdr(r, L, x, xe, good, ta, a, ae);
    good = good & (ta == 0x04u);
    rd_hdr(r, L, ae, xe, good, tc, h, he);
    q = xe;
  }
  for (uint32_t q = c0; good & (q < ev);) {  // the loop over the blocks (one finding is as good as many)
    bool hk = true;
    uint32_t tf, x, xe, ta, a, ae, tc, h, he;
    rd_hdr(r, L, q, ev, hk, tf, x, xe);
    rd_hdr(r, L, x, xe, hk, ta, a, ae);
    rd_hdr(r, L, ae, xe, hk, tc, h, he);
    q = xe;
    good = good & ((ae - a == 2u) | (ae - a == 3u));
    if ((tc == 0x05u) & (h == ae + 2u) & (he == h)) continue;  // asn1.NullBytes: inherit
    good = good & (tc == 0x30u) && rpki_list_fits(r, L, h, he);
    for (uint32_t y = h; good & (y < he);) {
      uint32_t te, e, ee, aft;
      rd_hdr(r, L, y, he, hk, te, e, ee);
      const uint32_t tn = te & 0x1fu;
      if (tn == 3u) good = rpki_bit_string(r, L, y, ee, aft);
      else if (tn == 16u) good = (te == 0x30u) && rpki_bit_string(r, L, e, ee, aft) && rpki_bit_string(r, L, aft, ee, aft);
      else good = false;
      y = ee;
    }
  }
  return good;
}
template <class R>
CTMR_HD bool rpki_asid_choice_ok(const R& r, uint32_t L, uint32_t c, uint32_t ce) {
  if ((ce - c == 2u) && ((ldc(r, c, L) & 0xffffu) == 0x0005u)) return true;  // 05 00: inherit
  bool good = true, hk = true;
  uint32_t t, h, he;
  rd_hdr(r, L, c, ce, good, t, h, he);
  good = good & (t == 0x30u) & (he == ce) && rpki_list_fits(r, L, h, he);
  for (uint32_t y = h; good & (y < he);) {
    uint32_t te, e, ee, aft;
    rd_hdr(r, L, y, he, hk, te, e, ee);
    const uint32_t tn = te & 0x1fu;
    if (tn == 2u) good = rpki_int(r, L, y, ee, aft);
    else if (tn == 16u) good = (te == 0x30u) && rpki_int(r, L, e, ee, aft) && rpki_int(r, L, aft, ee, aft);
    else good = false;
    y = ee;
  }
  return good;
}
template <class R>
CTMR_HD bool ext_asnum_ok(const R& r, uint32_t L, uint32_t cv, uint32_t ev) {
  bool good = true;
  uint32_t t, off, end;
  rd_hdr(r, L, cv, ev, good, t, off, end);
  good = good & (t == 0x30u) & (end == ev);
  bool found = true;
#pragma unroll
  for (uint32_t k = 0; k < 2u; k++) {  // ASNum `optional,tag:0`, RDI `optional,tag:1`: in order; another tag is skipped unconsumed
    if (good & (off < ev)) {
      uint32_t tf, fc, fe;
      rd_hdr<false>(r, L, off, ev, good, tf, fc, fe);
      if (good & ((tf == (0x80u | k)) | (tf == (0xa0u | k)))) {
        good = fe <= ev;
        if (good) found = found & rpki_asid_choice_ok(r, L, fc, fe);
        off = fe;
      }
    }
  }
  return good & found;
}

// `filter` may be null (no CN filter configured: cn_match = true).  r.touch(pos, need) tells a
// windowed reader that about `need` bytes from pos are read next; r.touch_tail(pos, tail) that the
// bytes from pos AND the bytes at `tail` are read next; other readers ignore both.
// The filter view is passed BY VALUE (use_filter + fv): a pointer that is either &local or null makes
// the compiler materialise the view in scratch memory — 28 bytes of HBM writes per certificate.
//
// The accept/reject rules are Go's encoding/asn1 struct-unmarshalling rules applied to crypto/x509's `certificate`,
// `tbsCertificate`, `publicKeyInfo`, `pkix.AlgorithmIdentifier`, `pkix.RDNSequence`, `pkix.Extension` and
// `basicConstraints` definitions, as far as the bytes are ones this path reads anyway (DESIGN.md §3.1 lists what is
// left out): a field must match its tag and lie inside the enclosing contents; bytes behind the last field of a
// SEQUENCE are ignored; an OPTIONAL field with another tag is skipped, but its header must parse; parsing resumes
// behind the INNER element of an EXPLICIT wrapper, whatever the wrapper's own length says.
// r.note_issuer(pos, len), for readers that have one
template <class R, class = void>
struct has_note_issuer : std::false_type {};
template <class R>
struct has_note_issuer<R, std::void_t<decltype(std::declval<R&>().note_issuer(0u, 0u))>> : std::true_type {};
template <class R>
CTMR_HD void note_issuer_if(R& r, uint32_t pos, uint32_t len) {
  if constexpr (has_note_issuer<R>::value) r.note_issuer(pos, len);
}

// TBS_ONLY: the buffer is a bare TBSCertificate (what a precertificate entry's MerkleTreeLeaf carries) — what CT-go's
// x509.ParseTBSCertificate accepts: the TBSCertificate SEQUENCE must fill the buffer ("trailing data" otherwise) and there
// is no signatureAlgorithm / signatureValue behind it; everything inside is parsed as for a certificate.
// NAMES_ONLY: stop behind the subject Name (round 3's pre-pass kernel stopped there; nothing instantiates it since round 4).
// spki: also parse the public key as CT-go's parsePublicKey does (spki_key.h; ctmr_set_strict_spki, on by default).
// EC_DEFER (the map kernels): an EC key's curve equation is not evaluated here — Walk.ec_* says what is owed (spki_key.h).
// strings: strict_strings inside the walk (round 4: the pre-pass of round 3 filled the same front window a second time,
// +11.5 ms per 100 M certificates) — inside walk_name, on each value where the walk meets it while the window holds it, the
// character sets are checked (value_strings_ok) and a violation is filed as WALK_NF_STRING.  The caller passes it only where the
// finding can matter (a precertificate, a Chain[0] issuer — an X509 entry keeps its certificate either way).
// STRINGS = false compiles the check out: the map kernels carry it in instantiations of their own (the code's mere presence
// cost the default kernel 0.5 ms per 100 M certificates, A/B on one box, round 4).
// Measurement builds (scripts/run.sh attribution; never shipped): -DCTMR_EXP_STOP_AFTER=k ends the walk behind stage k — the
// instruction stream up to there is the product's, the results are not.  1 serial  2 issuer Name  3 validity  4 subject Name
// 5 SubjectPublicKeyInfo and key  6 unique ids and extensions  (7 = the whole walk).
#ifdef CTMR_EXP_STOP_AFTER
#define CTMR_STAGE(k) do { if constexpr ((CTMR_EXP_STOP_AFTER) <= (k)) return ok; } while (0)
#else
#define CTMR_STAGE(k) do { } while (0)
#endif
template <class R, bool TBS_ONLY = false, bool NAMES_ONLY = false, bool EC_DEFER = false, bool STRINGS = true>
CTMR_HD bool walk_cert(R& r, uint32_t L, Walk& o, bool use_filter, const FilterView fv, bool spki = true, bool strings = false,
                       uint32_t ext_mode = 0u) {
  const bool ext = (ext_mode & WALK_EXT_ON) != 0u, ext_nf = (ext_mode & WALK_EXT_NF) != 0u;
  o.serial_off = o.serial_len = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) o.serial_w[k] = 0;
  o.cn_match = true;
  o.not_before = o.not_after = 0;
  o.cn_off = o.cn_len = 0;
  o.spki_off = o.spki_len = 0;
  o.bc_valid = o.is_ca = false;
  o.nonfatal = 0u;
  o.meta_issuer = o.meta_crl = META_NONE;
  o.issuer_name = o.subject_name = 0u;
  o.ec_curve = o.ec_pos = o.ec_shift = 0u;
  if constexpr (!STRINGS) strings = false;
  bool ok = L <= 0x7fffffffu;
  L = ok ? L : 0u;  // no early return: every lane of a wave stays on the same path (ok-accumulate)
  uint32_t tag, cs, ce;
  if constexpr (reader_has_head<R>(0)) {
    // (the first window begins behind these headers: they are read from the sixteen octets the lane holds in registers)
    const HeadView hv{r.hd[0], r.hd[1], r.hd[2], r.hd[3], !r.hd_ok};
    rd_hdr(hv, L, 0, L, ok, tag, cs, ce);
    ok = ok & (tag == 0x30u) & (ce == L);
    if constexpr (!TBS_ONLY) {
      rd_hdr(hv, L, cs, L, ok, tag, cs, ce);
      ok = ok & (tag == 0x30u);
    }
    if (hv.out) r.defer_exact();
  } else {
    r.touch(0, 256);
    if constexpr (!TBS_ONLY) {
      // Certificate ::= SEQUENCE filling the buffer exactly
      rd_hdr(r, L, 0, L, ok, tag, cs, ce);
      ok = ok & (tag == 0x30u) & (ce == L);
      // tbsCertificate
      rd_hdr(r, L, cs, L, ok, tag, cs, ce);
      ok = ok & (tag == 0x30u);
    } else {
      rd_hdr(r, L, 0, L, ok, tag, cs, ce);
      ok = ok & (tag == 0x30u) & (ce == L);
    }
  }
  const uint32_t tbs_end = ce;
  uint32_t q = cs;
  // Version int `asn1:"optional,explicit,default:0,tag:0"`: an empty wrapper is an error ("zero length explicit tag
  // was not an asn1.Flag"); otherwise the inner INTEGER is parsed against the TBSCertificate and parsing resumes
  // behind IT.
  if ((q < tbs_end) & ((ldc(r, q, L) & 0xffu) == 0xa0u)) {
    uint32_t vs, ve, t2, is_, ie;
    rd_hdr<false>(r, L, q, tbs_end, ok, tag, vs, ve);
    rd_hdr(r, L, vs, tbs_end, ok, t2, is_, ie);
    ok = ok & (ve != vs) & (t2 == 0x02u);
    int32_check(r, L, is_, ie - is_, ok, o.nonfatal);
    q = ie;
  }
  // serialNumber: raw content octets
  rd_hdr(r, L, q, tbs_end, ok, tag, cs, ce);
  ok = ok & (tag == 0x02u);
  {
    bool neg;
    int_check(r, L, cs, ce - cs, ok, o.nonfatal, neg);
    o.nonfatal = neg ? (o.nonfatal | WALK_NF_NEGATIVE_SERIAL) : o.nonfatal;
  }
  o.serial_off = cs;
  o.serial_len = ce - cs;
  {
    const uint32_t take = ok ? (o.serial_len < 20u ? o.serial_len : 20u) : 0u;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint32_t pos = 4u * k;
      if (pos < take) {
        const uint32_t rem = take - pos;
        const uint32_t v = r.ld4(cs + pos);
        o.serial_w[k] = rem >= 4 ? v : (v & (0xffffffffu >> (8 * (4 - rem))));
      }
    }
  }
  q = ce;
  CTMR_STAGE(1);
  // signature AlgorithmIdentifier
  alg_id(r, L, q, tbs_end, ok, q);
  // issuer Name → last string-typed CommonName
  {
    const uint32_t n0 = q;
    o.issuer_name = q;
    q = walk_name<true, STRINGS>(r, L, q, tbs_end, ok, o.cn_off, o.cn_len, o.nonfatal, strings);
    o.meta_issuer = meta_pack(n0, q - n0);
    note_issuer_if(r, n0, q - n0);  // readers that look the Name up while it is at hand (the map kernel's memo pre-check)
    if (use_filter) o.cn_match = ok ? cn_prefix_match(r, L, o.cn_off, o.cn_len, fv) : false;
  }
  CTMR_STAGE(2);
  touch_all(r, q, 48);
  // validity: two Times; anything behind them is ignored
  rd_hdr(r, L, q, tbs_end, ok, tag, cs, ce);
  ok = ok & (tag == 0x30u);
  {
    uint32_t t1, c1, e1;
    rd_hdr(r, L, cs, ce, ok, t1, c1, e1);
    rd_time(r, L, c1, t1, e1 - c1, ok, o.not_before);
    rd_hdr(r, L, e1, ce, ok, t1, c1, e1);
    rd_time(r, L, c1, t1, e1 - c1, ok, o.not_after);
  }
  q = ce;
  CTMR_STAGE(3);
  // subject Name: same structure, nothing of it is consumed
  {
#ifdef CTMR_EXPERIMENT_SKIP_SUBJECT  // sweep builds only: what validating the subject costs (round 1 skipped it by length)
    rd_hdr(r, L, q, tbs_end, ok, tag, cs, ce);
    ok = ok & (tag == 0x30u);
    q = ce;
#else
    uint32_t d0 = 0, d1 = 0;
    o.subject_name = q;
    q = walk_name<false, STRINGS>(r, L, q, tbs_end, ok, d0, d1, o.nonfatal, strings);
#endif
  }
  if constexpr (NAMES_ONLY) return ok;
  CTMR_STAGE(4);
  // subjectPublicKeyInfo (full TLV = RawSubjectPublicKeyInfo): publicKeyInfo ::= SEQUENCE { AlgorithmIdentifier,
  // BIT STRING }; the key bits themselves are skipped by length.  A long subject (OV/EV certificates) puts this header
  // past the front window: say so, instead of leaving a window-only reader to its slow exact path.
  // (A wave-cooperative form of this refill — the lanes in need served 16 lanes per certificate, as touch_tail does —
  //  measured no gain on the mixed corpus: 25.45 ms against 25.3 ms per 100 M, session 5.)
  touch_all(r, q, SPKI_HEAD_NEED);
  rd_hdr(r, L, q, tbs_end, ok, tag, cs, ce);
  ok = ok & (tag == 0x30u);
  o.spki_off = q;
  o.spki_len = ce - q;
  AlgView key_alg{0u, 0u, 0u, 0u, 0u, 0u};
  KeyPending key_pending;
  key_pending.alg = PK_OTHER;
  {
    uint32_t k, tk, ck, ek;
    alg_id(r, L, cs, ce, ok, k, key_alg);
    rd_hdr(r, L, k, ce, ok, tk, ck, ek);
    ok = ok & (tk == 0x03u);
    bit_string_check(r, L, ck, ek - ck, ok);
    // the key itself: what the window holds now, and the far reads of the common case issued (spki_key.h)
    if (spki) spki_key_begin(r, L, key_alg, ck, ek, ok, o.nonfatal, key_pending);
  }
  q = ce;
  // what follows the key (unique ids, extensions) and the tail behind the TBS are both known now:
  // a two-region reader fetches them in one burst
  r.touch_tail(q, tbs_end);
  if (spki) {
    key_tail_of(r, q, key_pending);  // the 16 octets around the key's end, when the reader fetched them with the tail
    EcPending ecp;
    spki_key_finish<EC_DEFER>(r, L, key_alg, key_pending, ok, o.nonfatal, ecp);
    o.ec_curve = ecp.curve; o.ec_pos = ecp.pos; o.ec_shift = ecp.shift;
  }
  CTMR_STAGE(5);
  // UniqueId, SubjectUniqueId asn1.BitString `optional,tag:1|2`, Extensions `optional,explicit,tag:3`: each parses the
  // header at the current position (which must be a valid header) and skips itself when the tag is not its own;
  // whatever is left in the TBSCertificate after the three is ignored.
  uint32_t nt = 0u;
  uint32_t san_cv = 0u, san_ev = 0u;  // strict_extensions, cooperative readers: the subjectAltName's value, walked behind the loop
  (void)san_cv; (void)san_ev;
  if (ok & (q < tbs_end)) {
    rd_hdr<false>(r, L, q, tbs_end, ok, tag, cs, ce);
    nt = ok ? tag : 0u;
  }
  if (nt == 0x81u) {
    ok = ok & (ce <= tbs_end);
    bit_string_check(r, L, cs, ce - cs, ok);
    q = ce;
    nt = 0u;
    if (ok & (q < tbs_end)) {
      rd_hdr<false>(r, L, q, tbs_end, ok, tag, cs, ce);
      nt = ok ? tag : 0u;
    }
  }
  if (nt == 0x82u) {
    ok = ok & (ce <= tbs_end);
    bit_string_check(r, L, cs, ce - cs, ok);
    q = ce;
    nt = 0u;
    if (ok & (q < tbs_end)) {
      rd_hdr<false>(r, L, q, tbs_end, ok, tag, cs, ce);
      nt = ok ? tag : 0u;
    }
  }
  // [3] matches when constructed or empty, and empty is an error
  if ((nt == 0xa3u) | (nt == 0x83u)) ok = ok & (ce != cs);
  if (ok & (nt == 0xa3u)) {
    uint32_t e, e_end;
    r.touch(q, 48);
    rd_hdr<false>(r, L, cs, tbs_end, ok, tag, e, e_end);
    // an inner element that is not a SEQUENCE leaves the optional field unset: no extensions at all
    ok = ok & ((tag != 0x30u) | (e_end <= tbs_end));
    e_end = tag == 0x30u ? e_end : e;
    while (ok & (e < e_end)) {
      uint32_t t1, x, x_end, to, co, eo, tv, cv, ev, oidw;
      r.touch(e, 16);  // the 12 octets of the fast form (and the OID's second word, for strict_extensions)
      // Fast form: Extension hdr, a 3-byte extnID with its hdr, optional critical BOOLEAN and the extnValue hdr
      // lie in the 12 bytes at e when every length is short form (every 2.5.29.x extension under 128 bytes).
      const uint32_t w0 = ldc(r, e, L), w1 = ldc(r, e + 4u, L), w2 = ldc(r, e + 8u, L);
      const bool p0 = ((w0 & 0xffff80ffu) == 0x03060030u) & ((w1 & 0x808080u) == 0u);  // three one-octet arcs (2.5.29.x)
      const uint32_t b7 = w1 >> 24, b9 = (w2 >> 8) & 0xffu;
      const bool nc = p0 & (b7 == 0x04u) & ((w2 & 0x80u) == 0u);
      const bool cr = p0 & (b7 == 0x01u) & ((w2 & 0x80ff00ffu) == 0x00040001u) & ((b9 == 0x00u) | (b9 == 0xffu));
      if (nc | cr) {
        x_end = e + 2u + ((w0 >> 8) & 0xffu);
        co = e + 4u;
        eo = e + 7u;
        oidw = w1;
        cv = e + (cr ? 12u : 9u);
        ev = cv + (cr ? (w2 >> 24) : (w2 & 0xffu));
        ok = ok & (x_end <= e_end) & (ev <= x_end);
      } else {
        r.touch(e, 24);  // three headers and the OID's first words (a subjectAltName's: 30 82 ll ll 06 03 55 1d 11 04 82 ll ll)
        rd_hdr(r, L, e, e_end, ok, t1, x, x_end);   // Extension
        rd_hdr(r, L, x, x_end, ok, to, co, eo);     // extnID
        oidw = ldc(r, co, L);
        rd_hdr(r, L, eo, x_end, ok, tv, cv, ev);    // critical or extnValue
        ok = ok & (t1 == 0x30u) & (to == 0x06u);
        ok = ok && oid_arcs_ok(r, L, co, eo);
        if (tv == 0x01u) {  // critical BOOLEAN
          const uint32_t bv = ldc(r, cv, L) & 0xffu;
          ok = ok & (ev - cv == 1u) & ((bv == 0x00u) | (bv == 0xffu));
          rd_hdr(r, L, ev, x_end, ok, tv, cv, ev);
        }
        ok = ok & (tv == 0x04u);
      }
      {  // cRLDistributionPoints 2.5.29.31: only located here, decoded by k_meta_new for new certificates
        const bool is_crl = (eo - co == 3u) & ((oidw & 0xffffffu) == 0x1f1d55u);
        const uint32_t pk = o.meta_crl == META_NONE ? meta_pack(cv, ev - cv) : META_HOST;
        o.meta_crl = is_crl ? pk : o.meta_crl;
      }
      if constexpr (STRINGS) {  // strict_extensions: the body of an extension Go unmarshals
        if (ext & ok) {
          const uint32_t kind = ext_kind(eo - co, oidw, ldc(r, co + 4u, L));
          if ((kind >= 1u) & (kind <= 6u)) {
            ext_body_check(r, L, kind, cv, ev, ok);
          } else if (kind == 7u) {
            if constexpr (has_coop_refill<R>::value) {  // walked behind the loop, by the whole wave (ext_san_coop)
              if ((san_ev == 0u) & (ev > cv)) { san_cv = cv; san_ev = ev; }
              else ext_san_check(r, L, cv, ev, ok, o.nonfatal);  // an empty value (an error), or a second subjectAltName
            } else {
              ext_san_check(r, L, cv, ev, ok, o.nonfatal);
            }
          } else if (kind == 9u) {
            uint32_t uo[1] = {0u}, ul[1] = {0u}, nu = 0u;
            r.touch(cv, ev - cv < 200u ? ev - cv : 200u);
            crl_dps<false, true, STRINGS, 1u>(r, L, cv, ev, ok, uo, ul, nu, o.nonfatal, strings);
          } else if (kind == 8u) {  // (contents: a reader that serves its window alone hands the certificate to the exact one)
            if constexpr (has_defer_exact<R>::value) r.defer_exact();
            else ext_nc_check(r, L, cv, ev, ok);
          } else if ((kind >= 12u) & ext_nf) {  // RFC 3779 (CT-go: findings, never fatal) — only where a finding can matter
            if constexpr (has_defer_exact<R>::value) r.defer_exact();
            else o.nonfatal = (kind == 12u ? ext_ipaddr_ok(r, L, cv, ev) : ext_asnum_ok(r, L, cv, ev)) ? o.nonfatal : (o.nonfatal | WALK_NF_EXT);
          } else if ((kind == 10u) & ext_nf && ((ldc(r, co + 8u, L) & 0xffffu) == 0x0204u)) {
            // the embedded SCT list: a finding only a precertificate or an issuer could lose its place over
            if constexpr (has_defer_exact<R>::value) r.defer_exact();
            else o.nonfatal = ext_sct_ok(r, L, cv, ev) ? o.nonfatal : (o.nonfatal | WALK_NF_EXT);
          }
        }
      }
      if (ok & (eo - co == 3u) & ((oidw & 0xffffffu) == 0x131d55u)) {
        // basicConstraints struct { IsCA bool `optional`; MaxPathLen int `optional,default:-1` } must be the whole
        // OCTET STRING ("x509: trailing data after X.509 BasicConstraints"); inside the SEQUENCE an element of
        // another type leaves the optional field at its default and whatever follows is ignored
        uint32_t tb, c, c_end, tf = 0u, cf = 0u, ef = 0u;
        rd_hdr(r, L, cv, ev, ok, tb, c, c_end);
        ok = ok & (tb == 0x30u) & (c_end == ev);
        bool ca = false;
        if (ok & (c < c_end)) {
          rd_hdr<false>(r, L, c, c_end, ok, tf, cf, ef);
          if (tf == 0x01u) {
            const uint32_t bv = ldc(r, cf, L) & 0xffu;
            ok = ok & (ef <= c_end) & (ef - cf == 1u) & ((bv == 0x00u) | (bv == 0xffu));
            ca = bv == 0xffu;
            c = ef;
            tf = 0u;
            if (ok & (c < c_end)) rd_hdr<false>(r, L, c, c_end, ok, tf, cf, ef);
          }
          if (ok & (c < c_end) & (tf == 0x02u)) {  // pathLenConstraint
            ok = ok & (ef <= c_end);
            int32_check(r, L, cf, ef - cf, ok, o.nonfatal);
          }
        }
        o.bc_valid = true;
        o.is_ca = ca;
      }
      e = x_end;
    }
  }
  if constexpr (STRINGS && has_coop_refill<R>::value) {
    if (ext) ext_san_coop(r, L, san_cv, ok ? san_ev : 0u, ok, o.nonfatal);
  }
  CTMR_STAGE(6);
  if constexpr (!TBS_ONLY) {
    // signatureAlgorithm, signatureValue BIT STRING (Go asn1 parseBitString); bytes behind them are ignored
    const TailView<R> tv{r};
    uint32_t sq;
    alg_id(tv, L, tbs_end, L, ok, sq);
    rd_hdr(tv, L, sq, L, ok, tag, cs, ce);
    ok = ok & (tag == 0x03u);
    bit_string_check(tv, L, cs, ce - cs, ok);
  }
  o.ec_curve = ok ? o.ec_curve : 0u;
  return ok;
}

template <class R>
CTMR_HD bool walk_cert(R& r, uint32_t L, Walk& o, const FilterView* filter = nullptr, bool spki = true, bool strings = false,
                       bool ext = false, bool ext_nf = true) {
  const uint32_t m = ext ? (ext_nf ? WALK_EXT_ON | WALK_EXT_NF : WALK_EXT_ON) : 0u;
  return filter ? walk_cert<R>(r, L, o, true, *filter, spki, strings, m)
                : walk_cert<R>(r, L, o, false, FilterView{0, nullptr, nullptr, nullptr}, spki, strings, m);
}
template <class R>
CTMR_HD bool walk_tbs(R& r, uint32_t L, Walk& o, bool spki = true, bool ext = false) {
  // (ct.LogEntryFromLeaf drops a precertificate entry over its TBSCertificate on FATAL errors only: no ext_nf)
  return walk_cert<R, true>(r, L, o, false, FilterView{0, nullptr, nullptr, nullptr}, spki, false, ext ? WALK_EXT_ON : 0u);
}

}  // namespace ctmr
