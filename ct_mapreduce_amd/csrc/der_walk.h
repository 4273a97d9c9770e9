// der_walk.h — the TBSCertificate field walk of the map kernel.
//
// One function, walk_cert<Reader>(), written for one-certificate-per-lane execution on
// CDNA4: every byte access goes through Reader::ld4(pos), a 4-byte little-endian window at an
// arbitrary byte position (LDS tile or global memory — kernels.hip supplies both), so one
// TLV header (tag, length byte, up to two long-form length bytes) costs a single load.
//
// It replaces, for the fields the reference path consumes, the call
// x509.ParseCertificate at cmd/ct-fetch/ct-fetch.go:202,221 and storage.NewSerial
// (storage/types.go:165-178).  The accept/reject profile is DESIGN.md §3.
//
// CTMR_HD lets tests/harness compile this exact code for the host to fuzz it against the
// oracle without a GPU; the shipped library only ever instantiates it in device code.
#pragma once
#include <stdint.h>

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
};

struct Hdr {
  uint32_t tag, hl, len;
};

// Decode the TLV header at p; header and content must fit inside [p, end).
// Go encoding/asn1 parseTagAndLength rules: single-byte tags, definite minimal lengths < 2^31.
template <class R>
CTMR_HD bool rd_hdr(const R& r, uint32_t p, uint32_t end, Hdr& h) {
  if (end < 2 || p > end - 2) return false;
  const uint32_t w = r.ld4(p);
  h.tag = w & 0xffu;
  if ((h.tag & 0x1fu) == 0x1fu) return false;
  const uint32_t b = (w >> 8) & 0xffu;
  if (b < 0x80u) {
    h.hl = 2;
    h.len = b;
  } else {
    const uint32_t n = b & 0x7fu;
    if (n == 0 || n > 4) return false;
    if (end - p - 2 < n) return false;
    uint32_t v;
    if (n == 1) {
      v = (w >> 16) & 0xffu;
      if (v < 0x80u) return false;  // non-minimal (also covers a zero byte)
    } else if (n == 2) {
      const uint32_t b2 = (w >> 16) & 0xffu;
      if (b2 == 0) return false;    // superfluous leading zero
      v = (b2 << 8) | (w >> 24);
    } else {
      const uint32_t x = r.ld4(p + 2);  // the n length bytes, big endian
      if ((x & 0xffu) == 0) return false;
      const uint32_t be = __builtin_bswap32(x);
      v = n == 3 ? (be >> 8) : be;
      if (v > 0x7fffffffu) return false;
    }
    h.hl = 2 + n;
    h.len = v;
  }
  return h.len <= end - p - h.hl;
}

CTMR_HD bool digits4(uint32_t w) {  // four ASCII digits?
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t c = (w >> (8 * i)) & 0xffu;
    ok = ok && (c - 0x30u) <= 9u;
  }
  return ok;
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

// UTCTime "YYMMDDHHMM[SS]Z" (tag 0x17, len 11/13) or GeneralizedTime "YYYYMMDDHHMMSSZ"
// (tag 0x18, len 15).  Only the Z forms are in the profile.
template <class R>
CTMR_HD bool rd_time(const R& r, uint32_t c, const Hdr& h, int64_t& out) {
  uint32_t w0 = r.ld4(c), w1 = r.ld4(c + 4), w2 = r.ld4(c + 8), w3 = r.ld4(c + 12);
  int32_t year;
  if (h.tag == 0x18u) {
    if (h.len != 15) return false;
    if (!digits4(w0)) return false;
    year = (int32_t)(d2(w0, 0) * 100u + d2(w0, 16));
    // drop the century: shift the 16-byte window down by two bytes
    w0 = (w0 >> 16) | (w1 << 16);
    w1 = (w1 >> 16) | (w2 << 16);
    w2 = (w2 >> 16) | (w3 << 16);
    w3 = w3 >> 16;
    // now w0.. = "YYMMDDHHMMSSZ"
    if (!digits4(w0) || !digits4(w1) || !digits4(w2)) return false;
    if ((w3 & 0xffu) != 'Z') return false;
  } else if (h.tag == 0x17u) {
    if (h.len == 13) {
      if (!digits4(w0) || !digits4(w1) || !digits4(w2)) return false;
      if ((w3 & 0xffu) != 'Z') return false;
    } else if (h.len == 11) {
      if (!digits4(w0) || !digits4(w1)) return false;
      if (!digits4((w2 & 0xffffu) | 0x30300000u)) return false;
      if (((w2 >> 16) & 0xffu) != 'Z') return false;
      w2 = (w2 & 0xffffu) | 0x30300000u;  // seconds = "00"
    } else {
      return false;
    }
    const uint32_t yy = d2(w0, 0);
    year = (int32_t)(yy < 50 ? 2000 + yy : 1900 + yy);
  } else {
    return false;
  }
  const uint32_t mon = d2(w0, 16), day = d2(w1, 0), hh = d2(w1, 16), mm = d2(w2, 0), ss = d2(w2, 16);
  if (mon < 1 || mon > 12) return false;
  uint32_t dim = 31u - ((0xA50u >> mon) & 1u);  // 30-day months: Apr Jun Sep Nov
  if (mon == 2) dim = ((year % 4 == 0) && (year % 100 != 0 || year % 400 == 0)) ? 29 : 28;
  if (day < 1 || day > dim) return false;
  if (hh > 23 || mm > 59 || ss > 59) return false;
  out = days_from_civil(year, mon, day) * 86400 + (int64_t)(hh * 3600u + mm * 60u + ss);
  return true;
}

// Go asn1 checkInteger on content [c, c+len): non-empty and minimally encoded.
template <class R>
CTMR_HD bool int_ok(const R& r, uint32_t c, uint32_t len) {
  if (len == 0) return false;
  if (len == 1) return true;
  const uint32_t w = r.ld4(c);
  const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu;
  if (b0 == 0x00u && (b1 & 0x80u) == 0) return false;
  if (b0 == 0xffu && (b1 & 0x80u) != 0) return false;
  return true;
}

CTMR_HD bool string_tag(uint32_t t) {
  return t == 0x0cu || t == 0x12u || t == 0x13u || t == 0x14u || t == 0x16u;
}

template <class R>
CTMR_HD bool walk_cert(const R& r, uint32_t L, Walk& o) {
  o.serial_off = o.serial_len = 0;
  o.not_before = o.not_after = 0;
  o.cn_off = o.cn_len = 0;
  o.spki_off = o.spki_len = 0;
  o.bc_valid = o.is_ca = false;
  if (L > 0x7fffffffu) return false;
  Hdr h;
  // Certificate ::= SEQUENCE filling the buffer exactly
  if (!rd_hdr(r, 0, L, h) || h.tag != 0x30u || h.hl + h.len != L) return false;
  uint32_t p = h.hl;
  if (!rd_hdr(r, p, L, h) || h.tag != 0x30u) return false;
  const uint32_t tbs_end = p + h.hl + h.len;
  uint32_t q = p + h.hl;
  // version [0] EXPLICIT INTEGER
  if (q < tbs_end && (r.ld4(q) & 0xffu) == 0xa0u) {
    if (!rd_hdr(r, q, tbs_end, h)) return false;
    Hdr v;
    const uint32_t vq = q + h.hl;
    if (!rd_hdr(r, vq, vq + h.len, v) || v.tag != 0x02u) return false;
    if (v.hl + v.len != h.len || v.len > 4 || !int_ok(r, vq + v.hl, v.len)) return false;
    q += h.hl + h.len;
  }
  // serialNumber
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x02u) return false;
  if (!int_ok(r, q + h.hl, h.len)) return false;
  o.serial_off = q + h.hl;
  o.serial_len = h.len;
  q += h.hl + h.len;
  // signature AlgorithmIdentifier
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x30u) return false;
  q += h.hl + h.len;
  // issuer Name → last string-typed CommonName
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x30u) return false;
  {
    uint32_t s = q + h.hl;
    const uint32_t s_end = s + h.len;
    while (s < s_end) {
      Hdr set;
      if (!rd_hdr(r, s, s_end, set) || set.tag != 0x31u) return false;
      uint32_t a = s + set.hl;
      const uint32_t a_end = a + set.len;
      while (a < a_end) {
        Hdr atv, oid, val;
        if (!rd_hdr(r, a, a_end, atv) || atv.tag != 0x30u) return false;
        const uint32_t b = a + atv.hl, b_end = b + atv.len;
        if (!rd_hdr(r, b, b_end, oid) || oid.tag != 0x06u || oid.len == 0) return false;
        const uint32_t vpos = b + oid.hl + oid.len;
        if (!rd_hdr(r, vpos, b_end, val)) return false;
        if (oid.len == 3 && (r.ld4(b + oid.hl) & 0xffffffu) == 0x030455u && string_tag(val.tag)) {
          o.cn_off = vpos + val.hl;
          o.cn_len = val.len;
        }
        a = b_end;
      }
      s = a_end;
    }
  }
  q += h.hl + h.len;
  // validity
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x30u) return false;
  {
    uint32_t v = q + h.hl;
    const uint32_t v_end = v + h.len;
    Hdr tm;
    if (!rd_hdr(r, v, v_end, tm) || !rd_time(r, v + tm.hl, tm, o.not_before)) return false;
    v += tm.hl + tm.len;
    if (!rd_hdr(r, v, v_end, tm) || !rd_time(r, v + tm.hl, tm, o.not_after)) return false;
  }
  q += h.hl + h.len;
  // subject
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x30u) return false;
  q += h.hl + h.len;
  // subjectPublicKeyInfo (full TLV = RawSubjectPublicKeyInfo)
  if (!rd_hdr(r, q, tbs_end, h) || h.tag != 0x30u) return false;
  o.spki_off = q;
  o.spki_len = h.hl + h.len;
  q += h.hl + h.len;
  // [1] issuerUniqueID, [2] subjectUniqueID
  uint32_t nt = q < tbs_end ? (r.ld4(q) & 0xffu) : 0u;
  if (nt == 0x81u) {
    if (!rd_hdr(r, q, tbs_end, h)) return false;
    q += h.hl + h.len;
    nt = q < tbs_end ? (r.ld4(q) & 0xffu) : 0u;
  }
  if (nt == 0x82u) {
    if (!rd_hdr(r, q, tbs_end, h)) return false;
    q += h.hl + h.len;
    nt = q < tbs_end ? (r.ld4(q) & 0xffu) : 0u;
  }
  // [3] EXPLICIT Extensions
  if (nt == 0xa3u) {
    if (!rd_hdr(r, q, tbs_end, h)) return false;
    Hdr seq;
    const uint32_t e0 = q + h.hl;
    if (!rd_hdr(r, e0, e0 + h.len, seq) || seq.tag != 0x30u) return false;
    uint32_t e = e0 + seq.hl;
    const uint32_t e_end = e + seq.len;
    while (e < e_end) {
      Hdr ext, oid, val;
      if (!rd_hdr(r, e, e_end, ext) || ext.tag != 0x30u) return false;
      uint32_t x = e + ext.hl;
      const uint32_t x_end = x + ext.len;
      if (!rd_hdr(r, x, x_end, oid) || oid.tag != 0x06u || oid.len == 0) return false;
      const bool is_bc = oid.len == 3 && (r.ld4(x + oid.hl) & 0xffffffu) == 0x131d55u;
      x += oid.hl + oid.len;
      if (!rd_hdr(r, x, x_end, val)) return false;
      if (val.tag == 0x01u) {  // critical
        if (val.len != 1) return false;
        const uint32_t bv = r.ld4(x + val.hl) & 0xffu;
        if (bv != 0x00u && bv != 0xffu) return false;
        x += val.hl + val.len;
        if (!rd_hdr(r, x, x_end, val)) return false;
      }
      if (val.tag != 0x04u) return false;
      if (is_bc) {
        Hdr bc, f;
        const uint32_t ob = x + val.hl, ob_end = ob + val.len;
        if (!rd_hdr(r, ob, ob_end, bc) || bc.tag != 0x30u || bc.hl + bc.len != val.len) return false;
        uint32_t c = ob + bc.hl;
        const uint32_t c_end = c + bc.len;
        bool ca = false;
        if (c < c_end) {
          if (!rd_hdr(r, c, c_end, f)) return false;
          if (f.tag == 0x01u) {
            if (f.len != 1) return false;
            const uint32_t bv = r.ld4(c + f.hl) & 0xffu;
            if (bv != 0x00u && bv != 0xffu) return false;
            ca = bv == 0xffu;
            c += f.hl + f.len;
            if (c < c_end && !rd_hdr(r, c, c_end, f)) return false;
          }
          if (c < c_end && (f.tag != 0x02u || !int_ok(r, c + f.hl, f.len))) return false;
        }
        o.bc_valid = true;
        o.is_ca = ca;
      }
      e = x_end;
    }
  }
  // signatureAlgorithm, signatureValue BIT STRING (Go asn1 parseBitString)
  p = tbs_end;
  if (!rd_hdr(r, p, L, h) || h.tag != 0x30u) return false;
  p += h.hl + h.len;
  if (!rd_hdr(r, p, L, h) || h.tag != 0x03u || h.len == 0) return false;
  const uint32_t pad = r.ld4(p + h.hl) & 0xffu;
  if (pad > 7 || (h.len == 1 && pad > 0)) return false;
  if (pad > 0 && ((r.ld4(p + h.hl + h.len - 1) & 0xffu) & ((1u << pad) - 1u)) != 0) return false;
  return true;
}

}  // namespace ctmr
