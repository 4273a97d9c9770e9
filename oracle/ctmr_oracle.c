/*
 * ctmr_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ctmr_oracle.h header).
 *
 * Plain-C restatement of the ct-mapreduce per-entry map + known-certificates reduce.
 * Citations are to /root/reference (jcjones/ct-mapreduce @ v1).
 *
 * The DER walk restates the part of x509.ParseCertificate (third-party,
 * certificate-transparency-go v1.1.0 — not on this machine, PARITY UNPINNED there) that
 * the reference path consumes, following RFC 5280 §4.1 and the Go encoding/asn1 rules
 * recalled in DESIGN.md §3 ("DER walk profile").
 */
#include "ctmr_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ DER walk
 *
 * What "x509.ParseCertificate succeeded" means here.  The parser itself is third-party
 * (certificate-transparency-go v1.1.0, a fork of Go's crypto/x509 + encoding/asn1; absent from this machine), so this
 * restates Go's encoding/asn1 *struct-unmarshalling rules* applied to the x509 `certificate` / `tbsCertificate` /
 * `pkix.*` struct definitions, for every element whose bytes the path reads anyway:
 *   - parseTagAndLength: definite minimal lengths < 2^31, high-tag-number form allowed (minimal, < 2^31);
 *   - a struct field must match its universal tag, lie inside the enclosing contents ("data truncated"), and
 *     *bytes left over at the end of a SEQUENCE are ignored* ("adding elements to the end has been used in X.509");
 *   - an OPTIONAL field whose tag does not match is skipped — but its header must still be a valid header;
 *   - an EXPLICIT wrapper's own length is never checked against its inner element: parsing resumes right behind
 *     the INNER element ([0] version, [3] extensions);
 *   - UTCTime / GeneralizedTime go through time.Parse("0601021504Z0700" | "060102150405Z0700" |
 *     "20060102150405Z0700") and must serialise back to the same string: 'Z' or a NON-ZERO numeric offset ±hhmm with
 *     mm <= 59 (hh is not range-checked by the Go 1.13 toolchain the reference pins; go.mod:24);
 *   - INTEGERs must be non-empty and minimal (checkInteger); int fields must fit int32.
 * CT-go returns some findings as x509.NonFatalErrors instead of failing: the certificate is still handed out.  The
 * reference keeps such a certificate when it arrives as an X509 entry (ct.LogEntryFromLeaf only drops the entry on
 * x509.IsFatal errors, cmd/ct-fetch/ct-fetch.go:452-459) and drops it when it is a precertificate or a Chain[0]
 * issuer (any err, :202-209, :221-225).  The findings modelled as non-fatal — recalled, not verifiable here — are in
 * ORC_NF_*: a negative serialNumber, and INTEGERs that only the "lax" re-parse accepts (not minimally encoded).
 * NOT checked (needs bytes the path never reads, or CT-go's source): the public key itself, extension bodies other
 * than basicConstraints, string character sets, attribute values of non-string types, OID arcs >= 2^31.
 */

typedef struct {
  uint8_t tag;  /* identifier octet (class | constructed | number; number == 0x1f: high-tag-number form) */
  uint32_t hl;  /* header length */
  uint32_t len; /* content length */
} tlv;

/* Go encoding/asn1 parseTagAndLength at p, reading no byte at or past `end`: the header only. */
static int rd_hdr(const uint8_t* d, uint64_t p, uint64_t end, tlv* t) {
  if (p >= end) return 0;
  uint64_t o = p;
  t->tag = d[o++];
  if ((t->tag & 0x1f) == 0x1f) { /* parseBase128Int: <= 5 octets, minimal, value in [0x1f, 2^31) */
    uint64_t v = 0;
    int k = 0;
    for (;;) {
      if (o >= end) return 0;      /* truncated base 128 integer */
      if (k == 5) return 0;        /* base 128 integer too large */
      uint8_t b = d[o++];
      if (k == 0 && b == 0x80) return 0; /* integer is not minimally encoded */
      v = (v << 7) | (b & 0x7f);
      k++;
      if (!(b & 0x80)) break;
    }
    if (v > 0x7fffffffu) return 0;
    if (v < 0x1f) return 0;        /* non-minimal tag */
  }
  if (o >= end) return 0;          /* truncated tag or length */
  uint8_t b = d[o++];
  if (b < 0x80) {
    t->len = b;
  } else {
    uint32_t n = b & 0x7f;
    if (n == 0 || n > 4) return 0; /* indefinite length; length too large */
    if (o + n > end) return 0;
    if (d[o] == 0) return 0;       /* superfluous leading zeros in length */
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; i++) v = (v << 8) | d[o + i];
    o += n;
    if (v < 0x80) return 0;        /* non-minimal length */
    if (v > 0x7fffffffu) return 0; /* length too large */
    t->len = (uint32_t)v;
  }
  t->hl = (uint32_t)(o - p);
  return 1;
}

/* header + contents inside [p, end) (Go: invalidLength → "data truncated") */
static int rd_tlv(const uint8_t* d, uint64_t p, uint64_t end, tlv* t) {
  if (!rd_hdr(d, p, end, t)) return 0;
  return p + t->hl + (uint64_t)t->len <= end;
}

static int is_digit(uint8_t c) { return c >= '0' && c <= '9'; }
static int two(const uint8_t* s) { return (s[0] - '0') * 10 + (s[1] - '0'); }

/* days since 1970-01-01 of a proleptic Gregorian civil date */
static int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  int64_t era = (y >= 0 ? y : y - 399) / 400;
  int64_t yoe = y - era * 400;
  int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}

static int days_in_month(int64_t y, int m) {
  static const int dm[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  if (m == 2) {
    int leap = (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0);
    return leap ? 29 : 28;
  }
  return dm[m - 1];
}

/* The "Z0700" element of the three layouts: 'Z', or sign hh mm.  Accepted iff the parsed time serialises back to
 * the input (asn1.parseUTCTime / parseGeneralizedTime): digits only, mm <= 59 (60 would print as the next hour),
 * and not ±0000 (offset 0 prints as "Z").  hh: any two digits — time.Parse of Go 1.13 does not range-check it and
 * Format prints offset/60/60 with two digits.  *off = seconds east of UTC. */
static int parse_zone(const uint8_t* s, uint32_t n, int64_t* off) {
  if (n == 1 && s[0] == 'Z') {
    *off = 0;
    return 1;
  }
  if (n != 5 || (s[0] != '+' && s[0] != '-')) return 0;
  for (int i = 1; i < 5; i++)
    if (!is_digit(s[i])) return 0;
  int hh = two(s + 1), mm = two(s + 3);
  if (mm > 59 || (hh == 0 && mm == 0)) return 0;
  *off = (int64_t)(hh * 3600 + mm * 60) * (s[0] == '-' ? -1 : 1);
  return 1;
}

/* UTCTime (tag 0x17) "YYMMDDHHMM[SS]" + zone, GeneralizedTime (0x18) "YYYYMMDDHHMMSS" + zone.
 * UTCTime years: time.Parse maps yy >= 69 to 19yy, else 20yy; parseUTCTime then subtracts a century when the year
 * (in the time's own zone) is >= 2050 — together: yy < 50 → 20yy, else 19yy. */
static int parse_time(const uint8_t* d, const tlv* t, uint64_t content, int64_t* out) {
  const uint8_t* s = d + content;
  int64_t year, off = 0;
  int mon, day, hh, mm, ss = 0;
  uint32_t n = t->len;
  if (t->tag == 0x17) {
    /* 10 digits, then either the zone (layout without seconds, tried first) or 2 more digits and the zone */
    if (n < 11) return 0;
    for (uint32_t i = 0; i < 10; i++)
      if (!is_digit(s[i])) return 0;
    uint32_t z = 10;
    if (is_digit(s[10])) {
      if (n < 13 || !is_digit(s[11])) return 0;
      ss = two(s + 10);
      z = 12;
    }
    if (!parse_zone(s + z, n - z, &off)) return 0;
    int yy = two(s);
    year = yy < 50 ? 2000 + yy : 1900 + yy;
    mon = two(s + 2);
    day = two(s + 4);
    hh = two(s + 6);
    mm = two(s + 8);
  } else if (t->tag == 0x18) {
    if (n < 15) return 0;
    for (uint32_t i = 0; i < 14; i++)
      if (!is_digit(s[i])) return 0;
    if (!parse_zone(s + 14, n - 14, &off)) return 0;
    year = two(s) * 100 + two(s + 2);
    mon = two(s + 4);
    day = two(s + 6);
    hh = two(s + 8);
    mm = two(s + 10);
    ss = two(s + 12);
  } else {
    return 0;
  }
  if (mon < 1 || mon > 12) return 0;
  if (day < 1 || day > days_in_month(year, mon)) return 0;
  if (hh > 23 || mm > 59 || ss > 59) return 0;
  *out = days_from_civil(year, mon, day) * 86400 + hh * 3600 + mm * 60 + ss - off;
  return 1;
}

/* Go asn1 checkInteger: non-empty, minimally encoded.  Returns 1 = ok, 0 = empty (an error even for the lax
 * re-parse), -1 = not minimal (strict parse fails, CT-go's lax re-parse accepts: a non-fatal finding). */
static int check_integer(const uint8_t* d, uint64_t content, uint32_t len) {
  if (len == 0) return 0;
  if (len == 1) return 1;
  if (d[content] == 0x00 && (d[content + 1] & 0x80) == 0) return -1;
  if (d[content] == 0xff && (d[content + 1] & 0x80) == 0x80) return -1;
  return 1;
}

/* An `int` field (Version, MaxPathLen): parseInt32 = checkInteger, at most 8 octets, value fits int32.
 * Same three-way result. */
static int check_int32(const uint8_t* d, uint64_t content, uint32_t len) {
  int c = check_integer(d, content, len);
  if (c == 0) return 0;
  if (len > 8) return 0;
  int64_t v = (d[content] & 0x80) ? -1 : 0;
  for (uint32_t i = 0; i < len; i++) v = (int64_t)(((uint64_t)v << 8) | d[content + i]);
  if (v != (int64_t)(int32_t)v) return 0;
  return c;
}

/* Go asn1 parseObjectIdentifier: non-empty, and every base-128 integer of it (parseBase128Int) at most 5 octets long, not
 * led by 0x80 ("integer is not minimally encoded"), at most 2^31 - 1 ("base 128 integer too large") and complete
 * ("truncated base 128 integer").  Round 4 (scripts/diff_openssl.py: OpenSSL rejects a 0x80-led arc as well — "invalid
 * object encoding"); before, only the last octet was looked at. */
static int oid_ok(const uint8_t* d, uint64_t content, uint32_t len) {
  if (len == 0) return 0;
  uint32_t i = 0;
  while (i < len) {
    uint64_t v = 0;
    uint32_t k = 0;
    for (;;) {
      if (i >= len) return 0;               /* truncated */
      if (k == 5) return 0;                 /* too large */
      uint8_t b = d[content + i++];
      if (k == 0 && b == 0x80) return 0;    /* not minimally encoded */
      v = (v << 7) | (b & 0x7f);
      k++;
      if (!(b & 0x80)) break;
    }
    if (v > 0x7fffffffu) return 0;
  }
  return 1;
}

/* parseBitString on contents [c, c+len) */
static int bit_string_ok(const uint8_t* d, uint64_t c, uint32_t len) {
  if (len == 0) return 0;
  uint8_t pad = d[c];
  if (pad > 7 || (len == 1 && pad > 0)) return 0;
  if (pad > 0 && (d[c + len - 1] & ((1u << pad) - 1)) != 0) return 0;
  return 1;
}

static int is_string_tag(uint8_t tag) {
  /* UTF8String, NumericString, PrintableString, T61String, IA5String: the value types Go's
   * asn1 decodes to a Go string, which pkix.Name.FillFromRDNSequence requires */
  return tag == 0x0c || tag == 0x12 || tag == 0x13 || tag == 0x14 || tag == 0x16;
}

#define FAIL(site)          \
  do {                      \
    out->ok = 0;            \
    out->err_site = (site); \
    return;                 \
  } while (0)
#define FAIL0(site)         \
  do {                      \
    *site_out = (site);     \
    return 0;               \
  } while (0)

/* pkix.AlgorithmIdentifier ::= SEQUENCE { algorithm OBJECT IDENTIFIER, parameters ANY OPTIONAL } at p */
static int alg_id(const uint8_t* d, uint64_t p, uint64_t end, tlv* t, int* site_out, int site) {
  if (!rd_tlv(d, p, end, t) || t->tag != 0x30) FAIL0(site);
  uint64_t a = p + t->hl, a_end = a + t->len;
  tlv o;
  if (!rd_tlv(d, a, a_end, &o) || o.tag != 0x06 || !oid_ok(d, a + o.hl, o.len)) FAIL0(site + 1);
  a += o.hl + o.len;
  if (a < a_end && !rd_tlv(d, a, a_end, &o)) FAIL0(site + 2); /* parameters asn1.RawValue `optional` */
  return 1;
}

/* pkix.RDNSequence at p: SEQUENCE OF SET OF SEQUENCE { type OID, value ANY }.  cn: the last AttributeTypeAndValue
 * with OID 2.5.4.3 whose value is a string type (pkix.Name.FillFromRDNSequence), or NULL. */
/* Go encoding/asn1 (go1.13), parseField for an `interface{}` target: a universal, primitive value is decoded by its tag,
 * and a string that breaks its type's character set is a parse error of the STDLIB:
 *   parsePrintableString  isPrintable(b, allowAsterisk, allowAmpersand)  "PrintableString contains invalid character"
 *   parseNumericString    '0'..'9' or ' '                                 "NumericString contains invalid character"
 *   parseIA5String        b < utf8.RuneSelf (0x80)                         "IA5String contains invalid character"
 *   parseUTF8String       utf8.Valid                                       "asn1: invalid UTF-8 string"
 * (T61String: taken as it is.)  What CT-go's lax fork does with them is unverified; the findings are collected apart from
 * `nonfatal` and only an engine with strict_strings set treats them as one more non-fatal finding. */
static int utf8_valid(const uint8_t* s, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    uint32_t c = s[i], len, cp, min;
    if (c < 0x80) { i++; continue; }
    if ((c & 0xe0) == 0xc0) { len = 2; cp = c & 0x1f; min = 0x80; }
    else if ((c & 0xf0) == 0xe0) { len = 3; cp = c & 0x0f; min = 0x800; }
    else if ((c & 0xf8) == 0xf0) { len = 4; cp = c & 0x07; min = 0x10000; }
    else return 0;                       /* a continuation octet or 0xF8.. where a sequence must start */
    if (i + len > n) return 0;           /* truncated */
    for (uint32_t k = 1; k < len; k++) {
      if ((s[i + k] & 0xc0) != 0x80) return 0;
      cp = (cp << 6) | (s[i + k] & 0x3f);
    }
    if (cp < min) return 0;              /* overlong */
    if (cp >= 0xd800 && cp <= 0xdfff) return 0; /* surrogate */
    if (cp > 0x10ffff) return 0;
    i += len;
  }
  return 1;
}

static int string_findings(const uint8_t* s, uint32_t n, uint32_t tag) {
  static const char printable[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789 '()+,-./:=?*&";
  switch (tag) {
    case 0x13:
      for (uint32_t i = 0; i < n; i++)
        if (s[i] == 0 || !memchr(printable, s[i], sizeof printable - 1)) return ORC_SF_PRINTABLE;
      return 0;
    case 0x12:
      for (uint32_t i = 0; i < n; i++)
        if (!((s[i] >= '0' && s[i] <= '9') || s[i] == ' ')) return ORC_SF_NUMERIC;
      return 0;
    case 0x16:
      for (uint32_t i = 0; i < n; i++)
        if (s[i] >= 0x80) return ORC_SF_IA5;
      return 0;
    case 0x0c:
      return utf8_valid(s, n) ? 0 : ORC_SF_UTF8;
    default:
      return 0;
  }
}

/* the elements of an RDNSequence, [r, r_end): SET OF SEQUENCE { type OID, value ANY } each (also what a
 * distributionPointName's nameRelativeToCRLIssuer holds behind its IMPLICIT [1] tag, ext_crldp_site) */
static int rdn_elements(const uint8_t* d, uint64_t r, uint64_t r_end, uint32_t* cn_off, uint32_t* cn_len,
                        int* site_out, int site, int32_t* sfind, int32_t* nfind);

static int rdn_sequence(const uint8_t* d, uint64_t p, uint64_t end, tlv* t, uint32_t* cn_off, uint32_t* cn_len,
                        int* site_out, int site, int32_t* sfind, int32_t* nfind) {
  if (!rd_tlv(d, p, end, t) || t->tag != 0x30) FAIL0(site);
  return rdn_elements(d, p + t->hl, p + t->hl + t->len, cn_off, cn_len, site_out, site, sfind, nfind);
}

static int rdn_elements(const uint8_t* d, uint64_t r, uint64_t r_end, uint32_t* cn_off, uint32_t* cn_len,
                        int* site_out, int site, int32_t* sfind, int32_t* nfind) {
  while (r < r_end) {
    tlv set;
    if (!rd_tlv(d, r, r_end, &set) || set.tag != 0x31) FAIL0(site + 1);
    uint64_t a = r + set.hl, a_end = a + set.len;
    while (a < a_end) {
      tlv atv, oid, val;
      if (!rd_tlv(d, a, a_end, &atv) || atv.tag != 0x30) FAIL0(site + 2);
      uint64_t b = a + atv.hl, b_end = b + atv.len;
      if (!rd_tlv(d, b, b_end, &oid) || oid.tag != 0x06 || !oid_ok(d, b + oid.hl, oid.len)) FAIL0(site + 3);
      uint64_t vpos = b + oid.hl + oid.len;
      if (!rd_tlv(d, vpos, b_end, &val)) FAIL0(site + 4);  /* ANY: must be there and fit; anything behind it is ignored */
      *sfind |= string_findings(d + vpos + val.hl, val.len, val.tag);
      { /* Value interface{}: a universal primitive INTEGER / BIT STRING / OBJECT IDENTIFIER / UTCTime / GeneralizedTime is
         * decoded (parseInt64, parseBitString, parseObjectIdentifier, parseUTCTime, parseGeneralizedTime) and fails the Name
         * when it does not parse; a not minimally encoded INTEGER is what the lax re-parse accepts (round 4) */
        uint64_t vc = vpos + val.hl;
        if (val.tag == 0x02) {
          int ci = check_integer(d, vc, val.len);
          if (ci == 0 || val.len > 8) FAIL0(site + 5);
          if (ci < 0) *nfind |= ORC_NF_LAX_INTEGER;
        } else if (val.tag == 0x03) {
          if (!bit_string_ok(d, vc, val.len)) FAIL0(site + 5);
        } else if (val.tag == 0x06) {
          if (!oid_ok(d, vc, val.len)) FAIL0(site + 5);
        } else if (val.tag == 0x17 || val.tag == 0x18) {
          int64_t unused;
          if (!parse_time(d, &val, vc, &unused)) FAIL0(site + 5);
        }
      }
      if (cn_off && oid.len == 3 && d[b + oid.hl] == 0x55 && d[b + oid.hl + 1] == 0x04 &&
          d[b + oid.hl + 2] == 0x03 && is_string_tag(val.tag)) {
        *cn_off = (uint32_t)(vpos + val.hl);
        *cn_len = val.len;
      }
      a += atv.hl + atv.len;
    }
    r += set.hl + set.len;
  }
  return 1;
}

/* ------------------------------------------------------------------ the public key
 * certificate-transparency-go x509.parsePublicKey (called by parseCertificate for every certificate the path parses:
 * cmd/ct-fetch/ct-fetch.go:202, :221, and :452 inside LogEntryFromLeaf).  CT-go v1.1.0 is not on this machine: the rules
 * are RECALLED (parity unpinned like the rest of the CT-go boundary); OpenSSL's X509_get_pubkey is the
 * independent opinion the tests hold against them.
 *   asn1Data := keyData.PublicKey.RightAlign()
 *   RSA (1.2.840.113549.1.1.1; RSAES-OAEP 1.2.840.113549.1.1.7 for the key part):
 *     parameters != asn1.NullBytes → nfe "RSA key missing NULL parameters" (RSA only);
 *     asn1.Unmarshal(asn1Data, &pkcs1PublicKey{N *big.Int; E int}) — on error the lax re-parse, its success is an nfe;
 *     rest != empty → fatal "trailing data after RSA public key"; N <= 0 → nfe; E <= 0 → fatal
 *   DSA (1.2.840.10040.4.1): asn1Data = INTEGER y (lax re-parse → nfe), no rest; parameters = dsaAlgorithmParameters
 *     {P, Q, G *big.Int} by the strict parser only; any of the four <= 0 → fatal
 *   ECDSA (1.2.840.10045.2.1): parameters = exactly one OBJECT IDENTIFIER; namedCurveFromOID: P-224, P-256, P-384, P-521,
 *     and CT-go's secp192r1 with an nfe; otherwise fatal "unsupported elliptic curve"; elliptic.Unmarshal (Go 1.13):
 *     len = 1 + 2*ceil(bits/8), data[0] = 4, x < p, y < p, IsOnCurve: y^2 = x^3 - 3x + b mod p; nil → fatal
 *   anything else: parsePublicKey returns (nil, nil) — the key is not looked at.
 * Big numbers here: 32-bit limbs, multiplication mod p by double-and-add over the bits of one factor — nothing in common
 * with the product's Montgomery form (ct_mapreduce_amd/csrc/spki_key.h). */
#define BN_W 18
typedef struct { uint32_t w[BN_W]; } bn;

static void bn_from_be(bn* r, const uint8_t* s, size_t n) {
  memset(r, 0, sizeof *r);
  for (size_t i = 0; i < n; i++) {
    size_t bit = 8 * (n - 1 - i);
    r->w[bit / 32] |= (uint32_t)s[i] << (bit % 32);
  }
}
static void bn_from_hex(bn* r, const char* h) {
  uint8_t b[72];
  size_t n = strlen(h) / 2;
  for (size_t i = 0; i < n; i++) {
    unsigned v;
    sscanf(h + 2 * i, "%2x", &v);
    b[i] = (uint8_t)v;
  }
  bn_from_be(r, b, n);
}
static int bn_cmp(const bn* a, const bn* b) {
  for (int i = BN_W - 1; i >= 0; i--)
    if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
  return 0;
}
static void bn_add(bn* r, const bn* a, const bn* b) { /* no overflow: values stay below 2^545 */
  uint64_t c = 0;
  for (int i = 0; i < BN_W; i++) {
    c += (uint64_t)a->w[i] + b->w[i];
    r->w[i] = (uint32_t)c;
    c >>= 32;
  }
}
static void bn_sub(bn* r, const bn* a, const bn* b) { /* a >= b */
  int64_t c = 0;
  for (int i = 0; i < BN_W; i++) {
    c += (int64_t)a->w[i] - b->w[i];
    r->w[i] = (uint32_t)c;
    c >>= 32;
  }
}
static void bn_addmod(bn* r, const bn* a, const bn* b, const bn* p) {
  bn t;
  bn_add(&t, a, b);
  if (bn_cmp(&t, p) >= 0) bn_sub(&t, &t, p);
  *r = t;
}
static void bn_submod(bn* r, const bn* a, const bn* b, const bn* p) {
  bn t;
  if (bn_cmp(a, b) >= 0) {
    bn_sub(&t, a, b);
  } else {
    bn_add(&t, a, p);
    bn_sub(&t, &t, b);
  }
  *r = t;
}
static void bn_mulmod(bn* r, const bn* a, const bn* b, const bn* p) {
  bn acc;
  memset(&acc, 0, sizeof acc);
  for (int bit = BN_W * 32 - 1; bit >= 0; bit--) {
    bn_addmod(&acc, &acc, &acc, p);
    if ((b->w[bit / 32] >> (bit % 32)) & 1) bn_addmod(&acc, &acc, a, p);
  }
  *r = acc;
}

typedef struct { const char* oid_hex; uint32_t bits; const char* p; const char* b; int insecure; } ec_curve;
static const ec_curve EC_CURVES[5] = {
    {"2a8648ce3d030107", 256, "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff",
     "5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b", 0},
    {"2b81040022", 384,
     "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffff0000000000000000ffffffff",
     "b3312fa7e23ee7e4988e056be3f82d19181d9c6efe8141120314088f5013875ac656398d8a2ed19d2a85c8edd3ec2aef", 0},
    {"2b81040023", 521,
     "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
     "0051953eb9618e1c9a1f929a21a0b68540eea2da725b99b315f3b8b489918ef109e156193951ec7e937b1652c0bd3bb1bf073573df883d2c34f1ef451fd46b503f00",
     0},
    {"2b81040021", 224, "ffffffffffffffffffffffffffffffff000000000000000000000001",
     "b4050a850c04b3abf54132565044b0b7d7bfd8ba270b39432355ffb4", 0},
    {"2a8648ce3d030101", 192, "fffffffffffffffffffffffffffffffeffffffffffffffff",
     "64210519e59c80e70fa7e9ab72243049feb8deecc146b9b1", 1}, /* CT-go's secp192r1: "insecure curve" nfe */
};

/* elliptic.Unmarshal(curve, data) != nil */
static int ec_unmarshal_ok(const ec_curve* c, const uint8_t* data, size_t n) {
  size_t bl = (c->bits + 7) / 8;
  if (n != 1 + 2 * bl || data[0] != 4) return 0;
  bn p, b, x, y, three, l, r, t;
  bn_from_hex(&p, c->p);
  bn_from_hex(&b, c->b);
  bn_from_be(&x, data + 1, bl);
  bn_from_be(&y, data + 1 + bl, bl);
  if (bn_cmp(&x, &p) >= 0 || bn_cmp(&y, &p) >= 0) return 0;
  memset(&three, 0, sizeof three);
  three.w[0] = 3;
  bn_mulmod(&l, &y, &y, &p);      /* y^2 */
  bn_mulmod(&r, &x, &x, &p);
  bn_mulmod(&r, &r, &x, &p);      /* x^3 */
  bn_mulmod(&t, &x, &three, &p);
  bn_submod(&r, &r, &t, &p);      /* - 3x */
  bn_addmod(&r, &r, &b, &p);      /* + b */
  return bn_cmp(&l, &r) == 0;
}

/* *big.Int / int at p inside [p, end) of buffer k: 1 = minimal, -1 = only the lax parse accepts it, 0 = error.
 * *sign = sign of the value, *len = content length, *after = end of the element. */
static int key_integer(const uint8_t* k, uint64_t p, uint64_t end, int* sign, uint32_t* len, uint64_t* after) {
  tlv t;
  if (!rd_tlv(k, p, end, &t) || t.tag != 0x02) return 0;
  uint64_t c = p + t.hl;
  int ci = check_integer(k, c, t.len);
  if (ci == 0) return 0;
  if (k[c] & 0x80) {
    *sign = -1;
  } else {
    *sign = 0;
    for (uint32_t i = 0; i < t.len; i++)
      if (k[c + i]) *sign = 1;
  }
  *len = t.len;
  *after = c + t.len;
  return ci;
}

static int hex_nibble(char c) { return c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10; }
static int bytes_are(const uint8_t* d, uint32_t n, const char* hex) {
  if (strlen(hex) != 2 * (size_t)n) return 0;
  for (uint32_t i = 0; i < n; i++)
    if (d[i] != (uint8_t)(hex_nibble(hex[2 * i]) << 4 | hex_nibble(hex[2 * i + 1]))) return 0;
  return 1;
}

/* spki = contents of the SubjectPublicKeyInfo SEQUENCE, already accepted by the structural walk.  Sets
 * out->spki_fatal (an error site, 0 = none) and ORs ORC_PK_* findings into out->spki_findings. */
static void check_public_key(const uint8_t* d, uint64_t k, uint64_t k_end, orc_cert* out) {
  tlv a, oid, par, bits;
  rd_tlv(d, k, k_end, &a);
  uint64_t x = k + a.hl, x_end = x + a.len;
  rd_tlv(d, x, x_end, &oid);
  const uint8_t* oc = d + x + oid.hl;
  x += oid.hl + oid.len;
  int has_par = x < x_end;
  uint64_t par_p = x;
  if (has_par) rd_tlv(d, x, x_end, &par);
  uint32_t par_total = has_par ? par.hl + par.len : 0;
  uint64_t bp = k + a.hl + a.len;
  rd_tlv(d, bp, k_end, &bits);
  uint64_t bc = bp + bits.hl;
  /* BitString.RightAlign */
  uint32_t n = bits.len - 1, shift = d[bc];
  uint8_t key_small[1032];
  uint8_t* key = (size_t)n + 8 <= sizeof key_small ? key_small : (uint8_t*)malloc((size_t)n + 8);
  memset(key + n, 0, 8);
  for (uint32_t i = 0; i < n; i++) {
    uint8_t cur = d[bc + 1 + i], prev = i ? d[bc + i] : 0;
    key[i] = shift ? (uint8_t)((prev << (8 - shift)) | (cur >> shift)) : cur;
  }
  int is_rsa = bytes_are(oc, oid.len, "2a864886f70d010101"), is_oaep = bytes_are(oc, oid.len, "2a864886f70d010107");
#define PKFAIL(site)            \
  do {                          \
    out->spki_fatal = (site);   \
    if (key != key_small) free(key); \
    return;                     \
  } while (0)
  if (is_rsa || is_oaep) {
    if (is_rsa && !(par_total == 2 && d[par_p] == 0x05 && d[par_p + 1] == 0x00)) out->spki_findings |= ORC_PK_RSA_PARAMS;
    tlv seq;
    if (!rd_tlv(key, 0, n, &seq) || seq.tag != 0x30) PKFAIL(80);
    if ((uint64_t)seq.hl + seq.len != n) PKFAIL(81); /* trailing data after RSA public key */
    uint64_t q = seq.hl, q_end = n, after;
    int sign;
    uint32_t len;
    int ci = key_integer(key, q, q_end, &sign, &len, &after);
    if (ci == 0) PKFAIL(82);
    if (ci < 0) out->spki_findings |= ORC_PK_LAX_INTEGER;
    if (sign <= 0) out->spki_findings |= ORC_PK_RSA_MODULUS;
    ci = key_integer(key, after, q_end, &sign, &len, &after);
    if (ci == 0) PKFAIL(83);
    if (len > 8) PKFAIL(84); /* parseInt64: integer too large */
    if (ci < 0) out->spki_findings |= ORC_PK_LAX_INTEGER;
    if (sign <= 0) PKFAIL(85); /* RSA public exponent is not a positive number */
  } else if (bytes_are(oc, oid.len, "2a8648ce380401")) {
    int sign;
    uint32_t len;
    uint64_t after;
    int ci = key_integer(key, 0, n, &sign, &len, &after);
    if (ci == 0) PKFAIL(86);
    if (after != n) PKFAIL(87); /* trailing data after DSA public key */
    if (ci < 0) out->spki_findings |= ORC_PK_LAX_INTEGER;
    if (sign <= 0) PKFAIL(88);
    if (!has_par || par.tag != 0x30) PKFAIL(89);
    uint64_t q = par_p + par.hl, q_end = par_p + par_total;
    for (int i = 0; i < 3; i++) {
      if (key_integer(d, q, q_end, &sign, &len, &q) != 1) PKFAIL(90); /* strict parse only */
      if (sign <= 0) PKFAIL(91);
    }
  } else if (bytes_are(oc, oid.len, "2a8648ce3d0201")) {
    if (!has_par || par.tag != 0x06) PKFAIL(92);
    const ec_curve* c = NULL;
    for (int i = 0; i < 5; i++)
      if (bytes_are(d + par_p + par.hl, par.len, EC_CURVES[i].oid_hex)) c = &EC_CURVES[i];
    if (!c) PKFAIL(93); /* unsupported elliptic curve (or not a well-formed OID at all) */
    if (c->insecure) out->spki_findings |= ORC_PK_INSECURE_CURVE;
    if (!ec_unmarshal_ok(c, key, n)) PKFAIL(94); /* failed to unmarshal elliptic curve point */
  }
#undef PKFAIL
  if (key != key_small) free(key);
}

/* strict_extensions (orc_engine_set_strict_extensions; ON by default since round 6, like the product).  Go 1.13 crypto/x509 parseCertificate unmarshals
 * the VALUE of some extensions and fails the certificate when that fails — restated here for the ones it parses with plain
 * encoding/asn1 struct rules (each followed by "x509: trailing data after X.509 …" when octets remain):
 *   2.5.29.15 keyUsage               var usageBits asn1.BitString
 *   2.5.29.14 subjectKeyIdentifier   var keyid []byte
 *   2.5.29.37 extKeyUsage            var keyUsage []asn1.ObjectIdentifier
 *   2.5.29.35 authorityKeyIdentifier struct { Id []byte `asn1:"optional,tag:0"` }
 *   2.5.29.32 certificatePolicies    []struct { Policy asn1.ObjectIdentifier }        (what follows the OID is ignored)
 *   1.3.6.1.5.5.7.1.1 authorityInfoAccess   []struct { Method asn1.ObjectIdentifier; Location asn1.RawValue }
 *   1.3.6.1.5.5.7.1.11 subjectInfoAccess    the same []accessDescription — CT-go's fork only (round 6; the standard library does
 *                                           not know the extension).  CT-go also refuses an EMPTY list of either kind
 *                                           ("x509: empty AuthorityInfoAccess / SubjectInfoAccess extension") — recalled by
 *                                           the round-5 review and by the author alike, verifiable by neither: modelled as fatal.
 * subjectAltName (URI / IP parsing), nameConstraints and cRLDistributionPoints (nested optional tags) are not restated.
 * Which of these errors CT-go's fork files as non-fatal cannot be verified here: recalled from the standard library, like
 * the string character sets.  Returns 0 = fine / not one of them, else an error site. */
static int ext_body_site(const uint8_t* d, const uint8_t* oid, uint32_t oid_len, uint64_t o, uint64_t o_end) {
  static const uint8_t AIA[7] = {0x2b, 0x06, 0x01, 0x05, 0x05, 0x07, 0x01};  /* id-pe; .1 = AIA, .11 = SIA */
  int arc = (oid_len == 3 && oid[0] == 0x55 && oid[1] == 0x1d) ? oid[2] : -1;
  int aia = oid_len == 8 && memcmp(oid, AIA, 7) == 0 && (oid[7] == 0x01 || oid[7] == 0x0b);
  tlv t;
  if (!(arc == 15 || arc == 14 || arc == 37 || arc == 35 || arc == 32 || aia)) return 0;
  if (!rd_tlv(d, o, o_end, &t)) return 100;                       /* the value is not one well-formed element */
  if (o + t.hl + (uint64_t)t.len != o_end) return 101;            /* trailing data */
  uint64_t c = o + t.hl, c_end = c + t.len;
  if (arc == 15) return (t.tag == 0x03 && bit_string_ok(d, c, t.len)) ? 0 : 102;
  if (arc == 14) return t.tag == 0x04 ? 0 : 103;
  if (t.tag != 0x30) return 104;
  if (arc == 35) {                                                /* optional first field: skipped when its tag differs */
    tlv f;
    if (c == c_end) return 0;
    if (!rd_hdr(d, c, c_end, &f)) return 105;
    if (f.tag == 0x80 && c + f.hl + (uint64_t)f.len > c_end) return 106;
    return 0;
  }
  if (aia && c == c_end) return 112;                              /* CT-go: "x509: empty …InfoAccess extension" */
  while (c < c_end) {                                             /* SEQUENCE OF */
    tlv e;
    if (!rd_tlv(d, c, c_end, &e)) return 107;
    uint64_t x = c + e.hl, x_end = x + e.len;
    if (arc == 37) {
      if (e.tag != 0x06 || !oid_ok(d, x, e.len)) return 108;
    } else {
      tlv id, loc;
      if (e.tag != 0x30) return 109;
      if (!rd_tlv(d, x, x_end, &id) || id.tag != 0x06 || !oid_ok(d, x + id.hl, id.len)) return 110;
      if (aia && !rd_tlv(d, x + id.hl + id.len, x_end, &loc)) return 111;
    }
    c = x_end;
  }
  return 0;
}

/* ------------------------------------------------------------------ strict_extensions, round 5: the three extensions whose
 * VALUE Go parses with code of its own (not a plain struct unmarshal), and CT-go's embedded SCT list.  Like everything at
 * the CT-go boundary: RECALLED from Go 1.13's crypto/x509, net/url, net and golang.org/x/crypto/cryptobyte (go.mod:24) and
 * from CT-go v1.1.0's fork of crypto/x509 — not verifiable on this machine.
 *
 * subjectAltName 2.5.29.17 — parseSANExtension → forEachSAN:
 *   asn1.Unmarshal(value, &seq RawValue): one element that fits; rest → "x509: trailing data after X.509 extension";
 *   !seq.IsCompound || seq.Tag != 16 || seq.Class != 0 → "bad SAN sequence"; then every element of seq.Bytes is
 *   unmarshalled as a RawValue (header parses, contents fit) and dispatched on v.Tag ALONE — the class is not looked at:
 *     1 rfc822Name, 2 dNSName   kept as they are (this toolchain checks no character set on the parse side)
 *     6 URI                     url.Parse must succeed and, when the URL has a host, domainToReverseLabels(host) → else fatal
 *     7 iPAddress               length 4 or 16 — CT-go files any other length as a NON-FATAL finding (the stdlib fails)
 *     anything else             ignored */

/* net/url (go1.13) — only what can make url.Parse fail, and the unescaped host it hands to domainToReverseLabels */
static int is_hex(uint8_t c) { return (c >= '0' && c <= '9') || ((c | 0x20) >= 'a' && (c | 0x20) <= 'f'); }
static int unhex(uint8_t c) { return c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10; }
static int is_alnum(uint8_t c) { return (c >= '0' && c <= '9') || ((c | 0x20) >= 'a' && (c | 0x20) <= 'z'); }
/* shouldEscape(c, encodeHost) == shouldEscape(c, encodeZone) */
static int host_should_escape(uint8_t c) {
  if (is_alnum(c)) return 0;
  return !c || !strchr("!$&'()*+,;=:[]<>\"-_.~", c);
}
enum { URL_MODE_PATH, URL_MODE_HOST, URL_MODE_ZONE };
/* unescape(s, mode): every '%' is followed by two hex digits; in a host "%XX" may only stand for a non-ASCII octet or be
 * "%25"; in a zone it may also stand for a space or an octet a host may hold; an ASCII octet of a host / zone must be one
 * shouldEscape lets through.  out (optional): the unescaped octets, *out_n their number. */
static int url_unescape(const uint8_t* s, uint32_t n, int mode, uint8_t* out, uint32_t* out_n) {
  uint32_t i = 0;
  while (i < n) {
    if (s[i] == '%') {
      if (i + 2 >= n || !is_hex(s[i + 1]) || !is_hex(s[i + 2])) return 0; /* EscapeError — `i+2 >= len(s)` as in the source */
      int is25 = s[i + 1] == '2' && s[i + 2] == '5';
      uint8_t v = (uint8_t)(unhex(s[i + 1]) << 4 | unhex(s[i + 2]));
      if (mode == URL_MODE_HOST && unhex(s[i + 1]) < 8 && !is25) return 0;
      if (mode == URL_MODE_ZONE && !is25 && v != ' ' && host_should_escape(v)) return 0;
      if (out) out[(*out_n)++] = v;
      i += 3;
    } else {
      if (mode != URL_MODE_PATH && s[i] != '+' && s[i] < 0x80 && host_should_escape(s[i])) return 0; /* InvalidHostError */
      if (out) out[(*out_n)++] = s[i];
      i++;
    }
  }
  return 1;
}
static int64_t idx_of(const uint8_t* s, uint32_t n, uint8_t c) {
  for (uint32_t i = 0; i < n; i++) if (s[i] == c) return i;
  return -1;
}
static int64_t last_idx_of(const uint8_t* s, uint32_t n, uint8_t c) {
  for (uint32_t i = n; i > 0; i--) if (s[i - 1] == c) return i - 1;
  return -1;
}
static int valid_optional_port(const uint8_t* s, uint32_t n) {
  if (n == 0) return 1;
  if (s[0] != ':') return 0;
  for (uint32_t i = 1; i < n; i++) if (s[i] < '0' || s[i] > '9') return 0;
  return 1;
}
/* x509.domainToReverseLabels(domain).ok: no empty label (no trailing dot, no "..") and every rune in 33..126 (an octet
 * >= 0x80 is, or decodes to, a rune above 126).  ONE LEADING DOT PASSES: the loop cuts labels off the end
 * (LastIndexByte; domain = domain[:i]) and stops when nothing is left, so the empty label in front of a leading dot is
 * never recorded — ".a" gives ["a"].  Later Go releases append "" when i == 0 ("domain is prefixed with an empty label");
 * the toolchain of go.mod and CT-go v1.1.0's fork predate that fix. */
static int domain_labels_ok(const uint8_t* s, uint32_t n) {
  if (n == 0) return 1; /* no labels at all: ok (the callers that need a host test len > 0 first) */
  if (s[n - 1] == '.') return 0;
  for (uint32_t i = 0; i < n; i++) {
    if (s[i] < 33 || s[i] > 126) return 0;
    if (i + 1 < n && s[i] == '.' && s[i + 1] == '.') return 0;
  }
  return 1;
}
/* parseHost(host): *out / *out_n = the unescaped host (with its port) */
static int url_parse_host(const uint8_t* h, uint32_t n, uint8_t* out, uint32_t* out_n) {
  if (n && h[0] == '[') {
    int64_t i = last_idx_of(h, n, ']');
    if (i < 0) return 0;                                         /* missing ']' in host */
    if (!valid_optional_port(h + i + 1, n - (uint32_t)i - 1)) return 0;
    int64_t zone = -1;
    for (int64_t z = 0; z + 3 <= i; z++) if (h[z] == '%' && h[z + 1] == '2' && h[z + 2] == '5') { zone = z; break; }
    if (zone >= 0)
      return url_unescape(h, (uint32_t)zone, URL_MODE_HOST, out, out_n) &&
             url_unescape(h + zone, (uint32_t)(i - zone), URL_MODE_ZONE, out, out_n) &&
             url_unescape(h + i, n - (uint32_t)i, URL_MODE_HOST, out, out_n);
  } else {
    int64_t i = last_idx_of(h, n, ':');
    if (i >= 0 && !valid_optional_port(h + i, n - (uint32_t)i)) return 0;
  }
  return url_unescape(h, n, URL_MODE_HOST, out, out_n);
}
static int valid_userinfo(const uint8_t* s, uint32_t n) {
  for (uint32_t i = 0; i < n; i++)
    if (!is_alnum(s[i]) && (!s[i] || !strchr("-._:~!$&'()*+,;=%@", s[i]))) return 0;
  return 1;
}
/* url.Parse(s) succeeds && (uri.Host == "" || domainToReverseLabels(uri.Host) ok) */
static int go_san_uri_ok(const uint8_t* s, uint32_t n) {
  /* Parse: u, frag = split(rawurl, "#", true) */
  int64_t hash = idx_of(s, n, '#');
  uint32_t un = hash < 0 ? n : (uint32_t)hash;
  const uint8_t* frag = hash < 0 ? s + n : s + hash + 1;
  uint32_t fragn = hash < 0 ? 0 : n - (uint32_t)hash - 1;
  uint8_t* host = (uint8_t*)malloc(n + 1);
  uint32_t hostn = 0;
  int ok = 0;
  /* parse(u, viaRequest = false) */
  for (uint32_t i = 0; i < un; i++) if (s[i] < 0x20 || s[i] == 0x7f) goto done; /* invalid control character in URL */
  if (un == 1 && s[0] == '*') { ok = 1; goto frag_check; }
  {
    /* getscheme */
    uint32_t rest0 = 0;
    int has_scheme = 0;
    for (uint32_t i = 0; i < un; i++) {
      uint8_t c = s[i];
      if (((c | 0x20) >= 'a' && (c | 0x20) <= 'z')) continue;
      if ((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.') {
        if (i == 0) break;
        continue;
      }
      if (c == ':') {
        if (i == 0) goto done; /* missing protocol scheme */
        has_scheme = 1;
        rest0 = i + 1;
      }
      break;
    }
    const uint8_t* rest = s + rest0;
    uint32_t rn = un - rest0;
    /* the query goes: a lone trailing '?' (ForceQuery) or everything from the first '?' */
    int64_t q = idx_of(rest, rn, '?');
    if (q >= 0) rn = (uint32_t)q;
    if (!(rn && rest[0] == '/')) {
      if (has_scheme) { ok = 1; goto frag_check; } /* opaque */
      int64_t colon = idx_of(rest, rn, ':'), slash = idx_of(rest, rn, '/');
      if (colon >= 0 && (slash < 0 || colon < slash)) goto done; /* first path segment in URL cannot contain colon */
    }
    if ((has_scheme || !(rn >= 3 && rest[0] == '/' && rest[1] == '/' && rest[2] == '/')) && rn >= 2 && rest[0] == '/' && rest[1] == '/') {
      const uint8_t* auth = rest + 2;
      int64_t sl = idx_of(auth, rn - 2, '/');
      uint32_t an = sl < 0 ? rn - 2 : (uint32_t)sl;
      /* parseAuthority */
      int64_t at = last_idx_of(auth, an, '@');
      if (!url_parse_host(auth + (at + 1), an - (uint32_t)(at + 1), host, &hostn)) goto done;
      if (at >= 0) {
        if (!valid_userinfo(auth, (uint32_t)at)) goto done;
        if (!url_unescape(auth, (uint32_t)at, URL_MODE_PATH, NULL, NULL)) goto done; /* user and password: escapes only */
      }
      rest = auth + an;
      rn = rn - 2 - an;
    }
    if (!url_unescape(rest, rn, URL_MODE_PATH, NULL, NULL)) goto done; /* setPath */
    ok = 1;
  }
frag_check:
  if (ok && fragn && !url_unescape(frag, fragn, URL_MODE_PATH, NULL, NULL)) ok = 0;
  if (ok && hostn && !domain_labels_ok(host, hostn)) ok = 0; /* "x509: cannot parse URI …: invalid domain" */
done:
  free(host);
  return ok;
}

static int ext_san_site(const uint8_t* d, uint64_t o, uint64_t o_end, int32_t* xf) {
  tlv t;
  if (!rd_tlv(d, o, o_end, &t)) return 120;
  if (o + t.hl + (uint64_t)t.len != o_end) return 121;           /* trailing data */
  if (t.tag != 0x30) return 122;                                  /* bad SAN sequence */
  uint64_t p = o + t.hl;
  while (p < o_end) {
    tlv g;
    if (!rd_tlv(d, p, o_end, &g)) return 123;
    uint64_t c = p + g.hl;
    int tagnum = (g.tag & 0x1f) == 0x1f ? -1 : (g.tag & 0x1f);    /* v.Tag; a high-tag-number form is >= 31 */
    if (tagnum == 6 && !go_san_uri_ok(d + c, g.len)) return 124;
    if (tagnum == 7 && g.len != 4 && g.len != 16) *xf |= ORC_XF_SAN_IP;
    p = c + g.len;
  }
  return 0;
}

/* cRLDistributionPoints 2.5.29.31 — asn1.Unmarshal(value, &[]distributionPoint) + "trailing data":
 *   distributionPoint     struct { DistributionPoint distributionPointName `optional,tag:0`; Reason asn1.BitString
 *                                  `optional,tag:1`; CRLIssuer asn1.RawValue `optional,tag:2` }
 *   distributionPointName struct { FullName []asn1.RawValue `optional,tag:0`; RelativeName pkix.RDNSequence `optional,tag:1` }
 * encoding/asn1's struct rules: the fields are taken IN ORDER; at each field the header at the current offset must
 * parse (unless the contents are used up); a field whose class/number (and, except for a RawValue, primitive/constructed
 * bit) does not match is skipped without consuming anything; a field that matches must fit; what is left behind the last
 * field is ignored.  uris (optional): collects the FullName elements with Tag == 6 in order (parseCertificate:
 * CRLDistributionPoints — again the tag number alone). */
static int dp_field_hdr(const uint8_t* d, uint64_t off, uint64_t end, tlv* t) { return off == end ? -1 : rd_hdr(d, off, end, t); }

static int ext_crldp_site(const uint8_t* d, uint64_t o, uint64_t o_end, orc_meta* uris, int deep, int32_t* sfind, int32_t* nfind) {
  tlv t;
  if (!rd_tlv(d, o, o_end, &t) || t.tag != 0x30) return 130;
  if (o + t.hl + (uint64_t)t.len != o_end) return 131;
  uint64_t p = o + t.hl;
  while (p < o_end) {                                             /* parseSequenceOf: SEQUENCE elements that fit */
    tlv dp;
    if (!rd_tlv(d, p, o_end, &dp) || dp.tag != 0x30) return 132;
    uint64_t off = p + dp.hl, end = off + dp.len;
    tlv f;
    int h = dp_field_hdr(d, off, end, &f);
    if (h == 0) return 133;
    if (h > 0 && f.tag == 0xa0) {                                 /* DistributionPoint */
      if (off + f.hl + (uint64_t)f.len > end) return 134;
      uint64_t n = off + f.hl, n_end = n + f.len;
      tlv g;
      int hn = dp_field_hdr(d, n, n_end, &g);
      if (hn == 0) return 135;
      if (hn > 0 && g.tag == 0xa0) {                              /* FullName []asn1.RawValue */
        if (n + g.hl + (uint64_t)g.len > n_end) return 136;
        uint64_t q = n + g.hl, q_end = q + g.len;
        while (q < q_end) {
          tlv nm;
          if (!rd_tlv(d, q, q_end, &nm)) return 137;
          if (uris && (nm.tag & 0x1f) == 6) {
            if (uris->n_crl < ORC_MAX_CRL) {
              uris->crl_off[uris->n_crl] = (uint32_t)(q + nm.hl);
              uris->crl_len[uris->n_crl] = nm.len;
            }
            uris->n_crl++;
          }
          q += nm.hl + nm.len;
        }
        n = q_end;
        hn = dp_field_hdr(d, n, n_end, &g);
        if (hn == 0) return 138;
      }
      if (hn > 0 && g.tag == 0xa1) {                              /* RelativeName pkix.RDNSequence */
        if (n + g.hl + (uint64_t)g.len > n_end) return 139;
        int site = 0; /* deep = 0 (orc_cert_meta: where the URIs lie): the name's own contents are not looked at */
        if (deep && !rdn_elements(d, n + g.hl, n + g.hl + g.len, NULL, NULL, &site, 140, sfind, nfind)) return site;
      }
      off += f.hl + f.len;
      h = dp_field_hdr(d, off, end, &f);
      if (h == 0) return 146;
    }
    if (h > 0 && f.tag == 0x81) {                                 /* Reason asn1.BitString */
      if (off + f.hl + (uint64_t)f.len > end || !bit_string_ok(d, off + f.hl, f.len)) return 147;
      off += f.hl + f.len;
      h = dp_field_hdr(d, off, end, &f);
      if (h == 0) return 148;
    }
    if (h > 0 && (f.tag == 0x82 || f.tag == 0xa2) && off + f.hl + (uint64_t)f.len > end) return 149; /* CRLIssuer asn1.RawValue */
    p += dp.hl + dp.len;
  }
  return 0;
}

/* nameConstraints 2.5.29.30 — parseNameConstraintsExtension reads it with golang.org/x/crypto/cryptobyte, whose ReadASN1
 * differs from encoding/asn1 in two ways that show: a tag is matched as the whole identifier octet, and the
 * high-tag-number form is refused outright.  PeekASN1Tag / ReadOptionalASN1 look at the first octet only.
 *   outer.ReadASN1(&toplevel, SEQUENCE) && outer.Empty()
 *   toplevel.ReadOptionalASN1(&permitted, [0] constructed) && toplevel.ReadOptionalASN1(&excluded, [1] constructed)
 *   && toplevel.Empty()                                      else "x509: invalid NameConstraints extension"
 *   neither present, or both empty                            →   "x509: empty name constraints extension"
 *   every subtree: ReadASN1(&seq, SEQUENCE) && seq.ReadAnyASN1(&value, &tag); minimum / maximum are not read
 *     [2] dNSName       IA5; without one leading '.', domainToReverseLabels
 *     [7] iPAddress     8 or 32 octets, the second half a contiguous mask
 *     [1] rfc822Name    IA5; with '@': parseRFC2821Mailbox; else without one leading '.', domainToReverseLabels
 *     [6] URI           IA5; not net.ParseIP; without one leading '.', domainToReverseLabels
 *     other tags        ignored */
static int cb_read(const uint8_t* d, uint64_t p, uint64_t end, tlv* t) {  /* cryptobyte readASN1 */
  if (end - p < 2) return 0;
  if ((d[p] & 0x1f) == 0x1f) return 0;
  return rd_tlv(d, p, end, t);
}
static int ip_mask_ok(const uint8_t* m, uint32_t n) {
  int seen_zero = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (seen_zero) {
      if (m[i]) return 0;
      continue;
    }
    if (m[i] == 0xff) continue;
    /* 0x00 0x80 0xc0 0xe0 0xf0 0xf8 0xfc 0xfe: ones then zeros */
    uint8_t inv = (uint8_t)~m[i];
    if ((inv & (inv + 1)) != 0) return 0;
    seen_zero = 1;
  }
  return 1;
}
/* net.ParseIP(s) != nil (go1.13: leading zeros in a dotted quad are fine, no zone) */
static int go_dtoi(const uint8_t* s, uint32_t n, uint32_t* v, uint32_t* used) {
  uint32_t x = 0, i = 0;
  for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
    x = x * 10 + (s[i] - '0');
    if (x >= 0xFFFFFF) return 0;
  }
  if (i == 0) return 0;
  *v = x; *used = i;
  return 1;
}
static int go_parse_ipv4(const uint8_t* s, uint32_t n) {
  for (int i = 0; i < 4; i++) {
    if (n == 0) return 0;
    if (i > 0) {
      if (s[0] != '.') return 0;
      s++; n--;
    }
    uint32_t v, c;
    if (!go_dtoi(s, n, &v, &c) || v > 0xFF) return 0;
    s += c; n -= c;
  }
  return n == 0;
}
static int go_xtoi(const uint8_t* s, uint32_t n, uint32_t* v, uint32_t* used) {
  uint32_t x = 0, i = 0;
  for (; i < n; i++) {
    if (!is_hex(s[i])) break;
    x = x * 16 + (uint32_t)unhex(s[i]);
    if (x >= 0xFFFFFF) return 0;
  }
  if (i == 0) return 0;
  *v = x; *used = i;
  return 1;
}
static int go_parse_ipv6(const uint8_t* s, uint32_t n) {
  int ellipsis = -1, i = 0;
  if (n >= 2 && s[0] == ':' && s[1] == ':') {
    ellipsis = 0;
    s += 2; n -= 2;
    if (n == 0) return 1;
  }
  while (i < 16) {
    uint32_t v, c;
    if (!go_xtoi(s, n, &v, &c) || v > 0xFFFF) return 0;
    if (c < n && s[c] == '.') {
      if (ellipsis < 0 && i != 12) return 0;
      if (i + 4 > 16) return 0;
      if (!go_parse_ipv4(s, n)) return 0;
      n = 0;
      i += 4;
      break;
    }
    i += 2;
    s += c; n -= c;
    if (n == 0) break;
    if (s[0] != ':' || n == 1) return 0;
    s++; n--;
    if (s[0] == ':') {
      if (ellipsis >= 0) return 0;
      ellipsis = i;
      s++; n--;
      if (n == 0) break;
    }
  }
  if (n != 0) return 0;
  if (i < 16) return ellipsis >= 0;
  return ellipsis < 0;
}
static int go_parse_ip(const uint8_t* s, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    if (s[i] == '.') return go_parse_ipv4(s, n);
    if (s[i] == ':') return go_parse_ipv6(s, n);
  }
  return 0;
}
/* x509.parseRFC2821Mailbox(in).ok */
static int go_mailbox_ok(const uint8_t* in, uint32_t n) {
  if (n == 0) return 0;
  uint32_t i = 0;
  if (in[0] == '"') {
    i = 1;
    for (;;) {
      if (i >= n) return 0;
      uint8_t c = in[i++];
      if (c == '"') break;
      if (c == '\\') {
        if (i >= n) return 0;
        uint8_t e = in[i];
        if (e == 11 || e == 12 || (e >= 1 && e <= 9) || (e >= 14 && e <= 127)) i++;
        else return 0;
      } else if (c == 11 || c == 12 || c == 32 || c == 33 || c == 127 || (c >= 1 && c <= 8) || (c >= 14 && c <= 31) ||
                 (c >= 35 && c <= 91) || (c >= 93 && c <= 126)) {
      } else {
        return 0;
      }
    }
  } else {
    uint32_t nlocal = 0;
    uint8_t first = 0, last = 0;
    int two_dots = 0;
    while (i < n) {
      uint8_t c = in[i];
      if (c == '\\') {
        i++;
        if (i >= n) return 0;
      } else if (!(is_alnum(c) || (c && strchr("!#$%&'*+-/=?^_`{|}~.", c)))) {
        break;
      }
      uint8_t b = in[i++];                     /* the octet that joins the local part: the escaped one after a backslash */
      if (nlocal == 0) first = b;
      if (nlocal && last == '.' && b == '.') two_dots = 1;
      last = b;
      nlocal++;
    }
    if (nlocal == 0) return 0;
    if (first == '.' || last == '.' || two_dots) return 0;
  }
  if (i >= n || in[i] != '@') return 0;
  i++;
  return domain_labels_ok(in + i, n - i);      /* (an empty domain has no labels: ok, as in the source) */
}
static int ia5_ok(const uint8_t* s, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) if (s[i] >= 0x80) return 0;
  return 1;
}
static int nc_subtrees_site(const uint8_t* d, uint64_t p, uint64_t end) {
  while (p < end) {
    tlv seq, v;
    if (!cb_read(d, p, end, &seq) || seq.tag != 0x30) return 155;
    uint64_t c = p + seq.hl, c_end = c + seq.len;
    if (!cb_read(d, c, c_end, &v)) return 156;
    const uint8_t* s = d + c + v.hl;
    uint32_t n = v.len;
    if (v.tag == 0x82) {
      if (!ia5_ok(s, n)) return 157;
      if (n && s[0] == '.') { s++; n--; }
      if (!domain_labels_ok(s, n)) return 158;
    } else if (v.tag == 0x87) {
      if (n != 8 && n != 32) return 159;
      if (!ip_mask_ok(s + n / 2, n / 2)) return 160;
    } else if (v.tag == 0x81) {
      if (!ia5_ok(s, n)) return 161;
      if (idx_of(s, n, '@') >= 0) {
        if (!go_mailbox_ok(s, n)) return 162;
      } else {
        if (n && s[0] == '.') { s++; n--; }
        if (!domain_labels_ok(s, n)) return 163;
      }
    } else if (v.tag == 0x86) {
      if (!ia5_ok(s, n)) return 164;
      if (go_parse_ip(s, n)) return 165;
      if (n && s[0] == '.') { s++; n--; }
      if (!domain_labels_ok(s, n)) return 166;
    }
    p = c_end;
  }
  return 0;
}
static int ext_nc_site(const uint8_t* d, uint64_t o, uint64_t o_end) {
  tlv top, f;
  if (!cb_read(d, o, o_end, &top) || top.tag != 0x30) return 150;
  if (o + top.hl + (uint64_t)top.len != o_end) return 150;
  uint64_t p = o + top.hl, end = o_end;
  int have_p = 0, have_e = 0;
  uint64_t ps = 0, pe = 0, es = 0, ee = 0;
  if (p < end && d[p] == 0xa0) {
    if (!cb_read(d, p, end, &f)) return 151;
    have_p = 1; ps = p + f.hl; pe = ps + f.len; p = pe;
  }
  if (p < end && d[p] == 0xa1) {
    if (!cb_read(d, p, end, &f)) return 152;
    have_e = 1; es = p + f.hl; ee = es + f.len; p = ee;
  }
  if (p != end) return 153;
  if ((!have_p && !have_e) || (pe == ps && ee == es)) return 154;   /* empty name constraints extension */
  int r = have_p ? nc_subtrees_site(d, ps, pe) : 0;
  if (!r && have_e) r = nc_subtrees_site(d, es, ee);
  return r;
}

/* RFC 3779 (round 6; CT-go ONLY — x509/rpki.go of certificate-transparency-go v1.1.0, go.mod:10; the standard library does not
 * know these extensions).  Recalled, not verifiable here: parseRPKIAddrBlocks / parseRPKIASIdentifiers decode the value with
 * plain (strict) asn1.Unmarshal calls and file EVERY failure with nfe.AddError — non-fatal: an X509 entry keeps its
 * certificate, a precertificate and a Chain[0] issuer are dropped (ct-fetch.go:202-209,221-225,452-459).
 *   sbgp-ipAddrBlock 1.3.6.1.5.5.7.1.7: []ipAddressFamily { AddressFamily []byte; Choice asn1.RawValue } filling the value;
 *     AddressFamily 2 or 3 octets; Choice == 05 00 (inherit), or else []asn1.RawValue whose elements are, by tag NUMBER
 *     alone, 3 → asn1.BitString (so: universal, primitive, parseBitString) or 16 → struct { Min, Max asn1.BitString };
 *     any other tag number is a finding.
 *   sbgp-autonomousSysNum 1.3.6.1.5.5.7.1.8: struct { ASNum RawValue `optional,tag:0`; RDI RawValue `optional,tag:1` }
 *     filling the value; each present choice's CONTENTS are 05 00 (inherit) or one []asn1.RawValue filling them whose
 *     elements are, by tag number, 2 → int (universal primitive INTEGER, minimal, at most 8 octets) or 16 → struct { Min, Max int }.
 * Returns 1 when rpki.go decodes the value without a finding. */
static int rpki_bit_string(const uint8_t* d, uint64_t p, uint64_t end, uint64_t* after) {  /* a field of type asn1.BitString */
  tlv t;
  if (!rd_tlv(d, p, end, &t) || t.tag != 0x03 || !bit_string_ok(d, p + t.hl, t.len)) return 0;
  *after = p + t.hl + t.len;
  return 1;
}
static int rpki_int(const uint8_t* d, uint64_t p, uint64_t end, uint64_t* after) {  /* a field of type int (64-bit): parseInt64 */
  tlv t;
  if (!rd_tlv(d, p, end, &t) || t.tag != 0x02 || check_integer(d, p + t.hl, t.len) != 1 || t.len > 8) return 0;
  *after = p + t.hl + t.len;
  return 1;
}
static int ext_ipaddr_ok(const uint8_t* d, uint64_t o, uint64_t o_end) {
  tlv t;
  if (!rd_tlv(d, o, o_end, &t) || t.tag != 0x30) return 0;
  if (o + t.hl + (uint64_t)t.len != o_end) return 0;              /* "trailing data after ipAddrBlocks extension" */
  int good = 1;
  uint64_t c = o + t.hl;
  /* pass 1: asn1.Unmarshal(data, &addrBlocks) as a whole — any failure is ONE finding and nothing else is looked at */
  for (uint64_t q = c; q < o_end;) {
    tlv f, af, ch;
    if (!rd_tlv(d, q, o_end, &f) || f.tag != 0x30) return 0;
    uint64_t x = q + f.hl, x_end = x + f.len;
    if (!rd_tlv(d, x, x_end, &af) || af.tag != 0x04) return 0;    /* AddressFamily []byte: universal primitive OCTET STRING */
    if (!rd_tlv(d, x + af.hl + af.len, x_end, &ch)) return 0;     /* Choice RawValue: must be there and fit */
    q = x_end;
  }
  /* pass 2: the loop over the blocks; each finding is filed and the loop goes on — one is enough here */
  for (uint64_t q = c; q < o_end;) {
    tlv f, af, ch;
    rd_tlv(d, q, o_end, &f);
    uint64_t x = q + f.hl, x_end = x + f.len;
    rd_tlv(d, x, x_end, &af);
    uint64_t cp = x + af.hl + af.len;
    rd_tlv(d, cp, x_end, &ch);
    q = x_end;
    if (af.len < 2 || af.len > 3) { good = 0; continue; }
    if (ch.tag == 0x05 && ch.hl == 2 && ch.len == 0) continue;     /* bytes.Equal(FullBytes, asn1.NullBytes) */
    if (ch.tag != 0x30) { good = 0; continue; }                    /* []asn1.RawValue: a universal SEQUENCE */
    uint64_t r = cp + ch.hl, r_end = r + ch.len;
    int list_ok = 1;
    for (uint64_t y = r; y < r_end;) {                             /* every element a RawValue that fits, else the Unmarshal fails */
      tlv e;
      if (!rd_tlv(d, y, r_end, &e)) { list_ok = 0; break; }
      y += e.hl + e.len;
    }
    if (!list_ok) { good = 0; continue; }
    for (uint64_t y = r; y < r_end;) {
      tlv e;
      uint64_t a;
      rd_tlv(d, y, r_end, &e);
      uint64_t e_end = y + e.hl + e.len;
      const int tn = e.tag & 0x1f;
      if (tn == 3) {
        if (!rpki_bit_string(d, y, e_end, &a)) good = 0;
      } else if (tn == 16) {                                        /* ipAddressRange { Min, Max asn1.BitString } */
        if (e.tag != 0x30 || !rpki_bit_string(d, y + e.hl, e_end, &a) || !rpki_bit_string(d, a, e_end, &a)) good = 0;
      } else {
        good = 0;
      }
      y = e_end;
    }
  }
  return good;
}
static int rpki_asid_choice_ok(const uint8_t* d, uint64_t c, uint64_t c_end) {  /* parseASIDChoice(val): val.Bytes = [c, c_end) */
  tlv t;
  if (c_end - c == 2 && d[c] == 0x05 && d[c + 1] == 0x00) return 1;  /* inherit */
  if (!rd_tlv(d, c, c_end, &t) || t.tag != 0x30) return 0;
  if (c + t.hl + (uint64_t)t.len != c_end) return 0;              /* "trailing data after ASIdentifiers.asIdsOrRanges" */
  uint64_t r = c + t.hl;
  for (uint64_t y = r; y < c_end;) {
    tlv e;
    if (!rd_tlv(d, y, c_end, &e)) return 0;
    y += e.hl + e.len;
  }
  int good = 1;
  for (uint64_t y = r; y < c_end;) {
    tlv e;
    uint64_t a;
    rd_tlv(d, y, c_end, &e);
    uint64_t e_end = y + e.hl + e.len;
    const int tn = e.tag & 0x1f;
    if (tn == 2) {
      if (!rpki_int(d, y, e_end, &a)) good = 0;
    } else if (tn == 16) {                                          /* ASIDRange { Min, Max int } */
      if (e.tag != 0x30 || !rpki_int(d, y + e.hl, e_end, &a) || !rpki_int(d, a, e_end, &a)) good = 0;
    } else {
      good = 0;
    }
    y = e_end;
  }
  return good;
}
static int ext_asnum_ok(const uint8_t* d, uint64_t o, uint64_t o_end) {
  tlv t, f;
  if (!rd_tlv(d, o, o_end, &t) || t.tag != 0x30) return 0;
  if (o + t.hl + (uint64_t)t.len != o_end) return 0;              /* "trailing data after ASIdentifiers extension" */
  uint64_t off = o + t.hl;
  int good = 1, have = 0;
  /* two optional fields in order: at each one the header at the current position must parse unless the contents are used
   * up; a field of another tag is skipped without consuming anything; a matching one must fit; what follows is ignored */
  for (int k = 0; k < 2; k++) {
    if (off == o_end) break;
    if (!rd_hdr(d, off, o_end, &f)) return 0;
    if (f.tag == (0x80 | k) || f.tag == (0xa0 | k)) {               /* RawValue `tag:k`: context class, any form */
      if (off + f.hl + (uint64_t)f.len > o_end) return 0;
      have |= 1 << k;
      if (!rpki_asid_choice_ok(d, off + f.hl, off + f.hl + f.len)) good = 0;
      off += f.hl + f.len;
    }
  }
  (void)have;
  return good;
}

/* CT-go only: the embedded SCT list, 1.3.6.1.4.1.11129.2.4.2 — asn1.Unmarshal(value, &RawSCT []byte), no rest, then
 * tls.Unmarshal(RawSCT, &SignedCertificateTimestampList{ SCTList []SerializedSCT `tls:"minlen:1,maxlen:65535"` }) with
 * SerializedSCT{ Val []byte `tls:"minlen:1,maxlen:65535"` }, no rest.  Every failure is an nfe.AddError: non-fatal. */
static int ext_sct_ok(const uint8_t* d, uint64_t o, uint64_t o_end) {
  tlv t;
  if (!rd_tlv(d, o, o_end, &t) || t.tag != 0x04) return 0;
  if (o + t.hl + (uint64_t)t.len != o_end) return 0;
  uint64_t p = o + t.hl;
  if (o_end - p < 2) return 0;
  uint32_t ll = (uint32_t)d[p] << 8 | d[p + 1];
  p += 2;
  if (ll < 1 || p + ll != o_end) return 0;
  while (p < o_end) {
    if (o_end - p < 2) return 0;
    uint32_t sl = (uint32_t)d[p] << 8 | d[p + 1];
    p += 2;
    if (sl < 1 || p + sl > o_end) return 0;
    p += sl;
  }
  return 1;
}

/* tbs_only: the buffer is a bare TBSCertificate — CT-go x509.ParseTBSCertificate, which ct.LogEntryFromLeaf applies to the
 * TBSCertificate of a precertificate entry's MerkleTreeLeaf (cmd/ct-fetch/ct-fetch.go:452): asn1.Unmarshal into
 * tbsCertificate, "trailing data" when anything follows it, then the same parseCertificate as for a whole certificate
 * (there is no signatureAlgorithm / signatureValue to look at).  CT-go v1.1.0 is not on this machine: recalled, unverified,
 * like the rest of the CT-go boundary. */
static void parse_impl(const uint8_t* d, size_t L, orc_cert* out, int tbs_only) {
  memset(out, 0, sizeof(*out));
  tlv t;
  int site = 0;
  if (L > 0x7fffffffu) FAIL(1);
  uint64_t cert_end = L;
  uint64_t p = 0;
  if (!tbs_only) {
    /* Certificate ::= SEQUENCE, no trailing data (x509.ParseCertificate) */
    if (!rd_tlv(d, 0, L, &t) || t.tag != 0x30) FAIL(2);
    if ((uint64_t)t.hl + t.len != L) FAIL(3);
    p = t.hl;
  }
  /* tbsCertificate */
  if (!rd_tlv(d, p, cert_end, &t) || t.tag != 0x30) FAIL(4);
  if (tbs_only && (uint64_t)t.hl + t.len != L) FAIL(3);
  out->tbs_off = (uint32_t)p;
  out->tbs_len = t.hl + t.len;
  uint64_t tbs_end = p + t.hl + t.len;
  uint64_t q = p + t.hl;
  /* Version int `asn1:"optional,explicit,default:0,tag:0"`.  Go resumes behind the INNER integer; the wrapper's
   * length is only used to tell "empty" (an error: not an asn1.Flag) from "has an element". */
  if (q < tbs_end && d[q] == 0xa0) {
    if (!rd_hdr(d, q, tbs_end, &t)) FAIL(5);
    if (t.len == 0) FAIL(5);
    tlv v;
    uint64_t vq = q + t.hl;
    if (!rd_tlv(d, vq, tbs_end, &v) || v.tag != 0x02) FAIL(6);
    int ci = check_int32(d, vq + v.hl, v.len);
    if (ci == 0) FAIL(8);
    if (ci < 0) out->nonfatal |= ORC_NF_LAX_INTEGER;
    q = vq + v.hl + v.len;
  }
  /* serialNumber INTEGER: raw content octets kept verbatim (types.go:165-178) */
  if (!rd_tlv(d, q, tbs_end, &t) || t.tag != 0x02) FAIL(9);
  {
    int ci = check_integer(d, q + t.hl, t.len);
    if (ci == 0) FAIL(10);
    if (ci < 0) out->nonfatal |= ORC_NF_LAX_INTEGER;
    if (d[q + t.hl] & 0x80) out->nonfatal |= ORC_NF_NEGATIVE_SERIAL; /* CT-go: "x509: negative serial number", non-fatal */
  }
  out->serial_off = (uint32_t)(q + t.hl);
  out->serial_len = t.len;
  q += t.hl + t.len;
  /* signature AlgorithmIdentifier */
  if (!alg_id(d, q, tbs_end, &t, &site, 11)) FAIL(site);
  q += t.hl + t.len;
  /* issuer Name (asn1.RawValue, then asn1.Unmarshal into pkix.RDNSequence) */
  if (!rdn_sequence(d, q, tbs_end, &t, &out->cn_off, &out->cn_len, &site, 50, &out->string_findings, &out->nonfatal)) FAIL(site);
  out->issuer_off = (uint32_t)q;
  out->issuer_len = t.hl + t.len;
  q += t.hl + t.len;
  /* validity SEQUENCE { notBefore Time, notAfter Time } */
  if (!rd_tlv(d, q, tbs_end, &t) || t.tag != 0x30) FAIL(17);
  {
    uint64_t v = q + t.hl, v_end = q + t.hl + t.len;
    tlv tm;
    if (!rd_tlv(d, v, v_end, &tm)) FAIL(18);
    if (!parse_time(d, &tm, v + tm.hl, &out->not_before)) FAIL(19);
    v += tm.hl + tm.len;
    if (!rd_tlv(d, v, v_end, &tm)) FAIL(20);
    if (!parse_time(d, &tm, v + tm.hl, &out->not_after)) FAIL(21);
  }
  q += t.hl + t.len;
  /* subject Name: same structure; no field of it is consumed */
  if (!rdn_sequence(d, q, tbs_end, &t, NULL, NULL, &site, 60, &out->string_findings, &out->nonfatal)) FAIL(site);
  q += t.hl + t.len;
  /* subjectPublicKeyInfo: full TLV = RawSubjectPublicKeyInfo (types.go:109-115);
   * publicKeyInfo ::= SEQUENCE { algorithm AlgorithmIdentifier, publicKey BIT STRING } */
  if (!rd_tlv(d, q, tbs_end, &t) || t.tag != 0x30) FAIL(23);
  out->spki_off = (uint32_t)q;
  out->spki_len = t.hl + t.len;
  {
    uint64_t k = q + t.hl, k_end = k + t.len;
    tlv a;
    if (!alg_id(d, k, k_end, &a, &site, 70)) FAIL(site);
    k += a.hl + a.len;
    if (!rd_tlv(d, k, k_end, &a) || a.tag != 0x03 || !bit_string_ok(d, k + a.hl, a.len)) FAIL(73);
    check_public_key(d, q + t.hl, k_end, out); /* parsePublicKey: spki_fatal / spki_findings, applied by the engine (strict_spki) */
  }
  q += t.hl + t.len;
  /* UniqueId, SubjectUniqueId asn1.BitString `optional,tag:1|2`; Extensions `optional,explicit,tag:3`.  Each
   * optional field parses the header at the current position (which must therefore be a valid header) and skips
   * itself when the tag is not its own; whatever is left in the TBSCertificate after the three is ignored. */
  if (q < tbs_end) {
    if (!rd_hdr(d, q, tbs_end, &t)) FAIL(24);
    if (t.tag == 0x81) {
      if (q + t.hl + (uint64_t)t.len > tbs_end || !bit_string_ok(d, q + t.hl, t.len)) FAIL(24);
      q += t.hl + t.len;
    }
  }
  if (q < tbs_end) {
    if (!rd_hdr(d, q, tbs_end, &t)) FAIL(25);
    if (t.tag == 0x82) {
      if (q + t.hl + (uint64_t)t.len > tbs_end || !bit_string_ok(d, q + t.hl, t.len)) FAIL(25);
      q += t.hl + t.len;
    }
  }
  if (q < tbs_end) {
    if (!rd_hdr(d, q, tbs_end, &t)) FAIL(26);
    /* [3] matches when constructed or empty; empty is an error ("zero length explicit tag was not an asn1.Flag") */
    if ((t.tag == 0xa3 || t.tag == 0x83) && t.len == 0) FAIL(26);
  }
  if (q < tbs_end && t.tag == 0xa3) {
    tlv seq;
    uint64_t e0 = q + t.hl;
    if (!rd_hdr(d, e0, tbs_end, &seq)) FAIL(27);
    /* an inner element that is not a SEQUENCE leaves the optional field unset: no extensions at all */
    if (seq.tag == 0x30) {
      if (e0 + seq.hl + (uint64_t)seq.len > tbs_end) FAIL(27);
      uint64_t e = e0 + seq.hl, e_end = e0 + seq.hl + seq.len;
      out->exts_off = (uint32_t)e;
      out->exts_end = (uint32_t)e_end;
      while (e < e_end) {
        tlv ext, oid, val;
        if (!rd_tlv(d, e, e_end, &ext) || ext.tag != 0x30) FAIL(28);
        uint64_t x = e + ext.hl, x_end = e + ext.hl + ext.len;
        if (!rd_tlv(d, x, x_end, &oid) || oid.tag != 0x06 || !oid_ok(d, x + oid.hl, oid.len)) FAIL(29);
        uint64_t oid_c = x + oid.hl;
        x += oid.hl + oid.len;
        if (!rd_hdr(d, x, x_end, &val)) FAIL(30);
        if (val.tag == 0x01) { /* critical BOOLEAN DEFAULT FALSE */
          if (x + val.hl + (uint64_t)val.len > x_end) FAIL(30);
          if (val.len != 1) FAIL(31);
          uint8_t bv = d[x + val.hl];
          if (bv != 0x00 && bv != 0xff) FAIL(32);
          x += val.hl + val.len;
          if (!rd_hdr(d, x, x_end, &val)) FAIL(33);
        }
        if (val.tag != 0x04) FAIL(34);
        if (x + val.hl + (uint64_t)val.len > x_end) FAIL(34);
        if (!out->ext_fatal) /* the first one in certificate order; applied by the engine (strict_extensions) */
          out->ext_fatal = (uint32_t)ext_body_site(d, d + oid_c, oid.len, x + val.hl, x + val.hl + val.len);
        if (!out->ext_fatal) { /* round 5: subjectAltName, nameConstraints, cRLDistributionPoints; CT-go's SCT list */
          static const uint8_t SCT[10] = {0x2b, 0x06, 0x01, 0x04, 0x01, 0xd6, 0x79, 0x02, 0x04, 0x02};
          const uint64_t o = x + val.hl, o_end = o + val.len;
          const int arc = (oid.len == 3 && d[oid_c] == 0x55 && d[oid_c + 1] == 0x1d) ? d[oid_c + 2] : -1;
          if (arc == 17) out->ext_fatal = (uint32_t)ext_san_site(d, o, o_end, &out->ext_findings);
          else if (arc == 30) out->ext_fatal = (uint32_t)ext_nc_site(d, o, o_end);
          else if (arc == 31) {
            int32_t lax = 0;
            out->ext_fatal = (uint32_t)ext_crldp_site(d, o, o_end, NULL, 1, &out->ext_string_findings, &lax);
            if (lax) out->ext_findings |= ORC_XF_LAX;
          }
          else if (oid.len == 10 && memcmp(d + oid_c, SCT, 10) == 0 && !ext_sct_ok(d, o, o_end)) out->ext_findings |= ORC_XF_SCT;
          else if (oid.len == 8 && memcmp(d + oid_c, "\x2b\x06\x01\x05\x05\x07\x01", 7) == 0 &&
                   (d[oid_c + 7] == 0x07 || d[oid_c + 7] == 0x08)) {  /* RFC 3779: sbgp-ipAddrBlock, sbgp-autonomousSysNum (CT-go: non-fatal) */
            if (!(d[oid_c + 7] == 0x07 ? ext_ipaddr_ok(d, o, o_end) : ext_asnum_ok(d, o, o_end))) out->ext_findings |= ORC_XF_RPKI;
          }
        }
        if (oid.len == 3 && d[oid_c] == 0x55 && d[oid_c + 1] == 0x1d && d[oid_c + 2] == 0x13) {
          /* basicConstraints struct { IsCA bool `optional`; MaxPathLen int `optional,default:-1` } must be the
           * whole OCTET STRING ("x509: trailing data after X.509 BasicConstraints"); inside the SEQUENCE an
           * element of another type leaves the optional field at its default, and trailing elements are ignored */
          tlv bc;
          uint64_t o = x + val.hl, o_end = x + val.hl + val.len;
          if (!rd_tlv(d, o, o_end, &bc) || bc.tag != 0x30) FAIL(35);
          if ((uint64_t)bc.hl + bc.len != val.len) FAIL(36);
          uint64_t c = o + bc.hl, c_end = o + bc.hl + bc.len;
          int ca = 0;
          if (c < c_end) {
            tlv f;
            if (!rd_hdr(d, c, c_end, &f)) FAIL(37);
            if (f.tag == 0x01) {
              if (c + f.hl + (uint64_t)f.len > c_end) FAIL(37);
              if (f.len != 1) FAIL(38);
              uint8_t bv = d[c + f.hl];
              if (bv != 0x00 && bv != 0xff) FAIL(39);
              ca = bv == 0xff;
              c += f.hl + f.len;
              if (c < c_end && !rd_hdr(d, c, c_end, &f)) FAIL(40);
            }
            if (c < c_end && f.tag == 0x02) { /* pathLenConstraint */
              if (c + f.hl + (uint64_t)f.len > c_end) FAIL(41);
              int ci = check_int32(d, c + f.hl, f.len);
              if (ci == 0) FAIL(41);
              if (ci < 0) out->nonfatal |= ORC_NF_LAX_INTEGER;
            }
          }
          out->bc_valid = 1;
          out->is_ca = ca; /* a repeated extension overwrites (last wins) */
        }
        e += ext.hl + ext.len;
      }
    }
  }
  if (!tbs_only) {
    /* signatureAlgorithm, signatureValue BIT STRING; whatever follows inside the Certificate is ignored */
    p = tbs_end;
    if (!alg_id(d, p, cert_end, &t, &site, 42)) FAIL(site);
    p += t.hl + t.len;
    if (!rd_tlv(d, p, cert_end, &t) || t.tag != 0x03) FAIL(45);
    if (!bit_string_ok(d, p + t.hl, t.len)) FAIL(46);
  }
  out->ok = 1;
}

void orc_parse_cert(const uint8_t* d, size_t L, orc_cert* out) { parse_impl(d, L, out, 0); }
void orc_parse_tbs(const uint8_t* d, size_t L, orc_cert* out) { parse_impl(d, L, out, 1); }

/* ------------------------------------------------------------------ SHA-256 */

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(uint32_t h[8], const uint8_t* blk) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
           ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
    uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t i = 0;
  for (; i + 64 <= len; i += 64) sha256_block(h, msg + i);
  uint8_t tail[128];
  size_t rem = len - i;
  memset(tail, 0, sizeof(tail));
  memcpy(tail, msg + i, rem);
  tail[rem] = 0x80;
  size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
  sha256_block(h, tail);
  if (tl == 128) sha256_block(h, tail + 64);
  for (int k = 0; k < 8; k++) {
    out[4 * k] = (uint8_t)(h[k] >> 24);
    out[4 * k + 1] = (uint8_t)(h[k] >> 16);
    out[4 * k + 2] = (uint8_t)(h[k] >> 8);
    out[4 * k + 3] = (uint8_t)h[k];
  }
}

/* ------------------------------------------------------------------ identities */

size_t orc_b64url(const uint8_t* in, size_t n, char* out) {
  static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_";
  size_t o = 0, i = 0;
  for (; i + 3 <= n; i += 3) {
    uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8) | in[i + 2];
    out[o++] = A[(v >> 18) & 63];
    out[o++] = A[(v >> 12) & 63];
    out[o++] = A[(v >> 6) & 63];
    out[o++] = A[v & 63];
  }
  if (n - i == 1) {
    uint32_t v = (uint32_t)in[i] << 16;
    out[o++] = A[(v >> 18) & 63];
    out[o++] = A[(v >> 12) & 63];
    out[o++] = '=';
    out[o++] = '=';
  } else if (n - i == 2) {
    uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8);
    out[o++] = A[(v >> 18) & 63];
    out[o++] = A[(v >> 12) & 63];
    out[o++] = A[(v >> 6) & 63];
    out[o++] = '=';
  }
  out[o] = 0;
  return o;
}

/* pem.EncodeToMemory(&pem.Block{Type: "CERTIFICATE", Headers: {}, Bytes: der}) as called by
 * FilesystemDatabase.Store (storage/filesystemdatabase.go:167-175,196-200).  Go stdlib encoding/pem
 * Encode: "-----BEGIN " Type "-----\n", no header block when len(Headers) == 0, then base64.StdEncoding
 * through a lineBreaker (64 characters then "\n"; Close() flushes a non-empty last line plus "\n"),
 * then "-----END " Type "-----\n".  Returns the length; out needs 28 + 4*ceil(n/3)*(65/64)+1 + 26. */
size_t orc_pem_encode(const uint8_t* der, size_t n, char* out) {
  static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  size_t o = 0, col = 0, i = 0;
  memcpy(out + o, "-----BEGIN CERTIFICATE-----\n", 28);
  o += 28;
#define PUTC(ch)                 \
  do {                           \
    out[o++] = (ch);             \
    if (++col == 64) {           \
      out[o++] = '\n';           \
      col = 0;                   \
    }                            \
  } while (0)
  for (; i + 3 <= n; i += 3) {
    uint32_t v = ((uint32_t)der[i] << 16) | ((uint32_t)der[i + 1] << 8) | der[i + 2];
    PUTC(A[(v >> 18) & 63]);
    PUTC(A[(v >> 12) & 63]);
    PUTC(A[(v >> 6) & 63]);
    PUTC(A[v & 63]);
  }
  if (n - i == 1) {
    uint32_t v = (uint32_t)der[i] << 16;
    PUTC(A[(v >> 18) & 63]);
    PUTC(A[(v >> 12) & 63]);
    PUTC('=');
    PUTC('=');
  } else if (n - i == 2) {
    uint32_t v = ((uint32_t)der[i] << 16) | ((uint32_t)der[i + 1] << 8);
    PUTC(A[(v >> 18) & 63]);
    PUTC(A[(v >> 12) & 63]);
    PUTC(A[(v >> 6) & 63]);
    PUTC('=');
  }
#undef PUTC
  if (col) out[o++] = '\n';
  memcpy(out + o, "-----END CERTIFICATE-----\n", 26);
  o += 26;
  return o;
}

void orc_issuer_id(const uint8_t* spki, size_t n, char out[45]) {
  uint8_t dg[32];
  orc_sha256(spki, n, dg);
  orc_b64url(dg, 32, out);
}

int32_t orc_exp_hour(int64_t s) {
  int64_t h = s / 3600;
  if (s % 3600 < 0) h -= 1; /* Truncate rounds toward the zero Time, i.e. floor */
  return (int32_t)h;
}

static void civil_from_days(int64_t z, int64_t* y, int* m, int* d) {
  z += 719468;
  int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  int64_t doe = z - era * 146097;
  int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t yy = yoe + era * 400;
  int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  int64_t mp = (5 * doy + 2) / 153;
  *d = (int)(doy - (153 * mp + 2) / 5 + 1);
  *m = (int)(mp < 10 ? mp + 3 : mp - 9);
  *y = yy + (*m <= 2);
}

void orc_exp_date_id(int32_t exp_hour, char out[16]) {
  int64_t days = exp_hour / 24;
  int hh = exp_hour % 24;
  if (hh < 0) {
    hh += 24;
    days -= 1;
  }
  int64_t y;
  int m, d;
  civil_from_days(days, &y, &m, &d);
  snprintf(out, 16, "%04lld-%02d-%02d-%02d", (long long)y, m, d, hh);
}

void orc_day_id(int64_t s, char out[16]) {
  int64_t days = s / 86400;
  if (s % 86400 < 0) days -= 1;
  int64_t y;
  int m, d;
  civil_from_days(days, &y, &m, &d);
  snprintf(out, 16, "%04lld-%02d-%02d", (long long)y, m, d);
}

/* certIsFilteredOut — cmd/ct-fetch/ct-fetch.go:44-70 */
int orc_cert_is_filtered_out(const uint8_t* der, const orc_cert* c, const char* filter,
                             size_t filter_len, int log_expired, int64_t now) {
  if (c->bc_valid && c->is_ca) return ORC_ST_FILTERED_CA; /* :47-50 */
  if (c->not_after < now && !log_expired) return ORC_ST_FILTERED_EXPIRED; /* :52-55 */
  int skip = filter_len != 0; /* :57 */
  /* strings.Split(filter, ","): pieces are not trimmed; "" yields one empty piece, which
   * HasPrefix matches — irrelevant because skip is already false for an empty filter. */
  size_t s = 0;
  for (;;) {
    size_t e = s;
    while (e < filter_len && filter[e] != ',') e++;
    size_t pl = e - s;
    if (pl <= c->cn_len && memcmp(der + c->cn_off, filter + s, pl) == 0) { /* HasPrefix :59 */
      skip = 0;
      break;
    }
    if (e >= filter_len) break;
    s = e + 1;
  }
  return skip ? ORC_ST_FILTERED_CN : ORC_ST_PASS;
}

/* ------------------------------------------------------------------ set store
 * Stands in for storage.RemoteCache set semantics (mockcache.go:38-61: sorted insert = set). */

typedef struct {
  uint8_t* bytes;   /* arena of [u32 len][data] */
  size_t used, cap;
  uint64_t* slots;  /* open addressing: arena offset+1, 0 = empty */
  size_t nslots, count;
} byteset;

static uint64_t fnv1a(const uint8_t* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  h ^= h >> 29;
  h *= 0xbf58476d1ce4e5b9ull;
  h ^= h >> 32;
  return h;
}

static void bs_init(byteset* s) {
  memset(s, 0, sizeof(*s));
  s->nslots = 16;
  s->slots = (uint64_t*)calloc(s->nslots, sizeof(uint64_t));
}
static void bs_free(byteset* s) {
  free(s->bytes);
  free(s->slots);
}
static void bs_rehash(byteset* s) {
  size_t nn = s->nslots * 2;
  uint64_t* ns = (uint64_t*)calloc(nn, sizeof(uint64_t));
  for (size_t i = 0; i < s->nslots; i++) {
    uint64_t o = s->slots[i];
    if (!o) continue;
    uint32_t len;
    memcpy(&len, s->bytes + o - 1, 4);
    size_t j = fnv1a(s->bytes + o - 1 + 4, len) & (nn - 1);
    while (ns[j]) j = (j + 1) & (nn - 1);
    ns[j] = o;
  }
  free(s->slots);
  s->slots = ns;
  s->nslots = nn;
}
/* returns index of member (arena offset) or -1; inserts when ins!=0; *was_new set */
static int64_t bs_find(byteset* s, const uint8_t* m, size_t n, int ins, int* was_new) {
  if (was_new) *was_new = 0;
  size_t j = fnv1a(m, n) & (s->nslots - 1);
  while (s->slots[j]) {
    uint64_t o = s->slots[j] - 1;
    uint32_t len;
    memcpy(&len, s->bytes + o, 4);
    if (len == n && memcmp(s->bytes + o + 4, m, n) == 0) return (int64_t)o;
    j = (j + 1) & (s->nslots - 1);
  }
  if (!ins) return -1;
  if (s->used + 4 + n > s->cap) {
    size_t nc = s->cap ? s->cap * 2 : 256;
    while (nc < s->used + 4 + n) nc *= 2;
    s->bytes = (uint8_t*)realloc(s->bytes, nc);
    s->cap = nc;
  }
  uint64_t o = s->used;
  uint32_t len = (uint32_t)n;
  memcpy(s->bytes + o, &len, 4);
  memcpy(s->bytes + o + 4, m, n);
  s->used += 4 + n;
  s->slots[j] = o + 1;
  s->count++;
  if (was_new) *was_new = 1;
  if (s->count * 2 > s->nslots) bs_rehash(s);
  return (int64_t)o;
}

typedef struct {
  byteset members;
  int has_expiry;
  int64_t expiry;
  uint64_t key_off; /* offset of the key string in engine->keys arena */
} setrec;

struct orc_engine {
  byteset keys;   /* key strings; arena offset identifies the key */
  setrec* sets;   /* parallel array indexed by insertion order */
  size_t nsets, capsets;
  uint64_t* key_to_set; /* map arena offset -> set index: small open-addressing table */
  size_t k2s_n;
  char* filter;
  size_t filter_len;
  int log_expired;
  int strict_strings; /* the stdlib's character-set rules for the Names' string values, as non-fatal findings (orc_engine_set_strict_strings) */
  int strict_ext;  /* the bodies of the extensions Go unmarshals (orc_engine_set_strict_extensions; ON by default since round 6, like the product): fatal */
  int strict_spki; /* parsePublicKey's verdict on the key inside subjectPublicKeyInfo (orc_engine_set_strict_spki; ON by default) */
  int strict_leaf; /* LogEntryFromLeaf's parse of a precertificate entry's leaf TBSCertificate (orc_engine_set_strict_leaf) */
  int64_t now;
  int64_t inserted;
  int64_t* sorted; /* lazily built sorted key index */
  size_t sorted_n;
};

orc_engine* orc_engine_new(const char* filter, size_t filter_len, int log_expired, int64_t now) {
  orc_engine* e = (orc_engine*)calloc(1, sizeof(*e));
  bs_init(&e->keys);
  e->filter = (char*)malloc(filter_len + 1);
  if (filter_len) memcpy(e->filter, filter, filter_len);
  e->filter[filter_len] = 0;
  e->filter_len = filter_len;
  e->log_expired = log_expired;
  /* the reference profile — the product's default since round 6 (ctmr_create = CTMR_PROFILE_REFERENCE): what
   * x509.ParseCertificate / ct.LogEntryFromLeaf decide at cmd/ct-fetch/ct-fetch.go:202-209,221-225,452-459 */
  e->strict_spki = e->strict_strings = e->strict_ext = e->strict_leaf = 1;
  e->now = now;
  e->k2s_n = 64;
  e->key_to_set = (uint64_t*)calloc(e->k2s_n * 2, sizeof(uint64_t));
  return e;
}

void orc_engine_free(orc_engine* e) {
  if (!e) return;
  for (size_t i = 0; i < e->nsets; i++) bs_free(&e->sets[i].members);
  free(e->sets);
  bs_free(&e->keys);
  free(e->key_to_set);
  free(e->filter);
  free(e->sorted);
  free(e);
}

static setrec* get_set(orc_engine* e, const char* key, size_t key_len, int create) {
  int was_new = 0;
  int64_t off = bs_find(&e->keys, (const uint8_t*)key, key_len, create, &was_new);
  if (off < 0) return NULL;
  if (was_new) {
    if (e->nsets == e->capsets) {
      e->capsets = e->capsets ? e->capsets * 2 : 64;
      e->sets = (setrec*)realloc(e->sets, e->capsets * sizeof(setrec));
    }
    setrec* s = &e->sets[e->nsets];
    bs_init(&s->members);
    s->has_expiry = 0;
    s->expiry = 0;
    s->key_off = (uint64_t)off;
    /* map off -> index */
    if ((e->nsets + 1) * 2 > e->k2s_n) {
      size_t nn = e->k2s_n * 2;
      uint64_t* nt = (uint64_t*)calloc(nn * 2, sizeof(uint64_t));
      for (size_t i = 0; i < e->k2s_n; i++) {
        if (!e->key_to_set[2 * i]) continue;
        size_t j = (e->key_to_set[2 * i] * 0x9e3779b97f4a7c15ull >> 20) & (nn - 1);
        while (nt[2 * j]) j = (j + 1) & (nn - 1);
        nt[2 * j] = e->key_to_set[2 * i];
        nt[2 * j + 1] = e->key_to_set[2 * i + 1];
      }
      free(e->key_to_set);
      e->key_to_set = nt;
      e->k2s_n = nn;
    }
    uint64_t tag = (uint64_t)off + 1;
    size_t j = (tag * 0x9e3779b97f4a7c15ull >> 20) & (e->k2s_n - 1);
    while (e->key_to_set[2 * j]) j = (j + 1) & (e->k2s_n - 1);
    e->key_to_set[2 * j] = tag;
    e->key_to_set[2 * j + 1] = e->nsets;
    e->nsets++;
    free(e->sorted);
    e->sorted = NULL;
    return s;
  }
  uint64_t tag = (uint64_t)off + 1;
  size_t j = (tag * 0x9e3779b97f4a7c15ull >> 20) & (e->k2s_n - 1);
  while (e->key_to_set[2 * j] != tag) j = (j + 1) & (e->k2s_n - 1);
  return &e->sets[e->key_to_set[2 * j + 1]];
}

int orc_set_insert(orc_engine* e, const char* key, size_t key_len, const uint8_t* m, size_t n) {
  setrec* s = get_set(e, key, key_len, 1);
  int was_new = 0;
  bs_find(&s->members, m, n, 1, &was_new);
  return was_new;
}

int orc_set_contains(orc_engine* e, const char* key, size_t key_len, const uint8_t* m, size_t n) {
  setrec* s = get_set(e, key, key_len, 0);
  if (!s) return 0;
  return bs_find(&s->members, m, n, 0, NULL) >= 0;
}

int64_t orc_set_cardinality(orc_engine* e, const char* key, size_t key_len) {
  setrec* s = get_set(e, key, key_len, 0);
  return s ? (int64_t)s->members.count : 0;
}

int64_t orc_key_count(orc_engine* e) { return (int64_t)e->nsets; }

static orc_engine* g_sort_engine;
static int cmp_keys(const void* a, const void* b) {
  const setrec* sa = &g_sort_engine->sets[*(const int64_t*)a];
  const setrec* sb = &g_sort_engine->sets[*(const int64_t*)b];
  uint32_t la, lb;
  memcpy(&la, g_sort_engine->keys.bytes + sa->key_off, 4);
  memcpy(&lb, g_sort_engine->keys.bytes + sb->key_off, 4);
  int c = memcmp(g_sort_engine->keys.bytes + sa->key_off + 4,
                 g_sort_engine->keys.bytes + sb->key_off + 4, la < lb ? la : lb);
  if (c) return c;
  return la < lb ? -1 : la > lb;
}

size_t orc_key_at(orc_engine* e, int64_t i, char* out, size_t cap) {
  if (!e->sorted) {
    e->sorted = (int64_t*)malloc(sizeof(int64_t) * (e->nsets ? e->nsets : 1));
    for (size_t k = 0; k < e->nsets; k++) e->sorted[k] = (int64_t)k;
    g_sort_engine = e;
    qsort(e->sorted, e->nsets, sizeof(int64_t), cmp_keys);
  }
  if (i < 0 || (size_t)i >= e->nsets) return 0;
  setrec* s = &e->sets[e->sorted[i]];
  uint32_t l;
  memcpy(&l, e->keys.bytes + s->key_off, 4);
  memcpy(out, e->keys.bytes + s->key_off + 4, l < cap ? l : cap);
  return l;
}

int orc_key_expiry(orc_engine* e, const char* key, size_t key_len, int64_t* t) {
  setrec* s = get_set(e, key, key_len, 0);
  if (!s || !s->has_expiry) return 0;
  *t = s->expiry;
  return 1;
}

typedef struct {
  const uint8_t* p;
  uint32_t n;
} mref;
static int cmp_mref(const void* a, const void* b) {
  const mref* x = (const mref*)a;
  const mref* y = (const mref*)b;
  int c = memcmp(x->p, y->p, x->n < y->n ? x->n : y->n);
  if (c) return c;
  return x->n < y->n ? -1 : x->n > y->n;
}

size_t orc_set_members(orc_engine* e, const char* key, size_t key_len, uint8_t* out, size_t cap) {
  setrec* s = get_set(e, key, key_len, 0);
  if (!s) return 0;
  size_t need = s->members.used;
  if (need > cap || !out) return need;
  mref* r = (mref*)malloc(sizeof(mref) * (s->members.count ? s->members.count : 1));
  size_t k = 0;
  for (size_t o = 0; o < s->members.used;) {
    uint32_t l;
    memcpy(&l, s->members.bytes + o, 4);
    r[k].p = s->members.bytes + o + 4;
    r[k].n = l;
    k++;
    o += 4 + l;
  }
  qsort(r, k, sizeof(mref), cmp_mref);
  size_t w = 0;
  for (size_t i = 0; i < k; i++) {
    memcpy(out + w, &r[i].n, 4);
    memcpy(out + w + 4, r[i].p, r[i].n);
    w += 4 + r[i].n;
  }
  free(r);
  return need;
}

/* storage-statistics.go:28-82 via GetIssuerAndDatesFromCache (filesystemdatabase.go:59-100):
 * keys "serials::<expDate>::<issuerID>" split on "::"; count = Σ SCARD per issuer. */
int64_t orc_issuer_count(orc_engine* e, const char* issuer_id) {
  size_t il = strlen(issuer_id);
  int64_t total = 0;
  for (size_t i = 0; i < e->nsets; i++) {
    uint32_t l;
    memcpy(&l, e->keys.bytes + e->sets[i].key_off, 4);
    const char* k = (const char*)e->keys.bytes + e->sets[i].key_off + 4;
    if (l < 9 + 2 + il || memcmp(k, "serials::", 9) != 0) continue;
    /* parts[2] is everything after the second "::" provided there are exactly 3 parts */
    const char* second = NULL;
    for (size_t j = 9; j + 1 < l; j++)
      if (k[j] == ':' && k[j + 1] == ':') {
        second = k + j + 2;
        break;
      }
    if (!second) continue;
    size_t rest = l - (size_t)(second - k);
    if (rest == il && memcmp(second, issuer_id, il) == 0) total += (int64_t)e->sets[i].members.count;
  }
  return total;
}

int64_t orc_total_count(orc_engine* e) {
  int64_t total = 0;
  for (size_t i = 0; i < e->nsets; i++) {
    uint32_t l;
    memcpy(&l, e->keys.bytes + e->sets[i].key_off, 4);
    const char* k = (const char*)e->keys.bytes + e->sets[i].key_off + 4;
    if (l >= 9 && memcmp(k, "serials::", 9) == 0) total += (int64_t)e->sets[i].members.count;
  }
  return total;
}

int64_t orc_inserted(orc_engine* e) { return e->inserted; }

/* One pass of insertCTWorker's loop body + FilesystemDatabase.Store.
 * ct-fetch.go:191-235; filesystemdatabase.go:158-211; knowncertificates.go:28-55. */
int orc_engine_entry(orc_engine* e, const uint8_t* leaf, size_t leaf_len, int entry_type, const uint8_t* issuer_der,
                     size_t issuer_len, int* was_unknown, int32_t* exp_hour,
                     const uint8_t** serial, uint32_t* serial_len) {
  orc_cert c;
  if (was_unknown) *was_unknown = 0;
  orc_parse_cert(leaf, leaf_len, &c); /* :198-204 */
  /* X509 entry: the certificate LogEntryFromLeaf parsed, kept unless the error was fatal (:452-459);
   * precertificate: parsed here, dropped on ANY error, x509.NonFatalErrors included (:202-209) */
  if (!c.ok || (e->strict_spki && c.spki_fatal) || (e->strict_ext && c.ext_fatal)) return ORC_ST_PARSE_ERROR;
  if (entry_type == 1 && (c.nonfatal || (e->strict_strings && c.string_findings) || (e->strict_spki && c.spki_findings) ||
                          (e->strict_ext && (c.ext_findings || (e->strict_strings && c.ext_string_findings)))))
    return ORC_ST_PARSE_ERROR; /* :206-209 */
  if (exp_hour) *exp_hour = orc_exp_hour(c.not_after);
  if (serial) *serial = leaf + c.serial_off;
  if (serial_len) *serial_len = c.serial_len;
  int f = orc_cert_is_filtered_out(leaf, &c, e->filter, e->filter_len, e->log_expired, e->now);
  if (f != ORC_ST_PASS) return f; /* :211-213 */
  if (!issuer_der) return ORC_ST_NO_ISSUER; /* :215-219 */
  orc_cert ic;
  orc_parse_cert(issuer_der, issuer_len, &ic); /* :221 */
  if (!ic.ok || ic.nonfatal || (e->strict_strings && ic.string_findings) || (e->strict_spki && (ic.spki_fatal || ic.spki_findings)) ||
      (e->strict_ext && (ic.ext_fatal || ic.ext_findings || (e->strict_strings && ic.ext_string_findings))))
    return ORC_ST_ISSUER_PARSE_ERROR; /* any err :222-225 */
  /* Store: filesystemdatabase.go:158-211 */
  int32_t eh = orc_exp_hour(c.not_after);           /* :160 */
  char issuer_id[45];
  orc_issuer_id(issuer_der + ic.spki_off, ic.spki_len, issuer_id); /* :161, types.go:124-130 */
  char exp_id[16];
  orc_exp_date_id(eh, exp_id);
  char key[96];
  int kl = snprintf(key, sizeof key, "serials::%s::%s", exp_id, issuer_id); /* knowncertificates.go:28-34 */
  setrec* s = get_set(e, key, (size_t)kl, 1);
  int was_new = 0;
  bs_find(&s->members, leaf + c.serial_off, c.serial_len, 1, &was_new); /* SetInsert :39 */
  if (!s->has_expiry) { /* :44-47 → ExpireAt(key, expDate.ExpireTime()) :98-104 */
    s->has_expiry = 1;
    s->expiry = (int64_t)eh * 3600;
  }
  if (was_unknown) *was_unknown = was_new;
  e->inserted++; /* ct-fetch.go:235 */
  return ORC_ST_PASS;
}

void orc_engine_batch(orc_engine* e, const uint8_t* payload, const uint64_t* offsets,
                      const uint32_t* issuer_idx, const uint8_t* entry_type, uint64_t n, const uint8_t* issuer_payload,
                      const uint64_t* issuer_offsets, uint32_t n_issuers, uint8_t* out_status,
                      uint8_t* out_unknown, int32_t* out_exp_hour) {
  for (uint64_t i = 0; i < n; i++) {
    const uint8_t* leaf = payload + offsets[i];
    size_t ll = (size_t)(offsets[i + 1] - offsets[i]);
    const uint8_t* idr = NULL;
    size_t il = 0;
    uint32_t ii = issuer_idx[i];
    if (ii != 0xffffffffu && ii < n_issuers) {
      idr = issuer_payload + issuer_offsets[ii];
      il = (size_t)(issuer_offsets[ii + 1] - issuer_offsets[ii]);
    }
    int unk = 0;
    int32_t eh = 0;
    int st = orc_engine_entry(e, leaf, ll, entry_type ? entry_type[i] : 0, idr, il, &unk, &eh, NULL, NULL);
    if (out_status) out_status[i] = (uint8_t)st;
    if (out_unknown) out_unknown[i] = (uint8_t)unk;
    if (out_exp_hour) out_exp_hour[i] = eh;
  }
}

/* ------------------------------------------------------------------------------------------------
 * ct.LogEntryFromLeaf (cmd/ct-fetch/ct-fetch.go:452) — TLS presentation-language reader.
 * A cursor over one buffer; every read checks what is left (CT-go tls.Unmarshal: "truncated"), and the
 * two top-level structures must consume their buffer exactly ("trailing data"). */
typedef struct {
  const uint8_t* p;
  size_t left;
  int bad;
} tls_cur;

static uint64_t tls_uint(tls_cur* c, int nbytes) {
  if (c->bad || c->left < (size_t)nbytes) {
    c->bad = 1;
    return 0;
  }
  uint64_t v = 0;
  for (int i = 0; i < nbytes; i++) v = (v << 8) | c->p[i];
  c->p += nbytes;
  c->left -= (size_t)nbytes;
  return v;
}

/* opaque<min..max> with an nbytes length prefix: returns a sub-cursor over the contents */
static tls_cur tls_opaque(tls_cur* c, int nbytes, uint64_t minlen) {
  tls_cur sub = {NULL, 0, 1};
  uint64_t n = tls_uint(c, nbytes);
  if (c->bad || n < minlen || n > c->left) {
    c->bad = 1;
    return sub;
  }
  sub.p = c->p;
  sub.left = (size_t)n;
  sub.bad = 0;
  c->p += n;
  c->left -= (size_t)n;
  return sub;
}

void orc_decode_entry(const uint8_t* leaf_input, size_t leaf_len, const uint8_t* extra_data, size_t extra_len,
                      orc_entry* out) {
  memset(out, 0, sizeof *out);
  /* MerkleTreeLeaf (RFC 6962 §3.4): Version version; MerkleLeafType leaf_type; TimestampedEntry */
  tls_cur L = {leaf_input, leaf_len, 0};
  (void)tls_uint(&L, 1);                 /* version: a uint8 enum, v1(0); CT-go only bounds it by 255 */
  uint64_t leaf_type = tls_uint(&L, 1);  /* timestamped_entry(0) selects the only variant */
  if (L.bad || leaf_type != 0) return;
  out->timestamp = tls_uint(&L, 8);
  uint64_t et = tls_uint(&L, 2);
  if (L.bad) return;
  tls_cur cert = {NULL, 0, 1};
  if (et == 0) {                         /* x509_entry: ASN.1Cert = opaque<1..2^24-1> */
    cert = tls_opaque(&L, 3, 1);
    if (L.bad) return;
    out->cert_in_extra = 0;
    out->cert_off = (uint32_t)(cert.p - leaf_input);
    out->cert_len = (uint32_t)cert.left;
  } else if (et == 1) {                  /* precert_entry: opaque issuer_key_hash[32]; TBSCertificate<1..2^24-1> */
    if (L.left < 32) return;
    L.p += 32;
    L.left -= 32;
    tls_cur tbs = tls_opaque(&L, 3, 1);
    if (L.bad) return;
    out->tbs_off = (uint32_t)(tbs.p - leaf_input);
    out->tbs_len = (uint32_t)tbs.left;
  } else {
    return;                              /* RawLogEntryFromLeaf: "unknown entry type" */
  }
  (void)tls_opaque(&L, 2, 0);            /* CtExtensions extensions<0..2^16-1> */
  if (L.bad || L.left != 0) return;      /* "MerkleTreeLeaf: trailing data" */

  /* extra_data (RFC 6962 §4.6) */
  tls_cur X = {extra_data, extra_len, 0};
  if (et == 1) {                         /* PrecertChainEntry.pre_certificate → Precert.Submitted (ct-fetch.go:202) */
    cert = tls_opaque(&X, 3, 1);
    if (X.bad) return;
    out->cert_in_extra = 1;
    out->cert_off = (uint32_t)(cert.p - extra_data);
    out->cert_len = (uint32_t)cert.left;
  }
  tls_cur chain = tls_opaque(&X, 3, 0);  /* ASN.1Cert chain<0..2^24-1> */
  if (X.bad || X.left != 0) return;      /* "CertificateChain / PrecertChainEntry: trailing data" */
  uint32_t n = 0;
  while (chain.left > 0) {
    tls_cur one = tls_opaque(&chain, 3, 1);
    if (chain.bad) {
      out->chain0_off = out->chain0_len = 0;
      return;
    }
    if (n == 0) {
      out->chain0_off = (uint32_t)(one.p - extra_data);
      out->chain0_len = (uint32_t)one.left;
    }
    n++;
  }
  out->n_chain = n;
  out->entry_type = (int32_t)et;
  out->ok = 1;
}

void orc_engine_set_strict_leaf(orc_engine* e, int on) { e->strict_leaf = on != 0; }
void orc_engine_set_strict_spki(orc_engine* e, int on) { e->strict_spki = on != 0; }
void orc_engine_set_strict_extensions(orc_engine* e, int on) { e->strict_ext = on != 0; }
void orc_engine_set_strict_strings(orc_engine* e, int on) { e->strict_strings = on != 0; }

void orc_engine_raw_batch(orc_engine* e, const uint8_t* blob, const uint64_t* bounds, uint64_t n,
                          uint8_t* out_status, uint8_t* out_unknown, int32_t* out_exp_hour,
                          uint64_t* out_timestamp) {
  for (uint64_t i = 0; i < n; i++) {
    const uint8_t* leaf = blob + bounds[2 * i];
    const uint8_t* extra = blob + bounds[2 * i + 1];
    orc_entry d;
    orc_decode_entry(leaf, (size_t)(bounds[2 * i + 1] - bounds[2 * i]), extra,
                     (size_t)(bounds[2 * i + 2] - bounds[2 * i + 1]), &d);
    int st = ORC_ST_ENTRY_DECODE_ERROR, unk = 0;
    int32_t eh = 0;
    if (d.ok && d.entry_type == 1 && e->strict_leaf) {
      /* ct.LogEntryFromLeaf: x509.ParseTBSCertificate(leaf TBSCertificate); a fatal error fails the whole entry, which the
       * downloader then drops (ct-fetch.go:452-459) — non-fatal findings are kept there */
      orc_cert tc;
      orc_parse_tbs(leaf + d.tbs_off, d.tbs_len, &tc);
      if (!tc.ok || (e->strict_spki && tc.spki_fatal) || (e->strict_ext && tc.ext_fatal)) d.ok = 0;
    }
    if (d.ok) {
      /* ct-fetch.go:198-204 the certificate; :215 len(Chain) < 1; :221 Chain[0] */
      const uint8_t* c = (d.cert_in_extra ? extra : leaf) + d.cert_off;
      st = orc_engine_entry(e, c, d.cert_len, d.entry_type, d.n_chain ? extra + d.chain0_off : NULL, d.chain0_len, &unk, &eh,
                            NULL, NULL);
    }
    if (out_status) out_status[i] = (uint8_t)st;
    if (out_unknown) out_unknown[i] = (uint8_t)unk;
    if (out_exp_hour) out_exp_hour[i] = eh;
    if (out_timestamp) out_timestamp[i] = d.ok ? d.timestamp : 0;
  }
}

/* ------------------------------------------------------------------------------------------------
 * IssuerMetadata.Accumulate inputs (storage/issuermetadata.go:92-138): RawIssuer and CRLDistributionPoints. */
/* where the URIs of a cRLDistributionPoints value lie: ext_crldp_site with deep = 0 — Go's positional struct rules, the
 * FullName elements with tag NUMBER 6 (round 5; rounds 1-4 took every [0] { [0] { [6] } } in any order).  0 = malformed. */
static int collect_dp_uris(const uint8_t* d, uint64_t s, uint64_t e, orc_meta* m) {
  int32_t a = 0, b = 0;
  return ext_crldp_site(d, s, e, m, 0, &a, &b) == 0;
}

int orc_cert_meta(const uint8_t* d, size_t L, orc_meta* m) {
  memset(m, 0, sizeof *m);
  orc_cert c;
  orc_parse_cert(d, L, &c);
  if (!c.ok) return 0;
  /* the parse located both inputs and validated every header on the way: the issuer Name, and the extension list */
  m->issuer_off = c.issuer_off;
  m->issuer_len = c.issuer_len;
  uint64_t e = c.exts_off, e_end = c.exts_end;
  while (e < e_end) {
    tlv ext, oid, val;
    rd_tlv(d, e, e_end, &ext);
    uint64_t x = e + ext.hl, x_end = e + ext.hl + ext.len;
    rd_tlv(d, x, x_end, &oid);
    uint64_t oid_c = x + oid.hl;
    x += oid.hl + oid.len;
    rd_tlv(d, x, x_end, &val);
    if (val.tag == 0x01) { x += val.hl + val.len; rd_tlv(d, x, x_end, &val); }
    if (oid.len == 3 && d[oid_c] == 0x55 && d[oid_c + 1] == 0x1d && d[oid_c + 2] == 0x1f) {
      m->n_crl_ext++;
      uint32_t before = m->n_crl;
      if (!collect_dp_uris(d, x + val.hl, x + val.hl + val.len, m)) {
        m->bad_crl = 1;
        m->n_crl = before;
      }
    }
    e += ext.hl + ext.len;
  }
  if (m->bad_crl) m->n_crl = 0;
  return 1;
}
