// synth.h — deterministic synthetic CT batch generator.
//
// Benchmark / test INPUT generator, not part of the reference's surface: it emits the packed
// layout the map kernel consumes.  Every certificate is a pure function of (config, index), so
// the host (ctmr_synth_leaf) and the device generator kernel (ctmr_synth_device) produce
// byte-identical output from the same code (CTMR_HD).  No crypto: signatures and moduli are
// random bytes — the reference never verifies signatures.
#pragma once
#include <stdint.h>

#include "der_walk.h"  // CTMR_HD
#include "synth_p256.h"

namespace ctmr {

struct SynthCfg {
  uint64_t seed;
  uint32_t n_issuers, zipf, dup_permille, ca_permille, expired_permille, mean_len;
  int64_t base_time;
  const uint32_t* zipf_cdf;  // n_issuers thresholds in [0,2^32): issuer = first k with u < cdf[k]
  uint32_t profile;          // 0 = SURVEY §8(d) corpus (RSA-2048, short names, UTCTime); 1 = mixed (see synth_leaf_emit)
};

CTMR_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
CTMR_HD uint64_t h3(uint64_t seed, uint64_t i, uint64_t salt) {
  return mix64(mix64(seed ^ (salt * 0xd6e8feb86659fd93ull)) + i);
}

struct Rng {  // splitmix64 stream
  uint64_t s;
  CTMR_HD uint64_t next() {
    s += 0x9e3779b97f4a7c15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
};

enum : uint64_t { SALT_DUP = 1, SALT_SRC = 2, SALT_IDENT = 3, SALT_KEY = 4, SALT_BODY = 5,
                  SALT_TYPE = 6, SALT_ISSUER = 7 };

CTMR_HD bool synth_is_dup(const SynthCfg& c, uint64_t i) {
  return i > 0 && c.dup_permille > 0 && (h3(c.seed, i, SALT_DUP) % 1000u) < c.dup_permille;
}
// The entry whose (issuer, serial, notAfter) entry i carries.
CTMR_HD uint64_t synth_src(const SynthCfg& c, uint64_t i) {
  if (!synth_is_dup(c, i)) return i;
  uint64_t j = h3(c.seed, i, SALT_SRC) % i;
  while (synth_is_dup(c, j)) j--;
  return j;
}

// Backward DER writer: content first, then its header in front of it.
struct BackWriter {
  uint8_t* buf;   // nullptr = count only
  uint32_t pos;   // next byte goes to buf[pos-1]
  CTMR_HD void put(uint8_t b) {
    --pos;
    if (buf) buf[pos] = b;
  }
  CTMR_HD void bytes(const uint8_t* s, uint32_t n) {
    for (uint32_t k = n; k > 0; k--) put(s[k - 1]);
  }
  CTMR_HD void str(const char* s, uint32_t n) { bytes((const uint8_t*)s, n); }
  CTMR_HD void random(Rng& r, uint32_t n) {  // n random bytes (consumes ceil(n/8) draws)
    // emitted so that the forward byte order is draw order: generate forward, store backward
    const uint32_t start = pos - n;
    uint64_t v = 0;
    for (uint32_t k = 0; k < n; k++) {
      if ((k & 7u) == 0) v = r.next();
      if (buf) buf[start + k] = (uint8_t)(v >> (8 * (k & 7u)));
    }
    pos = start;
  }
  // header for content that starts at the current pos and ends at `end`
  CTMR_HD void hdr(uint8_t tag, uint32_t end) {
    const uint32_t len = end - pos;
    if (len < 0x80u) {
      put((uint8_t)len);
    } else if (len < 0x100u) {
      put((uint8_t)len);
      put(0x81);
    } else if (len < 0x10000u) {
      put((uint8_t)len);
      put((uint8_t)(len >> 8));
      put(0x82);
    } else {
      put((uint8_t)len);
      put((uint8_t)(len >> 8));
      put((uint8_t)(len >> 16));
      put(0x83);
    }
    put(tag);
  }
};

CTMR_HD void civil_from_days(int64_t z, int32_t& y, uint32_t& m, uint32_t& d) {
  z += 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const uint32_t doe = (uint32_t)(z - era * 146097);
  const uint32_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const uint32_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const uint32_t mp = (5 * doy + 2) / 153;
  d = doy - (153 * mp + 2) / 5 + 1;
  m = mp < 10 ? mp + 3 : mp - 9;
  y = (int32_t)(yoe + era * 400) + (m <= 2);
}

// UTCTime "YYMMDDHHMMSSZ" TLV, written backward
CTMR_HD void put_utctime(BackWriter& w, int64_t t) {
  int64_t days = t / 86400;
  int64_t rem = t % 86400;
  if (rem < 0) {
    rem += 86400;
    days -= 1;
  }
  int32_t y;
  uint32_t mo, d;
  civil_from_days(days, y, mo, d);
  const uint32_t hh = (uint32_t)rem / 3600u, mi = ((uint32_t)rem % 3600u) / 60u, ss = (uint32_t)rem % 60u;
  const uint32_t yy = (uint32_t)(y % 100);
  const uint32_t end = w.pos;
  w.put('Z');
  w.put('0' + ss % 10); w.put('0' + ss / 10);
  w.put('0' + mi % 10); w.put('0' + mi / 10);
  w.put('0' + hh % 10); w.put('0' + hh / 10);
  w.put('0' + d % 10);  w.put('0' + d / 10);
  w.put('0' + mo % 10); w.put('0' + mo / 10);
  w.put('0' + yy % 10); w.put('0' + yy / 10);
  w.hdr(0x17, end);
}

CTMR_HD void put_dec3(BackWriter& w, uint32_t v) {  // at least 3 decimal digits
  uint32_t n = 0;
  do {
    w.put('0' + v % 10);
    v /= 10;
    n++;
  } while (v || n < 3);
}

CTMR_HD void put_hex(BackWriter& w, uint64_t v, uint32_t digits) {
  for (uint32_t k = 0; k < digits; k++) {
    const uint32_t x = (uint32_t)(v & 0xf);
    w.put((uint8_t)(x < 10 ? '0' + x : 'a' + x - 10));
    v >>= 4;
  }
}

// AttributeTypeAndValue RDN: SET { SEQ { OID 2.5.4.x, <tag> value } } with value already written
CTMR_HD void close_rdn(BackWriter& w, uint8_t attr, uint8_t strtag, uint32_t end) {
  w.hdr(strtag, end);
  w.put(attr); w.put(0x04); w.put(0x55); w.put(0x03); w.put(0x06);
  w.hdr(0x30, end);
  w.hdr(0x31, end);
}

// issuer DN: C=US, O=Synth CA Org, CN=Synth Issuer NNN  (written backward: CN first)
CTMR_HD void put_issuer_name(BackWriter& w, uint32_t k, uint32_t profile = 0) {
  const uint32_t end = w.pos;
  {
    const uint32_t e = w.pos;
    put_dec3(w, k);
    w.str("Synth Issuer ", 13);
    close_rdn(w, 0x03, 0x0c, e);
  }
  if (profile == 1 && k % 3u != 0) {  // mixed corpus: OU, and a long O for one issuer in three
    const uint32_t e = w.pos;
    w.str("Synth Trust Services Division", 29);
    close_rdn(w, 0x0b, 0x13, e);
  }
  {
    const uint32_t e = w.pos;
    if (profile == 1 && k % 3u == 2) w.str(" Certification Authority Holdings Incorporated", 46);
    w.str("Synth CA Org", 12);
    close_rdn(w, 0x0a, 0x0c, e);
  }
  {
    const uint32_t e = w.pos;
    w.str("US", 2);
    close_rdn(w, 0x06, 0x13, e);
  }
  w.hdr(0x30, end);
}

CTMR_HD void put_rsa_spki(BackWriter& w, Rng& r) {
  const uint32_t end = w.pos;
  const uint8_t exp_[5] = {0x02, 0x03, 0x01, 0x00, 0x01};
  w.bytes(exp_, 5);
  {
    const uint32_t e = w.pos;
    w.random(r, 256);
    if (w.buf) {
      w.buf[w.pos] |= 0x80;        // 2048-bit modulus
      w.buf[w.pos + 255] |= 0x01;  // odd
    }
    w.put(0x00);
    w.hdr(0x02, e);
  }
  w.hdr(0x30, end);   // RSAPublicKey
  w.put(0x00);        // unused bits
  w.hdr(0x03, end);   // BIT STRING
  const uint8_t alg[15] = {0x30, 0x0d, 0x06, 0x09, 0x2a, 0x86, 0x48, 0x86,
                                  0xf7, 0x0d, 0x01, 0x01, 0x01, 0x05, 0x00};
  w.bytes(alg, 15);
  w.hdr(0x30, end);
}

// id-ecPublicKey, prime256v1, uncompressed point: 91 bytes.  The point is a REAL curve point (one of the 64 of
// synth_p256.h, picked by the stream): CT-go's parsePublicKey (elliptic.Unmarshal) — like OpenSSL — rejects a
// certificate whose point is off the curve, so 64 random bytes (rounds 1–3) made a corpus the reference drops.
CTMR_HD void put_ec_spki(BackWriter& w, Rng& r) {
  static constexpr uint8_t kPoints[64][64] = CTMR_P256_POINTS;
  const uint32_t end = w.pos;
  w.bytes(kPoints[r.next() & 63u], 64);
  w.put(0x04);
  w.put(0x00);
  w.hdr(0x03, end);
  const uint8_t alg[21] = {0x30, 0x13, 0x06, 0x07, 0x2a, 0x86, 0x48, 0xce, 0x3d, 0x02, 0x01,
                           0x06, 0x08, 0x2a, 0x86, 0x48, 0xce, 0x3d, 0x03, 0x01, 0x07};
  w.bytes(alg, 21);
  w.hdr(0x30, end);
}

// GeneralizedTime "YYYYMMDDHHMMSSZ" TLV, written backward
CTMR_HD void put_gentime(BackWriter& w, int64_t t) {
  int64_t days = t / 86400;
  int64_t rem = t % 86400;
  if (rem < 0) {
    rem += 86400;
    days -= 1;
  }
  int32_t y;
  uint32_t mo, d;
  civil_from_days(days, y, mo, d);
  const uint32_t hh = (uint32_t)rem / 3600u, mi = ((uint32_t)rem % 3600u) / 60u, ss = (uint32_t)rem % 60u;
  const uint32_t end = w.pos;
  w.put('Z');
  w.put('0' + ss % 10); w.put('0' + ss / 10);
  w.put('0' + mi % 10); w.put('0' + mi / 10);
  w.put('0' + hh % 10); w.put('0' + hh / 10);
  w.put('0' + d % 10);  w.put('0' + d / 10);
  w.put('0' + mo % 10); w.put('0' + mo / 10);
  uint32_t yy = (uint32_t)y;
  for (int k = 0; k < 4; k++) {
    w.put('0' + yy % 10);
    yy /= 10;
  }
  w.hdr(0x18, end);
}

CTMR_HD void put_sig(BackWriter& w, Rng& r) {  // sha256WithRSAEncryption + 256-byte signature
  const uint32_t end = w.pos;
  w.random(r, 256);
  w.put(0x00);
  w.hdr(0x03, end);
  const uint8_t alg[15] = {0x30, 0x0d, 0x06, 0x09, 0x2a, 0x86, 0x48, 0x86,
                                  0xf7, 0x0d, 0x01, 0x01, 0x0b, 0x05, 0x00};
  w.bytes(alg, 15);
}

CTMR_HD void put_sigalg(BackWriter& w) {
  const uint8_t alg[15] = {0x30, 0x0d, 0x06, 0x09, 0x2a, 0x86, 0x48, 0x86,
                                  0xf7, 0x0d, 0x01, 0x01, 0x0b, 0x05, 0x00};
  w.bytes(alg, 15);
}

// Extension ::= SEQ { OID 2.5.29.x, [critical TRUE,] OCTET STRING value } with value written
CTMR_HD void close_ext(BackWriter& w, uint8_t id, bool critical, uint32_t end) {
  w.hdr(0x04, end);
  if (critical) {
    w.put(0xff); w.put(0x01); w.put(0x01);
  }
  w.put(id); w.put(0x1d); w.put(0x55); w.put(0x03); w.put(0x06);
  w.hdr(0x30, end);
}

CTMR_HD void issuer_key_id(const SynthCfg& c, uint32_t k, uint8_t out[20]) {
  Rng r{h3(c.seed, k, SALT_ISSUER) ^ 0x5157};
  for (int j = 0; j < 20; j += 8) {
    const uint64_t v = r.next();
    for (int b = 0; b < 8 && j + b < 20; b++) out[j + b] = (uint8_t)(v >> (8 * b));
  }
}

CTMR_HD uint32_t synth_pick_issuer(const SynthCfg& c, uint32_t u) {
  if (c.n_issuers <= 1) return 0;
  if (!c.zipf || !c.zipf_cdf) return (uint32_t)(((uint64_t)u * c.n_issuers) >> 32);
  uint32_t lo = 0, hi = c.n_issuers - 1;  // first k with u < cdf[k]; cdf[n-1] treated as 2^32
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (u < c.zipf_cdf[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Emit leaf certificate i backward into w; returns nothing — length = cap - w.pos afterwards.
CTMR_HD void synth_leaf_emit(const SynthCfg& c, uint64_t i, BackWriter& w, uint32_t& issuer_idx,
                             uint8_t& entry_type, uint32_t* tbs_total = nullptr) {
  const uint64_t src = synth_src(c, i);
  const bool identical = src != i && (h3(c.seed, i, SALT_IDENT) & 1u);
  Rng kr{h3(c.seed, src, SALT_KEY)};
  Rng br{h3(c.seed, identical ? src : i, SALT_BODY)};
  entry_type = (uint8_t)(h3(c.seed, i, SALT_TYPE) & 1u);

  // ---- key material (shared by duplicates)
  const uint64_t k0 = kr.next();
  const uint32_t issuer = synth_pick_issuer(c, (uint32_t)(k0 >> 32));
  issuer_idx = issuer;
  const bool is_ca = ((uint32_t)(k0 & 0xffffu) % 1000u) < c.ca_permille;
  const bool expired = ((uint32_t)((k0 >> 16) & 0xffffu) % 1000u) < c.expired_permille;
  uint8_t serial[16];
  {
    const uint64_t a = kr.next(), b = kr.next();
    for (int j = 0; j < 8; j++) {
      serial[j] = (uint8_t)(a >> (8 * j));
      serial[8 + j] = (uint8_t)(b >> (8 * j));
    }
    if (serial[0] == 0) serial[0] = 1;  // keep the INTEGER minimal
  }
  const int64_t base = c.base_time;
  const uint64_t tr = kr.next();
  const int64_t not_after = expired ? base - 1 - (int64_t)(tr % (30ull * 86400ull))
                                    : base + 3600 + (int64_t)(tr % (90ull * 86400ull - 3600ull));
  const int64_t not_before = base - 90 * 86400;

  // ---- body randomness
  const uint64_t lr = br.next();
  uint32_t s16 = 0;  // Irwin–Hall: 8 × U16 → approx normal, all integer
  {
    uint64_t a = lr, b = br.next();
    for (int j = 0; j < 4; j++) {
      s16 += (uint32_t)(a & 0xffffu) + (uint32_t)(b & 0xffffu);
      a >>= 16;
      b >>= 16;
    }
  }
  const int32_t mean = (int32_t)(c.mean_len ? c.mean_len : 1536u);
  int32_t target = mean + ((int32_t)s16 - 8 * 32768 + 4) * 64 / 53510;
  if (target < mean - 336) target = mean - 336;  // [1200, 2000] at mean 1536
  if (target > mean + 464) target = mean + 464;
  const uint64_t subj = br.next();

  const uint32_t cap = w.pos;
  // signatureValue + signatureAlgorithm
  put_sig(w, br);
  const uint32_t tbs_end = w.pos;
  // ---- extensions (backward: SAN last in the certificate, so written first)
  const uint32_t ext_end = w.pos;
  {
    // subjectAltName: dNSNames until the certificate reaches the target length.
    // bytes still to come in front of the SAN ≈ fixed_front; SAN header overhead 4+9+4.
    const uint32_t fixed_front = 4 + 4 + 5 + 19 + 15 + 67 + 32 + 48 + 294 + 8 + 16 + 31 + 17 + 31 + 33 + 62 + 17;
    const uint32_t tail = cap - w.pos;
    int32_t budget = target - (int32_t)fixed_front - (int32_t)tail;
    if (budget < 24) budget = 24;
    const uint32_t e = w.pos;
    uint32_t nname = 0;
    while (budget > 0) {
      uint32_t nl = 12 + (uint32_t)(br.next() % 28u);  // label length 12..39 before ".example"
      if ((int32_t)(nl + 10) > budget) nl = budget > 22 ? (uint32_t)budget - 10 : 12;
      const uint32_t ne = w.pos;
      w.str(".example", 8);
      uint64_t v = br.next();
      for (uint32_t k = 0; k < nl; k++) {
        if ((k & 15u) == 15u) v = br.next();
        w.put((uint8_t)('a' + (v & 0xf)));
        v >>= 4;
      }
      w.hdr(0x82, ne);
      budget -= (int32_t)(nl + 8 + 2);
      nname++;
    }
    (void)nname;
    w.hdr(0x30, e);
    close_ext(w, 0x11, false, e);
  }
  {  // cRLDistributionPoints: one http URI
    const uint32_t e = w.pos;
    w.str(".example/ca.crl", 15);
    put_dec3(w, issuer);
    w.str("http://crl.synth-", 17);
    w.hdr(0x86, e);
    w.hdr(0xa0, e);
    w.hdr(0xa0, e);
    w.hdr(0x30, e);
    w.hdr(0x30, e);
    close_ext(w, 0x1f, false, e);
  }
  {  // authorityKeyIdentifier = issuer's subjectKeyIdentifier
    const uint32_t e = w.pos;
    uint8_t kid[20];
    issuer_key_id(c, issuer, kid);
    w.bytes(kid, 20);
    w.hdr(0x80, e);
    w.hdr(0x30, e);
    close_ext(w, 0x23, false, e);
  }
  {  // subjectKeyIdentifier
    const uint32_t e = w.pos;
    w.random(br, 20);
    w.hdr(0x04, e);
    close_ext(w, 0x0e, false, e);
  }
  {  // basicConstraints (critical): CA:FALSE = empty SEQUENCE, CA:TRUE = { TRUE }
    const uint32_t e = w.pos;
    if (is_ca) {
      w.put(0xff); w.put(0x01); w.put(0x01);
    }
    w.hdr(0x30, e);
    close_ext(w, 0x13, true, e);
  }
  {  // extKeyUsage: serverAuth, clientAuth
    const uint32_t e = w.pos;
    const uint8_t eku[20] = {0x06, 0x08, 0x2b, 0x06, 0x01, 0x05, 0x05, 0x07, 0x03, 0x01,
                                    0x06, 0x08, 0x2b, 0x06, 0x01, 0x05, 0x05, 0x07, 0x03, 0x02};
    w.bytes(eku, 20);
    w.hdr(0x30, e);
    close_ext(w, 0x25, false, e);
  }
  {  // keyUsage (critical): digitalSignature, keyEncipherment
    const uint32_t e = w.pos;
    w.put(0xa0); w.put(0x05);
    w.hdr(0x03, e);
    close_ext(w, 0x0f, true, e);
  }
  w.hdr(0x30, ext_end);
  w.hdr(0xa3, ext_end);
  // ---- subjectPublicKeyInfo (mixed corpus: half the keys are EC P-256, 91 instead of 294 bytes)
  const bool mixed = c.profile == 1;
  if (mixed && (subj & 1u)) put_ec_spki(w, br); else put_rsa_spki(w, br);
  // ---- subject: CN=host-%016x.example; mixed corpus: 40 % of the subjects are OV-like
  //      (C, ST, L, O of 8…71 characters, OU) — 120…260 bytes instead of 38
  {
    const uint32_t end = w.pos;
    {
      const uint32_t e = w.pos;
      w.str(".example", 8);
      put_hex(w, subj, 16);
      w.str("host-", 5);
      close_rdn(w, 0x03, 0x0c, e);
    }
    if (mixed && ((subj >> 8) % 5u) < 2u) {
      { const uint32_t e = w.pos; w.str("Platform Engineering", 20); close_rdn(w, 0x0b, 0x0c, e); }
      {
        const uint32_t e = w.pos;
        const uint32_t ol = 8u + (uint32_t)((subj >> 16) & 63u);
        uint64_t v = subj;
        for (uint32_t k = 0; k < ol; k++) {
          w.put((uint8_t)('a' + (v & 0xf)));
          v = (v >> 4) | (v << 60);
        }
        close_rdn(w, 0x0a, 0x0c, e);
      }
      { const uint32_t e = w.pos; w.str("San Francisco", 13); close_rdn(w, 0x07, 0x0c, e); }
      { const uint32_t e = w.pos; w.str("California", 10); close_rdn(w, 0x08, 0x0c, e); }
      { const uint32_t e = w.pos; w.str("US", 2); close_rdn(w, 0x06, 0x13, e); }
    }
    w.hdr(0x30, end);
  }
  // ---- validity (mixed corpus: one notAfter in four is a GeneralizedTime)
  {
    const uint32_t end = w.pos;
    if (mixed && ((subj >> 12) & 3u) == 0u) put_gentime(w, not_after); else put_utctime(w, not_after);
    put_utctime(w, not_before);
    w.hdr(0x30, end);
  }
  put_issuer_name(w, issuer, c.profile);
  put_sigalg(w);
  {  // serialNumber: positive, DER-minimal (leading 00 when the top bit is set)
    const uint32_t e = w.pos;
    w.bytes(serial, 16);
    if (serial[0] & 0x80) w.put(0x00);
    w.hdr(0x02, e);
  }
  w.put(0x02); w.put(0x01); w.put(0x02); w.put(0x03); w.put(0xa0);  // [0] { INTEGER 2 }
  w.hdr(0x30, tbs_end);
  if (tbs_total) *tbs_total = tbs_end - w.pos;  // the TBSCertificate TLV
  w.hdr(0x30, cap);
}

// Issuer (CA) certificate k: subject = issuer = the DN the leaves carry; distinct RSA SPKI.
CTMR_HD void synth_issuer_emit(const SynthCfg& c, uint32_t k, BackWriter& w) {
  Rng r{h3(c.seed, k, SALT_ISSUER)};
  const uint32_t cap = w.pos;
  put_sig(w, r);
  const uint32_t tbs_end = w.pos;
  const uint32_t ext_end = w.pos;
  {  // subjectKeyIdentifier
    const uint32_t e = w.pos;
    uint8_t kid[20];
    issuer_key_id(c, k, kid);
    w.bytes(kid, 20);
    w.hdr(0x04, e);
    close_ext(w, 0x0e, false, e);
  }
  {  // keyUsage (critical): keyCertSign, cRLSign
    const uint32_t e = w.pos;
    w.put(0x06); w.put(0x01);
    w.hdr(0x03, e);
    close_ext(w, 0x0f, true, e);
  }
  {  // basicConstraints (critical) CA:TRUE
    const uint32_t e = w.pos;
    w.put(0xff); w.put(0x01); w.put(0x01);
    w.hdr(0x30, e);
    close_ext(w, 0x13, true, e);
  }
  w.hdr(0x30, ext_end);
  w.hdr(0xa3, ext_end);
  put_rsa_spki(w, r);
  put_issuer_name(w, k, c.profile);  // subject
  {
    const uint32_t end = w.pos;
    put_utctime(w, c.base_time + 3650ll * 86400);
    put_utctime(w, c.base_time - 365ll * 86400);
    w.hdr(0x30, end);
  }
  put_issuer_name(w, k, c.profile);  // issuer (self-issued)
  put_sigalg(w);
  {
    const uint32_t e = w.pos;
    w.random(r, 8);
    if (w.buf) w.buf[w.pos] = (w.buf[w.pos] & 0x7f) | 0x01;
    w.hdr(0x02, e);
  }
  w.put(0x02); w.put(0x01); w.put(0x02); w.put(0x03); w.put(0xa0);
  w.hdr(0x30, tbs_end);
  w.hdr(0x30, cap);
}

constexpr uint32_t SYNTH_MAX_LEN = 2560;  // scratch size for one certificate

// ---------------------------------------------------------------- raw get-entries form (RFC 6962 §3.4, §4.6)
// entry i = leaf_input ‖ extra_data around the certificate of synth_leaf_emit(i):
//   entry_type 0: MerkleTreeLeaf{v1, timestamped_entry, ts, x509_entry, ASN.1Cert cert, ext<>} ‖ chain<[issuer]>
//   entry_type 1: MerkleTreeLeaf{v1, timestamped_entry, ts, precert_entry, issuer_key_hash, TBS, ext<>} ‖
//                 PrecertChainEntry{pre_certificate = cert, chain<[issuer]>}
// (the synthetic precertificate carries no poison extension and issuer_key_hash is pseudo-random: neither is
// consumed by the path).  Written backward like everything else; returns the length of leaf_input.
constexpr uint32_t SYNTH_ENTRY_MAX = 6144;  // scratch size for one raw entry

CTMR_HD void put_be24(BackWriter& w, uint32_t v) {
  w.put((uint8_t)v); w.put((uint8_t)(v >> 8)); w.put((uint8_t)(v >> 16));
}

CTMR_HD uint32_t synth_entry_issuer(const SynthCfg& c, uint64_t i) {  // the issuer synth_leaf_emit(i) picks
  Rng kr{h3(c.seed, synth_src(c, i), SALT_KEY)};
  return synth_pick_issuer(c, (uint32_t)(kr.next() >> 32));
}

CTMR_HD uint32_t synth_entry_emit(const SynthCfg& c, uint64_t i, BackWriter& w) {
  const uint32_t cap = w.pos;
  const uint8_t et = (uint8_t)(h3(c.seed, i, SALT_TYPE) & 1u);
  const uint32_t issuer = synth_entry_issuer(c, i);
  uint32_t iss2, tbs_total = 0;
  uint8_t et2;
  // ---- extra_data (behind the leaf in memory, so written first)
  {
    const uint32_t e = w.pos;
    synth_issuer_emit(c, issuer, w);
    put_be24(w, e - w.pos);  // ASN.1Cert
    put_be24(w, e - w.pos);  // chain<0..2^24-1>
  }
  uint32_t cert_pos = 0;
  if (et == 1) {
    const uint32_t e = w.pos;
    synth_leaf_emit(c, i, w, iss2, et2, &tbs_total);
    cert_pos = w.pos;
    put_be24(w, e - w.pos);  // pre_certificate
  }
  const uint32_t extra_len = cap - w.pos;
  // ---- leaf_input
  w.put(0); w.put(0);  // CtExtensions<>
  if (et == 0) {
    const uint32_t e = w.pos;
    synth_leaf_emit(c, i, w, iss2, et2);
    put_be24(w, e - w.pos);
  } else {
    const uint32_t e = w.pos;
    // TBSCertificate = the TBS TLV of the certificate just written (it starts behind the 4-byte outer header)
    if (w.buf) {
      for (uint32_t k = tbs_total; k > 0; k--) w.put(w.buf[cert_pos + 4 + k - 1]);
    } else {
      w.pos -= tbs_total;
    }
    put_be24(w, e - w.pos);
    Rng hr{h3(c.seed, issuer, SALT_ISSUER) ^ 0x1b5};
    w.random(hr, 32);  // issuer_key_hash
  }
  w.put(et); w.put(0);  // entry_type (u16)
  {
    uint64_t ts = (uint64_t)c.base_time * 1000ull + i;
    for (int k = 0; k < 8; k++) {
      w.put((uint8_t)ts);
      ts >>= 8;
    }
  }
  w.put(0);  // leaf_type timestamped_entry
  w.put(0);  // version v1
  return cap - w.pos - extra_len;
}

}  // namespace ctmr
