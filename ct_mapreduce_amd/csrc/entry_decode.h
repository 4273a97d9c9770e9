// entry_decode.h — RFC 6962 get-entries decode.
//
// decode_entry<Bytes>() replaces ct.LogEntryFromLeaf as far as this path consumes it
// (cmd/ct-fetch/ct-fetch.go:452): the TLS structures of `leaf_input` (MerkleTreeLeaf, RFC 6962 §3.4) and
// `extra_data` (§4.6: X509ChainEntry.certificate_chain | PrecertChainEntry) are validated to the byte and
// the three things insertCTWorker needs are located INSIDE the blob — nothing is copied:
//   the certificate it parses   X509 entry: TimestampedEntry.signed_entry (:199-200)
//                               precert entry: PrecertChainEntry.pre_certificate = Precert.Submitted.Data (:201-203)
//   Chain[0]                    first element of certificate_chain / precertificate_chain (:215,:221)
//   the timestamp               TimestampedEntry.timestamp (:476)
// The TLS structure follows certificate-transparency-go v1.1.0's struct tags (types.go; the module is not
// vendored): opaque ASN.1Cert<1..2^24-1>, chains <0..2^24-1>, CtExtensions<0..2^16-1>, trailing
// bytes after either structure are an error, entry types other than x509_entry(0)/precert_entry(1) are an error.
//
// One entry per lane; the same code is compiled for the host (engine: host-side staging checks; tests/harness).
#pragma once
#include <stdint.h>

#include "der_walk.h"  // CTMR_HD

namespace ctmr {

struct EntryDec {
  bool ok;
  uint32_t entry_type;   // 0 | 1 (valid when ok)
  uint64_t timestamp;    // ms
  uint64_t cert_lo, cert_hi;
  uint64_t chain0_lo;
  uint32_t chain0_len;   // 0 = len(Chain) < 1
  uint32_t n_chain;
  uint64_t tbs_lo;       // precert: TBSCertificate inside the leaf
  uint32_t tbs_len;
  uint64_t ikh_lo;       // precert: issuer_key_hash[32]
};

// Bytes: u8(pos) and be(pos, k) = k-byte big-endian integer (k ≤ 4) at an arbitrary byte position.  The
// decoder proves pos + k ≤ end of the entry before every access.
template <class B>
CTMR_HD void decode_entry(const B& b, uint64_t l0, uint64_t l1, uint64_t x1, EntryDec& o) {
  o.ok = false;
  o.entry_type = 0;
  o.timestamp = 0;
  o.cert_lo = o.cert_hi = 0;
  o.chain0_lo = 0;
  o.chain0_len = 0;
  o.n_chain = 0;
  o.tbs_lo = 0;
  o.tbs_len = 0;
  o.ikh_lo = 0;
  if (l1 < l0 || x1 < l1) return;
  // ---- MerkleTreeLeaf: version(1) leaf_type(1) | TimestampedEntry: timestamp(8) entry_type(2) …
  if (l1 - l0 < 12) return;
  // version: any value (CT-go's tls enum check is maxval 255 only); leaf_type must select timestamped_entry(0)
  if (b.u8(l0 + 1) != 0) return;
  o.timestamp = ((uint64_t)b.be(l0 + 2, 4) << 32) | b.be(l0 + 6, 4);
  const uint32_t et = b.be(l0 + 10, 2);
  uint64_t p = l0 + 12;
  if (et == 0) {  // ASN.1Cert signed_entry
    if (l1 - p < 3) return;
    const uint32_t L = b.be(p, 3);
    p += 3;
    if (L < 1 || l1 - p < L) return;
    o.cert_lo = p;
    o.cert_hi = p + L;
    p += L;
  } else if (et == 1) {  // PreCert: issuer_key_hash[32], TBSCertificate<1..2^24-1>
    if (l1 - p < 35) return;
    o.ikh_lo = p;
    p += 32;
    const uint32_t T = b.be(p, 3);
    p += 3;
    if (T < 1 || l1 - p < T) return;
    o.tbs_lo = p;
    o.tbs_len = T;
    p += T;
  } else {
    return;  // "unknown entry type"
  }
  // CtExtensions extensions<0..2^16-1>, then nothing
  if (l1 - p < 2) return;
  const uint32_t E = b.be(p, 2);
  p += 2;
  if (l1 - p != E) return;
  // ---- extra_data
  uint64_t q = l1;
  if (et == 1) {  // PrecertChainEntry.pre_certificate
    if (x1 - q < 3) return;
    const uint32_t P = b.be(q, 3);
    q += 3;
    if (P < 1 || x1 - q < P) return;
    o.cert_lo = q;
    o.cert_hi = q + P;
    q += P;
  }
  // ASN.1Cert chain<0..2^24-1>, then nothing
  if (x1 - q < 3) return;
  const uint32_t C = b.be(q, 3);
  q += 3;
  if (x1 - q != C) return;
  uint32_t n = 0;
  bool good = true;
  while (good & (q < x1)) {
    good = x1 - q >= 3;
    if (good) {
      const uint32_t n1 = b.be(q, 3);
      q += 3;
      good = (n1 >= 1) & (x1 - q >= n1);
      if (good) {
        if (n == 0) {
          o.chain0_lo = q;
          o.chain0_len = n1;
        }
        n++;
        q += n1;
      }
    }
  }
  if (!good) {
    o.chain0_lo = 0;
    o.chain0_len = 0;
    return;
  }
  o.n_chain = n;
  o.entry_type = et;
  o.ok = true;
}

// ---------------------------------------------------------------- Chain[0] → issuer table index
// Candidate selection hash of a certificate: length, first 16 and last 16 bytes (certificates shorter than
// 32 bytes: every byte).  A candidate is always verified bytewise against the registered certificate, so the
// hash only has to spread, never to identify.  Words: w16(pos, k) = k-th little-endian dword at byte pos.
CTMR_HD uint64_t qh_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
template <class B>
CTMR_HD uint64_t cert_quick_hash(const B& b, uint64_t lo, uint32_t len) {
  uint64_t h = qh_mix(0x9e3779b97f4a7c15ull + len);
  if (len >= 32) {
    uint32_t hd[4], tl[4];
    b.le128(lo, hd);             // one 16-byte load each on the device
    b.le128(lo + len - 16, tl);
    for (int k = 0; k < 4; k++) h = qh_mix(h ^ ((uint64_t)hd[k] << 1 | 1));
    for (int k = 0; k < 4; k++) h = qh_mix(h ^ ((uint64_t)tl[k] << 1));
  } else {
    for (uint32_t k = 0; k < len; k++) h = qh_mix(h ^ b.u8(lo + k));
  }
  return h ? h : 1;
}

struct HostBytes {  // host instantiation (registration, staging checks, tests)
  const uint8_t* p;
  uint8_t u8(uint64_t pos) const { return p[pos]; }
  uint32_t be(uint64_t pos, int k) const {
    uint32_t v = 0;
    for (int i = 0; i < k; i++) v = (v << 8) | p[pos + i];
    return v;
  }
  uint32_t le32(uint64_t pos) const {
    return (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
  }
  void le128(uint64_t pos, uint32_t out[4]) const {
    for (int k = 0; k < 4; k++) out[k] = le32(pos + 4 * k);
  }
};

}  // namespace ctmr
