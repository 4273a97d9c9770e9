// kernels/sha256.h — SHA-256 (issuer ids): one message per lane, round constants in LDS.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "readers.h"

namespace ctmr {

// ------------------------------------------------------------------ SHA-256 (issuer ids)
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) {
  return __builtin_amdgcn_alignbit(x, x, n);
}

__constant__ uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// One SHA-256 compression: h += F(h, w); w[] is the 16-word block, used as the rolling schedule.
__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16], const uint32_t* kc) {
  uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = hh + S1 + ch + kc[i] + w[i & 15];
    const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    const uint32_t mj = (a & bb) ^ (a & c) ^ (bb & c);
    const uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
  }
  h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// One lane hashes one message; round constants come from LDS (kc), message bytes through the
// reader.  w[] is a 16-word rolling schedule.
template <class R>
__device__ void sha256_lane(const R& r, uint32_t off, uint32_t len, const uint32_t* kc,
                            uint32_t out[8]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const uint32_t nblk = (len + 9 + 63) / 64;
  for (uint32_t b = 0; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t pos = b * 64 + i * 4;
      uint32_t v = 0;
      if (pos + 4 <= len) {
        v = __builtin_bswap32(r.ld4(off + pos));
      } else if (pos <= len) {
        // tail: message bytes, then 0x80, then zeros
        const uint32_t rem = len - pos;  // 0..3 message bytes in this word
        const uint32_t raw = rem ? r.ld4(off + pos) : 0u;
        const uint32_t m = rem ? (raw & (0xffffffffu >> (8 * (4 - rem)))) : 0u;
        v = __builtin_bswap32(m | (0x80u << (8 * rem)));
      }
      w[i] = v;
    }
    if (b == nblk - 1) {
      w[14] = (uint32_t)(((unsigned long long)len * 8ull) >> 32);
      w[15] = (uint32_t)((unsigned long long)len * 8ull);
    }
    sha256_compress(h, w, kc);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = h[i];
}

// Issuer table: one issuer certificate per lane.  Replaces x509.ParseCertificate(Chain[0])
// (ct-fetch.go:221) + NewIssuer + Issuer.ID()'s SHA-256 (storage/types.go:109-130,155-159).
__global__ void CTMR_WALK_BOUNDS k_issuer_ids(const uint8_t* der, const uint64_t* offsets,
                                                   uint32_t n, uint8_t* valid, uint32_t* digest, uint32_t strict_strings,
                                                   uint32_t strict_spki, uint32_t strict_ext) {
  __shared__ uint32_t kc[64];
  kc[threadIdx.x] = K256[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  GlobalReader r{(const uint32_t*)der, offsets[i]};
  const uint32_t L = (uint32_t)(offsets[i + 1] - offsets[i]);
  Walk w;
  // any err of x509.ParseCertificate(Chain[0]) skips the entry, non-fatal findings included (ct-fetch.go:221-225)
  bool ok = (offsets[i + 1] - offsets[i]) <= 0x7fffffffull && walk_cert(r, L, w, nullptr, strict_spki != 0u, strict_strings != 0u, strict_ext != 0u) &&
            w.nonfatal == 0u;  // strict_strings: a character-set finding in either Name is one more of them (WALK_NF_STRING)
  valid[i] = ok ? 1 : 0;
  uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) sha256_lane(r, w.spki_off, w.spki_len, kc, dg);
#pragma unroll
  for (int k = 0; k < 8; k++) digest[i * 8 + k] = dg[k];
}

// SHA-256 of one host-supplied message (SPKI.Sha256DigestURLEncodedBase64 for an Issuer built from raw SPKI
// bytes, storage/types.go:155-159): one lane, round constants in LDS.
__global__ void __launch_bounds__(64) k_sha256_one(const uint8_t* msg, uint32_t len, uint32_t* digest) {
  __shared__ uint32_t kc[64];
  kc[threadIdx.x] = K256[threadIdx.x];
  __syncthreads();
  if (threadIdx.x != 0) return;
  GlobalReader r{(const uint32_t*)msg, 0};
  uint32_t dg[8];
  sha256_lane(r, 0, len, kc, dg);
#pragma unroll
  for (int k = 0; k < 8; k++) digest[k] = dg[k];
}

}  // namespace ctmr
