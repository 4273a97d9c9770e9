// kernels/pem.h — PEM write-back.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "exchange.h"

namespace ctmr {

// ------------------------------------------------------------------ PEM write-back
// pem.EncodeToMemory(&pem.Block{Type: "CERTIFICATE", Bytes: aCert.Raw}) of every newly unknown
// certificate (storage/filesystemdatabase.go:167-175,196-200): "-----BEGIN CERTIFICATE-----\n",
// base64.StdEncoding in 64-column lines each ended by "\n", "-----END CERTIFICATE-----\n".
__host__ __device__ inline uint64_t pem_len(uint64_t L) {
  const uint64_t b64 = 4 * ((L + 2) / 3);
  return 28 + b64 + (b64 + 63) / 64 + 26;
}

__global__ void __launch_bounds__(256) k_pem_len(const uint64_t* offsets, const uint64_t* ends, const uint64_t* idx,
                                                 uint64_t n_idx, uint64_t* pem_off) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > n_idx) return;
  if (r == n_idx) {
    pem_off[r] = 0;  // the exclusive scan turns this slot into the total
    return;
  }
  uint64_t lo, hi;
  cert_range(offsets, ends, idx[r], lo, hi);
  pem_off[r] = pem_len(hi - lo);
}

constexpr uint32_t PEM_PER_WAVE = 16;
struct __attribute__((packed, aligned(1))) U12 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t b64_char(uint32_t v) {  // base64.StdEncoding alphabet
  int32_t off = 65;                 // 'A'
  off = v >= 26u ? 71 : off;        // 'a' - 26
  off = v >= 52u ? -4 : off;        // '0' - 52
  off = v == 62u ? -19 : off;       // '+'
  off = v == 63u ? -16 : off;       // '/'
  return (uint32_t)((int32_t)v + off);
}
// three input bytes (little-endian in the low 24 bits of w) → four characters, little-endian
__device__ __forceinline__ uint32_t b64_group(uint32_t w) {
  const uint32_t b0 = w & 0xffu, b1 = (w >> 8) & 0xffu, b2 = (w >> 16) & 0xffu;
  const uint32_t v = (b0 << 16) | (b1 << 8) | b2;
  return b64_char(v >> 18) | (b64_char((v >> 12) & 63u) << 8) | (b64_char((v >> 6) & 63u) << 16) |
         (b64_char(v & 63u) << 24);
}

// One workgroup per certificate; one task = 12 input bytes → 16 characters (a quarter line), so
// adjacent lanes read adjacent 12-byte pieces and write adjacent 16-byte pieces (unaligned
// dwordx3 / dwordx4 accesses; gfx950 runs with unaligned access mode).
__global__ void __launch_bounds__(256) k_pem_encode(const uint8_t* payload, const uint64_t* offsets,
                                                    const uint64_t* ends, const uint64_t* idx, uint64_t n_idx,
                                                    const uint64_t* pem_off, uint8_t* out) {
  // One WAVE per certificate (no workgroup-level cooperation is needed), PEM_PER_WAVE certificates per wave in turn:
  // every wave follows its own idx → offsets → bytes chain, so a CU has 32 certificates in flight instead of 16
  // two-wave workgroups' worth, and the chain of the next certificate is not behind a workgroup's slowest wave.
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
  // the idx → offsets → pem_off chains of all PEM_PER_WAVE certificates of this wave in ONE round: lane c fetches
  // certificate c's, the loop below broadcasts them
  uint64_t m_lo = 0, m_hi = 0, m_po = 0;
  {
    const uint64_t rr = wave * PEM_PER_WAVE + lane;
    if (lane < PEM_PER_WAVE && rr < n_idx) {
      cert_range(offsets, ends, idx[rr], m_lo, m_hi);
      m_po = pem_off[rr];
    }
  }
  for (uint32_t cc = 0; cc < PEM_PER_WAVE; cc++) {
  const uint64_t r = wave * PEM_PER_WAVE + cc;
  if (r >= n_idx) return;
  const uint64_t lo = __shfl(m_lo, (int)cc), hi = __shfl(m_hi, (int)cc);
  const uint64_t L = hi - lo;
  const uint8_t* in = payload + lo;
  uint8_t* o = out + __shfl(m_po, (int)cc);
  const uint64_t b64 = 4 * ((L + 2) / 3), nlines = (b64 + 63) / 64;
  // framing lines as a handful of wide unaligned stores (they were 54 single-byte stores on two threads)
  if (lane == 0) {         // "-----BEGIN CERTIFICATE-----\n" = 16 + 12 bytes
    *(U16*)o = U16{0x2d2d2d2du, 0x4745422du, 0x43204e49u, 0x49545245u};
    *(U12*)(o + 16) = U12{0x41434946u, 0x2d2d4554u, 0x0a2d2d2du};
  } else if (lane == 32) { // "-----END CERTIFICATE-----\n" = 26 bytes: 16 + 12 overlapping by two
    uint8_t* e = o + 28 + b64 + nlines;
    *(U16*)e = U16{0x2d2d2d2du, 0x444e452du, 0x52454320u, 0x49464954u};
    *(U12*)(e + 14) = U12{0x41434946u, 0x2d2d4554u, 0x0a2d2d2du};  // bytes 14..25 (two bytes overlap the store above)
  }
  const uint64_t nq = (L + 11) / 12;
  // (tried, session 5: two tasks per lane per round with both loads issued first — 19.1 ms against 16.2 ms per 16 M
  //  certificates; the kernel is not waiting on these loads)
  for (uint64_t k = lane; k < nq; k += 64) {
    const uint64_t ip = 12 * k;
    const uint32_t nin = (uint32_t)(L - ip < 12 ? L - ip : 12);
    const U12 v = *(const U12*)(in + ip);  // may read ≤ 11 bytes past the certificate: CTMR_PAYLOAD_PAD
    uint32_t g[4] = {v.a & 0xffffffu, (v.a >> 24) | ((v.b & 0xffffu) << 8), (v.b >> 16) | ((v.c & 0xffu) << 16),
                     v.c >> 8};
    uint8_t* q = o + 28 + (k >> 2) * 65 + (k & 3) * 16;
    if (nin == 12) {
      U16 w{b64_group(g[0]), b64_group(g[1]), b64_group(g[2]), b64_group(g[3])};
      *(U16*)q = w;
      if ((k & 3) == 3 || k == nq - 1) q[16] = (uint8_t)'\n';
    } else {  // last, partial task: whole groups, then one padded group, then the line end
      uint32_t done = 0, c = 0;
      for (; done + 3 <= nin; done += 3, c += 4) {
        const uint32_t w = b64_group(g[done / 3]);
        q[c] = (uint8_t)w; q[c + 1] = (uint8_t)(w >> 8); q[c + 2] = (uint8_t)(w >> 16); q[c + 3] = (uint8_t)(w >> 24);
      }
      const uint32_t rem = nin - done;
      if (rem) {
        const uint32_t x = g[done / 3] & (rem == 1 ? 0xffu : 0xffffu);
        const uint32_t w = b64_group(x);
        q[c] = (uint8_t)w; q[c + 1] = (uint8_t)(w >> 8);
        q[c + 2] = rem == 2 ? (uint8_t)(w >> 16) : (uint8_t)'=';
        q[c + 3] = (uint8_t)'=';
        c += 4;
      }
      q[c] = (uint8_t)'\n';
    }
  }
  }
}

}  // namespace ctmr
