// kernels.h — hand-written HIP kernels for gfx950 (CDNA4, wave64).  No MFMA anywhere: this
// path is byte/integer scan + hash work bounded by HBM bandwidth.
//
//   k_issuer_ids     issuer table: walk Chain[0], SHA-256(RawSubjectPublicKeyInfo)
//   k_map_tile       THE dominant kernel: packed DER → LDS tile (coalesced 16 B/lane) →
//                    one-cert-per-lane TBS walk → 3 filters → 32-B record
//   k_map_direct     same map, reading DER straight from global memory (no LDS staging);
//                    also the fallback for tiles larger than the LDS budget
//   k_insert         known-certificate table insert (CAS claim + atomicMin of batch index)
//   k_resolve        WasUnknown decision, wave-aggregated per-issuer / per-(expDate,issuer)
//                    counters, status histogram, per-block NEW counts
//   k_compact        ballot/popcount stream compaction of NEW entries (ascending log index)
//   k_set_op / k_sweep / k_list   RemoteCache-style point ops and scans on the table
//   k_synth_*        synthetic batch generator
#pragma once
#include "ctmr_dev.h"
#include "entry_decode.h"
#include "synth.h"

namespace ctmr {

struct __attribute__((packed, aligned(1))) U16t { uint32_t a, b, c, d; };  // unaligned 16-byte access

// ------------------------------------------------------------------ byte readers
// 4-byte little-endian window at an arbitrary byte position, from two aligned dwords.
struct LdsReader {
  const uint32_t* lds;  // tile words (LDS)
  uint32_t base;        // byte offset of this certificate inside the tile
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t a = base + pos;
    const uint32_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(lds[i + 1], lds[i], a & 3u);
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  __device__ __forceinline__ void touch(uint32_t, uint32_t) const {}
  __device__ __forceinline__ void touch_tail(uint32_t, uint32_t) const {}
};

struct GlobalReader {
  const uint32_t* words;  // 4-byte aligned base of the buffer (kernel argument: global address space)
  uint64_t base;          // byte offset of this certificate inside the buffer
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(words[i + 1], words[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  __device__ __forceinline__ void touch(uint32_t, uint32_t) const {}
  __device__ __forceinline__ void touch_tail(uint32_t, uint32_t) const {}
};

// Per-lane LDS window ("software cache") over a certificate that stays in global memory.
// Each lane owns WCH 16-byte chunks of LDS (lane stride WCH*16+16 bytes: 16-B aligned for
// ds_write_b128, and ≤4-way bank conflicts on the dword reads).  touch(pos, need) refills the
// window with WCH independent global_load_dwordx4 (one burst, one memory latency) when the next
// `need` bytes are not resident; ld4 hits LDS inside the window and falls back to a plain global
// load outside it — so correctness never depends on where the window is.  The walk touches the
// front of the certificate and the extension block; SPKI body, SAN body and signature are
// skipped by length and therefore never fetched from HBM.
template <int WCH>
struct WinReader {
  const uint32_t* g32;  // payload, dword view (global)
  uint64_t base;        // certificate start (byte offset into payload)
  uint64_t limit;       // readable bytes of payload (offsets[n] + CTMR_PAYLOAD_PAD)
  uint32_t* win;        // this lane's window words in LDS
  int32_t grel;         // window start relative to the certificate start; (base+grel) % 16 == 0
  static constexpr uint32_t WBYTES = WCH * 16;

  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel <= WBYTES - 8u) {
      const uint32_t i = rel >> 2;
      return __builtin_amdgcn_alignbyte(win[i + 1], win[i], rel & 3u);
    }
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const {  // straight from global memory
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ void refill(uint32_t pos) {
    const uint64_t g = (base + pos) & ~15ull;
    grel = (int32_t)(int64_t)(g - base);
    const uint4* src = (const uint4*)g32 + (g >> 4);
    uint4 v[WCH];
#pragma unroll
    for (int k = 0; k < WCH; k++)
      v[k] = (g + 16u * k + 16u <= limit) ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WCH; k++) ((uint4*)win)[k] = v[k];
  }
  __device__ __forceinline__ void touch(uint32_t pos, uint32_t need) {
    if (need > WBYTES - 16u) need = WBYTES - 16u;  // the window start is 16-B aligned in HBM
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel > WBYTES - need) refill(pos);
  }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t) { touch(pos, 256); }
};

// WinReader whose touch_tail() (the one refill every lane of the wave reaches at the same program point,
// right behind the SubjectPublicKeyInfo header) is wave-cooperative like the first fill of k_map_winc:
// 16 adjacent lanes fetch the 16 chunks of one certificate's window, 4 certificates per load
// instruction.  Falls back to the per-lane refill when some lane of the wave is not at that point.
template <int WCH>
struct WinReaderC : WinReader<WCH> {
  static constexpr uint32_t STRIDE = WCH * 16 + 16;
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t) {
    if (__ballot(1) != ~0ull) {
      this->refill(pos);
      return;
    }
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 15u;
    const uint64_t g_me = (this->base + pos) & ~15ull;
    this->grel = (int32_t)(int64_t)(g_me - this->base);
    uint8_t* lds0 = (uint8_t*)this->win - lane * STRIDE;
    uint4 v[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint64_t g = __shfl(g_me, 4 * it + (int)(lane >> 4));
      const uint64_t at = g + 16u * sub;
      v[it] = (at + 16u <= this->limit) ? *((const uint4*)this->g32 + (at >> 4)) : make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 16; it++)
      *(uint4*)(lds0 + (4 * it + (lane >> 4)) * STRIDE + 16u * sub) = v[it];
    __builtin_amdgcn_wave_barrier();
  }
};

// WinReaderC whose ld4() is served by the LDS window ALONE: no per-access "outside the window → global load"
// branch (88 ld4 per certificate, each of which used to carry its own exec-mask dance).  An access that does fall
// outside is clamped and remembered in `miss`; the kernel then repeats that certificate with the exact GlobalReader.
// On well-formed certificates the walk's touch() hints keep every ld4 inside (measured on the synthetic corpus: 0
// misses in 200 000 certificates once the three reads behind the TBS go through ldg()).
template <int WCH>
struct WinReaderS : WinReaderC<WCH> {
  static constexpr bool kNoClamp = true;  // ld4 clamps into the window itself
  mutable uint32_t miss;
  // The 32 bytes behind the TBSCertificate (signatureAlgorithm, the signatureValue header, its pad octet), fetched
  // by touch_tail() TOGETHER with the extension-block refill: the three ldg() reads at the end of the walk were
  // three dependent, uncoalesced global round trips per wave; now they are register selects.
  uint32_t tl[8];
  uint32_t tl_pos;  // certificate offset of tl[0]'s first byte; 0x80000000 = not fetched (positions are < 2^31)
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    uint32_t rel = pos - (uint32_t)this->grel;
    constexpr uint32_t LAST = WinReader<WCH>::WBYTES - 8u;
    miss |= (uint32_t)(rel > LAST);
    rel = rel > LAST ? LAST : rel;
    const uint32_t i = rel >> 2;
    return __builtin_amdgcn_alignbyte(this->win[i + 1], this->win[i], rel & 3u);
  }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t tail) {
    const uint64_t ta = this->base + tail;
    const bool have = ta + 32u <= this->limit;
    const uint8_t* tp = (const uint8_t*)this->g32 + (have ? ta : 0ull);
    const U16t a = *(const U16t*)tp, b = *(const U16t*)(tp + 16);  // in flight with the refill below
    WinReaderC<WCH>::touch_tail(pos, tail);
    tl[0] = a.a; tl[1] = a.b; tl[2] = a.c; tl[3] = a.d;
    tl[4] = b.a; tl[5] = b.b; tl[6] = b.c; tl[7] = b.d;
    tl_pos = have ? tail : 0x80000000u;
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const {
    const uint32_t off = pos - tl_pos;
    const uint32_t wi = off >> 2;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) {
      lo = wi == k ? tl[k] : lo;
      hi = wi == k ? tl[k + 1] : hi;
    }
    uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
    if (off > 27u) v = WinReader<WCH>::ldg(pos);  // a long AlgorithmIdentifier, or no prefetch: the real load
    return v;
  }
};

// Line-trimmed window.  HBM is fetched in 128-byte lines (scripts/calib_fetch.hip: FETCH_SIZE x2 equals
// the unique 128-B lines of every window pattern tried), so a refill that ends in the middle of a line
// pays for the whole line and keeps only part of it.  This reader ends every refill at the end of the
// line that holds byte pos+N-1 (N = bytes the walk is expected to need from there: NF for the front of
// the certificate, NE for the extension block and on-demand refills), capped at WCH chunks: a 256-byte
// refill at a random 16-byte phase touches 2.875 lines on average, a trimmed one 2.4.  Shorter windows
// only ever cost an extra refill — ld4 falls back to global loads outside the window as before.
template <int WCH, int NF, int NE>
struct WinReaderT {
  const uint32_t* g32;
  uint64_t base;
  uint64_t limit;
  uint32_t* win;
  int32_t grel;
  uint32_t wlen;  // valid bytes in the window (multiple of 16)

  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel + 8u <= wlen && rel < 0x7fffffffu) {
      const uint32_t i = rel >> 2;
      return __builtin_amdgcn_alignbyte(win[i + 1], win[i], rel & 3u);
    }
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ void refill(uint32_t pos, uint32_t n) {
    const uint64_t p = base + pos;
    const uint64_t g = p & ~15ull;
    const uint64_t end = ((p + n - 1u) | 127ull) + 1ull;
    grel = (int32_t)(int64_t)(g - base);
    uint32_t cnt = (uint32_t)((end - g) >> 4);
    cnt = cnt < (uint32_t)WCH ? cnt : (uint32_t)WCH;
    const uint64_t room = limit > g ? (limit - g) >> 4 : 0ull;
    cnt = room < cnt ? (uint32_t)room : cnt;
    wlen = cnt * 16u;
    const uint4* src = (const uint4*)g32 + (g >> 4);
    uint4 v[WCH];
#pragma unroll
    for (int k = 0; k < WCH; k++) v[k] = (uint32_t)k < cnt ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WCH; k++) ((uint4*)win)[k] = v[k];
  }
  __device__ __forceinline__ void touch(uint32_t pos, uint32_t need) {
    if (need > (uint32_t)NE) need = NE;
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel + need > wlen || rel >= 0x7fffffffu) refill(pos, NE);
  }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t) { touch(pos, NE); }
};

// Two-region window: MAIN (WCH chunks, moves with the walk) + TAIL (3 chunks pinned at the end of
// the TBSCertificate: signatureAlgorithm and the BIT STRING header of signatureValue).  The walk
// knows both addresses as soon as it has decoded the SubjectPublicKeyInfo header — the extension
// block starts right behind the key, the tail at tbs_end — so touch_tail() fetches both regions
// in ONE burst of WCH+3 independent global_load_dwordx4: the dependent HBM round trips per
// certificate drop from ≈8 (front, [3] tag, extensions, sigalg header, BIT STRING header, pad
// byte, last byte, …) to 2 (front; extensions + tail).  Lane stride (WCH+3)·16 B with WCH even:
// an odd number of 16-B chunks keeps the dword reads at ≤4-way bank conflicts without a pad chunk.
template <int WCH>
struct WinReader2 {
  static constexpr int TCH = 3;
  static constexpr uint32_t WBYTES = WCH * 16, TBYTES = TCH * 16;
  const uint32_t* g32;
  uint64_t base;
  uint64_t limit;
  uint32_t* win;  // main window words; the tail window follows at win + WCH*4
  int32_t grel;   // main window start relative to the certificate start
  int32_t trel;   // tail window start (0x7fffff00 = not loaded)

  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel <= WBYTES - 8u) {
      const uint32_t i = rel >> 2;
      return __builtin_amdgcn_alignbyte(win[i + 1], win[i], rel & 3u);
    }
    const uint32_t rel2 = pos - (uint32_t)trel;
    if (rel2 <= TBYTES - 8u) {
      const uint32_t i = WCH * 4 + (rel2 >> 2);
      return __builtin_amdgcn_alignbyte(win[i + 1], win[i], rel2 & 3u);
    }
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ void refill(uint32_t pos) {
    const uint64_t g = (base + pos) & ~15ull;
    grel = (int32_t)(int64_t)(g - base);
    const uint4* src = (const uint4*)g32 + (g >> 4);
    uint4 v[WCH];
#pragma unroll
    for (int k = 0; k < WCH; k++)
      v[k] = (g + 16u * k + 16u <= limit) ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WCH; k++) ((uint4*)win)[k] = v[k];
  }
  __device__ __forceinline__ void touch(uint32_t pos, uint32_t need) {
    if (need > WBYTES - 16u) need = WBYTES - 16u;
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel > WBYTES - need) refill(pos);
  }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t tailpos) {
    const uint64_t g = (base + pos) & ~15ull;
    const uint64_t t = (base + tailpos) & ~15ull;
    grel = (int32_t)(int64_t)(g - base);
    trel = (int32_t)(int64_t)(t - base);
    const uint4* src = (const uint4*)g32 + (g >> 4);
    const uint4* tsrc = (const uint4*)g32 + (t >> 4);
    uint4 v[WCH], u[TCH];
#pragma unroll
    for (int k = 0; k < WCH; k++)
      v[k] = (g + 16u * k + 16u <= limit) ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < TCH; k++)
      u[k] = (t + 16u * k + 16u <= limit) ? tsrc[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WCH; k++) ((uint4*)win)[k] = v[k];
#pragma unroll
    for (int k = 0; k < TCH; k++) ((uint4*)win)[WCH + k] = u[k];
  }
};

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
__global__ void __launch_bounds__(64) k_issuer_ids(const uint8_t* der, const uint64_t* offsets,
                                                   uint32_t n, uint8_t* valid, uint32_t* digest) {
  __shared__ uint32_t kc[64];
  kc[threadIdx.x] = K256[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  GlobalReader r{(const uint32_t*)der, offsets[i]};
  const uint32_t L = (uint32_t)(offsets[i + 1] - offsets[i]);
  Walk w;
  const bool ok = (offsets[i + 1] - offsets[i]) <= 0x7fffffffull && walk_cert(r, L, w);
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

// ------------------------------------------------------------------ the map
struct MapArgs {
  const uint8_t* payload;
  const uint64_t* offsets;    // packed batch: n+1 offsets; entry view (ends != null): n range starts
  const uint64_t* ends;       // null = packed batch; else certificate i is [offsets[i], ends[i]) (ctmr_entry_view)
  uint64_t limit;             // entry view: readable bytes of payload (blob bytes + CTMR_PAYLOAD_PAD)
  const uint32_t* issuer_idx;
  const uint8_t* entry_type;  // may be null
  ctmr_record* records;
  const uint8_t* issuer_valid;
  const FilterDev* filt;
  uint64_t n;
  uint32_t n_issuers;
  uint32_t certs_per_tile;
  uint32_t lds_bytes;  // dynamic LDS size of the launch
  uint2* meta_loc;          // null, or per entry (Walk.meta_issuer, Walk.meta_crl) for k_meta_new (config.collect_meta)
  uint32_t optimistic_new;  // 1: PASS records leave the map with CTMR_FL_WAS_UNKNOWN already set — the
                            // reduce only CLEARS it for the (rare) duplicates, so the common case costs
                            // no second scattered write into the record array
};

extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

// Byte range of certificate i and the readable size of the payload, for both input forms.
__device__ __forceinline__ void cert_range(const uint64_t* offsets, const uint64_t* ends, uint64_t i, uint64_t& lo,
                                           uint64_t& hi) {
  lo = offsets[i];
  hi = ends ? ends[i] : offsets[i + 1];
  if (hi < lo) hi = lo;
}
__device__ __forceinline__ uint64_t map_limit(const MapArgs& a) {
  return a.ends ? a.limit : a.offsets[a.n] + CTMR_PAYLOAD_PAD;
}

// Everything after the bytes are addressable: walk, filters, record.
template <class R>
__device__ __forceinline__ void map_one(R& r, uint64_t len64, uint64_t idx, const MapArgs& a, uint4& o0,
                                        uint4& o1) {
  Walk w;
  const uint32_t L = len64 <= 0x7fffffffull ? (uint32_t)len64 : 0x80000000u;
  const FilterDev* f = a.filt;
  const FilterView fv{f->n_pieces, f->piece_len, f->piece_word, f->words};
  const bool ok = walk_cert(r, L, w, f->active != 0u, fv);  // L > 2^31-1 is rejected inside, without divergence
  const uint32_t iss = a.issuer_idx[idx];
  const uint32_t et = a.entry_type ? a.entry_type[idx] : 0u;
  uint32_t status;
  if (et == CTMR_ENTRY_INVALID) {
    status = CTMR_ST_ENTRY_DECODE_ERROR;  // never reached entryChan (ct-fetch.go:452-459)
  } else if (!ok) {
    status = CTMR_ST_PARSE_ERROR;
  } else if (w.bc_valid && w.is_ca) {
    status = CTMR_ST_FILTERED_CA;
  } else if (w.not_after < f->now && !f->log_expired) {
    status = CTMR_ST_FILTERED_EXPIRED;
  } else if (!w.cn_match) {
    status = CTMR_ST_FILTERED_CN;
  } else if (iss == CTMR_NO_ISSUER || iss >= a.n_issuers) {
    status = CTMR_ST_NO_ISSUER;
  } else if (!a.issuer_valid[iss]) {
    status = CTMR_ST_ISSUER_PARSE_ERROR;
  } else {
    status = CTMR_ST_PASS;
  }
  uint32_t flags = et == 1u ? CTMR_FL_PRECERT : 0u;
  if (a.optimistic_new && status == CTMR_ST_PASS) flags |= CTMR_FL_WAS_UNKNOWN;
  uint32_t slen = 0, s[5] = {0, 0, 0, 0, 0};
  int32_t exp_hour = 0;
  if (ok) {
    // NewExpDateFromTime: Truncate(time.Hour) = floor (storage/types.go:339-346)
    long long q = w.not_after / 3600;
    if (w.not_after % 3600 < 0) q -= 1;
    exp_hour = (int32_t)q;
    slen = w.serial_len > 0xffffu ? 0xffffu : w.serial_len;
    if (w.serial_len > 20) flags |= CTMR_FL_LONG_SERIAL;
#pragma unroll
    for (int k = 0; k < 5; k++) s[k] = w.serial_w[k];
  }
  o0 = make_uint4(status | (flags << 8) | (slen << 16), (uint32_t)exp_hour, iss, s[0]);
  o1 = make_uint4(s[1], s[2], s[3], s[4]);
  if (a.meta_loc) a.meta_loc[idx] = make_uint2(ok ? w.meta_issuer : META_NONE, ok ? w.meta_crl : META_NONE);
}

template <class R>
__device__ __forceinline__ void map_one(R& r, uint64_t len64, uint64_t idx, const MapArgs& a) {
  uint4 o0, o1;
  map_one(r, len64, idx, a, o0, o1);
  uint4* out = (uint4*)(a.records + idx);
  out[0] = o0;
  out[1] = o1;
}

// Record store for the one-wave-per-workgroup window kernels: the 64 records of the wave (2 KiB,
// contiguous) are transposed through LDS so that each of the two store instructions writes 1 KiB of
// consecutive bytes (whole 64-B sectors) instead of 64 half-sectors 32 B apart.  The window area is
// free by now: every lane of the wave has finished its walk.
__device__ __forceinline__ void store_records_wave(const MapArgs& a, uint64_t first, bool live, const uint4& o0,
                                                   const uint4& o1) {
  uint4* t = (uint4*)smem;
  const uint32_t lane = threadIdx.x;
  __builtin_amdgcn_wave_barrier();
  if (live) {
    t[2 * lane] = o0;
    t[2 * lane + 1] = o1;
  }
  __builtin_amdgcn_wave_barrier();
  const uint64_t rem = a.n - first;  // records of this wave
  const uint32_t nvec = rem >= 64 ? 128u : (uint32_t)rem * 2u;
  uint4* out = (uint4*)(a.records + first);
  if (lane < nvec) out[lane] = t[lane];
  if (64u + lane < nvec) out[64 + lane] = t[64 + lane];
}

// LDS-tile map.  One wave per workgroup, one tile of `certs_per_tile` consecutive
// certificates per workgroup: the tile's byte range [offsets[first], offsets[last+1]) is
// contiguous in the packed payload, so it is copied with perfectly coalesced 16-B/lane loads
// (1 KiB per wave instruction) into LDS; then lane l walks certificate first+l out of LDS.

__global__ void __launch_bounds__(64) k_map_tile(MapArgs a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t C = a.certs_per_tile;
  const uint64_t first = (uint64_t)blockIdx.x * C;
  if (first >= a.n) return;
  const uint32_t cnt = (uint32_t)((a.n - first) < C ? (a.n - first) : C);
  uint64_t my_lo = 0, my_hi = 0;
  if (lane < cnt) {
    my_lo = a.offsets[first + lane];
    my_hi = a.offsets[first + lane + 1];
  }
  const uint64_t tile_lo = __shfl(my_lo, 0);
  const uint64_t tile_hi = __shfl(my_hi, cnt - 1);
  const uint64_t a_lo = tile_lo & ~15ull;
  const uint64_t span = tile_hi - a_lo;
  if (tile_hi < tile_lo || span + 48 > a.lds_bytes) {
    // oversize (or malformed offsets): walk straight from global memory
    if (lane < cnt) {
      if (my_hi < my_lo) my_hi = my_lo;
      GlobalReader r{(const uint32_t*)a.payload, my_lo};
      map_one(r, my_hi - my_lo, first + lane, a);
    }
    return;
  }
  // ---- stage the tile: global → VGPR → LDS, 8 × 1 KiB in flight per wave
  {
    const uint4* src = (const uint4*)(a.payload + a_lo);
    uint4* dst = (uint4*)smem;
    const uint32_t nvec = (uint32_t)((span + 15) >> 4);
    for (uint32_t base = 0; base < nvec; base += 8 * 64) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t i = base + k * 64 + lane;
        if (i < nvec) v[k] = src[i];
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t i = base + k * 64 + lane;
        if (i < nvec) dst[i] = v[k];
      }
    }
  }
  __syncthreads();
  if (lane < cnt) {
    if (my_hi < my_lo) my_hi = my_lo;
    LdsReader r{(const uint32_t*)smem, (uint32_t)(my_lo - a_lo)};
    map_one(r, my_hi - my_lo, first + lane, a);
  }
}

// Direct map: one certificate per lane straight from global memory.
__global__ void __launch_bounds__(256) k_map_direct(MapArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  uint64_t lo, hi;
  cert_range(a.offsets, a.ends, i, lo, hi);
  GlobalReader r{(const uint32_t*)a.payload, lo};
  map_one(r, hi - lo, i, a);
}

// Window map: one certificate per lane, all 64 lanes busy, DER stays in global memory and is
// pulled through a per-lane LDS window (WinReader).  One wave per workgroup, so LDS (not the
// 256-thread granule) sets the occupancy: 64 × (WCH·16+16) bytes per wave.
template <int WCH>
__global__ void __launch_bounds__(64) k_map_win(MapArgs a) {
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint64_t i = first + threadIdx.x;
  const bool live = i < a.n;
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  if (live) {
    uint64_t lo, hi;
    cert_range(a.offsets, a.ends, i, lo, hi);
    constexpr uint32_t STRIDE = WCH * 16 + 16;
    WinReader<WCH> r{(const uint32_t*)a.payload, lo, map_limit(a),
                     (uint32_t*)(smem + threadIdx.x * STRIDE), 0};
    r.refill(0);
    map_one(r, hi - lo, i, a, o0, o1);
  }
  store_records_wave(a, first, live, o0, o1);
}

// Window map with a wave-cooperative first fill: instead of every lane issuing 16 loads of ITS certificate
// (64 uncoalesced 16-byte requests per instruction), 16 adjacent lanes fetch the 16 chunks of one
// certificate's front window, 4 certificates per instruction — the texture addresser sees 8 lanes per
// 128-byte line — and each lane parks its chunk directly in the owning lane's LDS window.
template <int WCH>
__global__ void __launch_bounds__(64) k_map_winc(MapArgs a) {
  static_assert(WCH == 16, "cooperative fill assumes 16 chunks");
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint32_t lane = threadIdx.x;
  const uint64_t i = first + lane;
  const bool live = i < a.n;
  constexpr uint32_t STRIDE = WCH * 16 + 16;
  const uint64_t limit = map_limit(a);
  uint64_t lo = 0, hi = 0;
  if (live) cert_range(a.offsets, a.ends, i, lo, hi);
  const uint64_t g_me = live ? (lo & ~15ull) : ~0ull;
  {
    uint4 v[16];
    const uint32_t sub = lane & 15u;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint64_t g = __shfl(g_me, 4 * it + (int)(lane >> 4));
      const uint64_t at = g + 16u * sub;
      v[it] = (g != ~0ull && at + 16u <= limit) ? *(const uint4*)(a.payload + at) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 16; it++)
      *(uint4*)(smem + (4 * it + (lane >> 4)) * STRIDE + 16u * sub) = v[it];
  }
  __builtin_amdgcn_wave_barrier();
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  if (live) {
    WinReaderC<WCH> r{{(const uint32_t*)a.payload, lo, limit, (uint32_t*)(smem + lane * STRIDE),
                       (int32_t)(int64_t)(g_me - lo)}};
    map_one(r, hi - lo, i, a, o0, o1);
  }
  store_records_wave(a, first, live, o0, o1);
}

// Line-trimmed window map (WinReaderT).
template <int WCH, int NF, int NE>
__global__ void __launch_bounds__(64) k_map_wint(MapArgs a) {
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint64_t i = first + threadIdx.x;
  const bool live = i < a.n;
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  if (live) {
    uint64_t lo, hi;
    cert_range(a.offsets, a.ends, i, lo, hi);
    constexpr uint32_t STRIDE = WCH * 16 + 16;
    WinReaderT<WCH, NF, NE> r{(const uint32_t*)a.payload, lo, map_limit(a),
                              (uint32_t*)(smem + threadIdx.x * STRIDE), 0, 0};
    r.refill(0, NF);
    map_one(r, hi - lo, i, a, o0, o1);
  }
  store_records_wave(a, first, live, o0, o1);
}

// Two-region window map (WinReader2): same walk, 2 dependent HBM round trips per certificate.
template <int WCH>
__global__ void __launch_bounds__(64) k_map_win2(MapArgs a) {
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint64_t i = first + threadIdx.x;
  const bool live = i < a.n;
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  if (live) {
    uint64_t lo, hi;
    cert_range(a.offsets, a.ends, i, lo, hi);
    constexpr uint32_t STRIDE = (WCH + 3) * 16;
    WinReader2<WCH> r{(const uint32_t*)a.payload, lo, map_limit(a),
                      (uint32_t*)(smem + threadIdx.x * STRIDE), 0, 0x7fffff00};
    r.refill(0);
    map_one(r, hi - lo, i, a, o0, o1);
  }
  store_records_wave(a, first, live, o0, o1);
}

// ------------------------------------------------------------------ the reduce
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT);
}

// Find-or-insert one key.  Returns the slot index (or SID_FULL); *created tells whether this
// call claimed the slot.  idx32 = batch index merged with atomicMin (pass 0xffffffff for
// point operations).  Visibility: payload words are written through (agent-scope atomic
// stores), drained with s_waitcnt vmcnt(0), then w[1] is published; readers poll w[1] with
// agent-scope loads (MI355X_MICROARCH.md "handoff-flag": write-through payload + drained flag).
__device__ __forceinline__ uint32_t table_upsert(Slot* table, uint64_t mask, unsigned long long meta,
                                                 const unsigned long long s[5], uint32_t idx32,
                                                 uint32_t epoch, bool insert, bool* created,
                                                 unsigned long long* prev_w0 = nullptr) {
  const unsigned long long h = key_hash(meta, s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & mask;
  *created = false;
  uint64_t probes = 0;
  for (;;) {
    Slot* sl = table + j;
    unsigned long long w0 = ld_agent(&sl->w[0]);
    if (w0 == 0ull) {
      if (!insert) return SID_NONE;
      const unsigned long long old = atomicCAS(&sl->w[0], 0ull, tagw | idx32);
      if (old == 0ull) {
        st_agent(&sl->w[2], (unsigned long long)epoch);
#pragma unroll
        for (int k = 0; k < 5; k++) st_agent(&sl->w[3 + k], s[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(&sl->w[1], meta);
        *created = true;
        return (uint32_t)j;
      }
      w0 = old;
    }
    if ((w0 & 0xffffffff00000000ull) == tagw && w0 != SLOT_TOMB) {
      const unsigned long long m = ld_agent(&sl->w[1]);
      if (!(m & SLOT_VALID)) continue;  // creator has not published yet: poll again
      bool eq = m == meta;
#pragma unroll
      for (int k = 0; k < 5; k++) eq = eq && ld_agent(&sl->w[3 + k]) == s[k];
      if (eq) {
        if (insert && idx32 != 0xffffffffu) {
          const unsigned long long old = atomicMin(&sl->w[0], tagw | idx32);
          if (prev_w0) *prev_w0 = old;
        }
        return (uint32_t)j;
      }
    }
    j = (j + 1) & mask;
    if (++probes > mask) return SID_FULL;
  }
}

struct InsertArgs {
  const ctmr_record* records;
  const uint8_t* payload;  // for serials longer than the 20 octets a record carries
  const uint64_t* offsets;
  const uint64_t* ends;    // null = packed batch (MapArgs)
  const uint32_t* canon;   // issuer_idx → canonical issuer
  Slot* table;
  uint64_t mask;
  uint32_t* slot_id;       // candidate slot of DEFER entries (written for those only)
  uint32_t* ent;           // per entry: status(0..2) | state(3..5) | canonical issuer << 8
  uint64_t n;
  uint32_t epoch;
};

// Per-entry state of the reduce (bits 3..5 of ent[i]); the low 3 bits carry record.status.
enum : uint32_t {
  ES_NONE = 0,     // did not reach the set (filtered / parse error / no issuer)
  ES_CLAIMED = 1,  // claimed an empty slot: WasUnknown unless a lower log index of the same key marks it
  ES_DEFER = 2,    // met a same-tag slot of this batch: decided in pass 2; WasUnknown unless marked
  ES_DUP = 3,      // known: since an earlier batch, or a lower log index of this batch holds the key
  ES_HOST = 4,     // serial longer than CTMR_MAX_SERIAL: exact host-side set
  ES_FULL = 5      // table full
};
__device__ __forceinline__ uint32_t ent_pack(uint32_t status, uint32_t state, uint32_t canon) {
  return (status & 7u) | (state << 3) | (canon << 8);
}
__device__ __forceinline__ uint32_t ent_state(uint32_t e) { return (e >> 3) & 7u; }
__device__ __forceinline__ bool ent_is_new(uint32_t e) {
  const uint32_t st = ent_state(e);
  return (st == ES_CLAIMED) | (st == ES_DEFER);
}
// a PASS entry lost to a lower log index of the same key: its ent byte 0 and its record flag
__device__ __forceinline__ void mark_dup(uint32_t* ent, ctmr_record* records, uint32_t loser) {
  ((uint8_t*)(ent + loser))[0] = (uint8_t)(CTMR_ST_PASS | (ES_DUP << 3));
  uint8_t* fl = (uint8_t*)(records + loser) + 1;
  *fl = (uint8_t)(*fl & ~CTMR_FL_WAS_UNKNOWN);
}

// Offset of the serialNumber content octets (certificate already accepted by the map).
__device__ __forceinline__ uint32_t serial_content_off(const GlobalReader& r, uint32_t L) {
  bool ok = true;
  uint32_t tag, cs, ce;
  rd_hdr(r, L, 0, L, ok, tag, cs, ce);
  rd_hdr(r, L, cs, L, ok, tag, cs, ce);
  uint32_t q = cs;
  if ((r.ld4(q) & 0xffu) == 0xa0u) {
    rd_hdr(r, L, q, L, ok, tag, cs, ce);
    q = ce;
  }
  rd_hdr(r, L, q, L, ok, tag, cs, ce);
  return cs;
}

__device__ __forceinline__ void record_key(const InsertArgs& a, uint64_t i, const uint4& r0,
                                           const uint4& r1, unsigned long long s[5]) {
  const uint32_t slen = r0.x >> 16;
  s[0] = (unsigned long long)r0.w | ((unsigned long long)r1.x << 32);
  s[1] = (unsigned long long)r1.y | ((unsigned long long)r1.z << 32);
  s[2] = (unsigned long long)r1.w;
  s[3] = 0;
  s[4] = 0;
  if (slen > 20) {
    // octets 20..slen-1 come from the certificate itself
    uint64_t lo, hi;
    cert_range(a.offsets, a.ends, i, lo, hi);
    GlobalReader g{(const uint32_t*)a.payload, lo};
    const uint32_t so = serial_content_off(g, (uint32_t)(hi - lo));
    uint32_t x[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint32_t pos = 20u + 4u * k;
      if (pos < slen) {
        const uint32_t rem = slen - pos;
        const uint32_t v = g.ld4(so + pos);
        x[k] = rem >= 4 ? v : (v & (0xffffffffu >> (8 * (4 - rem))));
      }
    }
    s[2] |= (unsigned long long)x[0] << 32;
    s[3] = (unsigned long long)x[1] | ((unsigned long long)x[2] << 32);
    s[4] = (unsigned long long)x[3] | ((unsigned long long)x[4] << 32);
  }
}

// KnownCertificates.WasUnknown → RemoteCache.SetInsert (knowncertificates.go:38-55) for every
// PASS entry, against the in-HBM table.  PASS 1 (this kernel) never reads anything another lane
// of the same launch wrote except the CAS word itself:
//   empty slot      → one atomicCAS claims it (state CLAIMED); the 64-byte slot image is written
//                     after the probe loop, four lanes per slot, so that one store instruction emits
//                     whole 64-byte slots (one memory transaction each) instead of four partial ones
//   same tag, slot of an OLDER batch (epoch in [1, cur)) → fully visible: compare now (state DUP)
//   same tag, slot of THIS batch (epoch 0 = not written yet, or cur) → remember the slot (state
//                     DEFER), decide in pass 2 after the kernel boundary made every pass-1 store visible
// Records arrive with WAS_UNKNOWN set optimistically by the map; it is cleared here / in pass 2
// for duplicates only.  The reduce's later passes read the 4-byte ent[] word, never the table.
// Pass-1 set insert of one PASS record held in registers (r0, r1 = the two 16-byte halves of the record).
// On a claim the 64-byte slot image is returned in q0..q3 and `claimed` is the slot index; the caller
// stores it cooperatively (store_slots_wave).  Returns the ES_* state.
__device__ __forceinline__ uint32_t insert_probe(const InsertArgs& a, uint64_t i, const uint4& r0, const uint4& r1,
                                                 uint32_t canon, uint64_t& claimed, uint4& q0, uint4& q1, uint4& q2,
                                                 uint4& q3) {
  const uint32_t slen = r0.x >> 16;
  if (slen > CTMR_MAX_SERIAL) return ES_HOST;
  unsigned long long s[5];
  record_key(a, i, r0, r1, s);
  const unsigned long long meta = key_meta((int32_t)r0.y, canon, slen);
  const unsigned long long h = key_hash(meta, s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & a.mask;
  for (uint64_t probes = 0; probes <= a.mask; probes++) {
    Slot* sl = a.table + j;
    const unsigned long long w0 = tagw | (uint32_t)i;
    const unsigned long long old = atomicCAS(&sl->w[0], 0ull, w0);
    if (old == 0ull) {  // claimed
      q0 = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)meta, (uint32_t)(meta >> 32));
      q1 = make_uint4(a.epoch, 0u, (uint32_t)s[0], (uint32_t)(s[0] >> 32));
      q2 = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
      q3 = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
      claimed = j;
      return ES_CLAIMED;
    }
    if ((old & 0xffffffff00000000ull) == tagw) {
      const uint32_t ep = (uint32_t)ld_agent(&sl->w[2]);
      if (ep != 0u && ep != a.epoch) {  // older batch: complete and visible
        bool eq = sl->w[1] == meta;
#pragma unroll
        for (int k = 0; k < 5; k++) eq = eq && sl->w[3 + k] == s[k];
        if (eq) return ES_DUP;
      } else {
        a.slot_id[i] = (uint32_t)j;
        return ES_DEFER;
      }
    }
    j = (j + 1) & a.mask;
  }
  return ES_FULL;
}

// Cooperative slot write of one wave: lane L parks its 64-byte image at img[L*4 .. L*4+3]; store
// instruction r then has lane L write quarter L%4 of the slot of lane 16r + L/4, so four adjacent
// lanes emit one whole slot.  (w[0] is rewritten with the value the CAS stored: concurrent CAS
// attempts of this pass see a non-zero word either way; atomicMin only runs in pass 2.)
__device__ __forceinline__ void store_slots_wave(Slot* table, uint4* img, uint32_t lane, uint64_t claimed,
                                                 const uint4& q0, const uint4& q1, const uint4& q2, const uint4& q3) {
  uint4* my = img + lane * 4;
  my[0] = q0; my[1] = q1; my[2] = q2; my[3] = q3;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t src = 16u * r + (lane >> 2);
    const uint64_t sj = __shfl(claimed, src);
    if (sj != ~0ull) {
      const uint4 v = img[r * 64 + lane];
      ((uint4*)(table + sj))[lane & 3u] = v;
    }
  }
}

__global__ void __launch_bounds__(256) k_insert(InsertArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 img[4][64 * 4];  // per wave: 64 slot images
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t state = ES_NONE, status = CTMR_ST__COUNT, canon = 0;
  uint64_t claimed = ~0ull;  // slot index when this lane claimed one
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
  if (i < a.n) {
    const uint4* rp = (const uint4*)(a.records + i);
    const uint4 r0 = rp[0];
    status = r0.x & 0xffu;
    if (status == CTMR_ST_PASS) {
      canon = a.canon[r0.z];
      const uint4 r1 = rp[1];
      state = insert_probe(a, i, r0, r1, canon, claimed, q0, q1, q2, q3);
      if (state != ES_CLAIMED && state != ES_DEFER) {  // not (yet) unknown: drop the optimistic flag
        uint8_t* fl = (uint8_t*)(a.records + i) + 1;
        *fl = (uint8_t)((r0.x >> 8) & ~CTMR_FL_WAS_UNKNOWN);
      }
    }
    a.ent[i] = ent_pack(status, state, canon);
  }
  store_slots_wave(a.table, img[wv], lane, claimed, q0, q1, q2, q3);
}

// PASS 2: DEFER entries — their candidate slot was created by this batch and is complete now.
// Equal key → atomicMin the batch index into w[0]; the RETURNED previous minimum tells who loses:
// whichever of (previous holder, me) has the higher log index is marked DUP, so after this pass
// exactly the lowest log index of every new key is still CLAIMED/DEFER — nobody has to re-read the
// table to find out.  A 32-bit tag collision between different keys (≈2^-32 per probe) falls back
// to the fully synchronised upsert, which is also safe against other pass-2 lanes inserting the
// same key concurrently.
__global__ void __launch_bounds__(256) k_insert2(InsertArgs a, ctmr_record* records) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t e = a.ent[i];
  if (ent_state(e) != ES_DEFER) return;
  const uint32_t sid = a.slot_id[i];
  const uint4* rp = (const uint4*)(a.records + i);
  const uint4 r0 = rp[0], r1 = rp[1];
  unsigned long long s[5];
  record_key(a, i, r0, r1, s);
  const unsigned long long meta = key_meta((int32_t)r0.y, e >> 8, r0.x >> 16);
  Slot* sl = a.table + sid;
  bool eq = sl->w[1] == meta;
#pragma unroll
  for (int k = 0; k < 5; k++) eq = eq && sl->w[3 + k] == s[k];
  unsigned long long prev = ~0ull;
  if (eq) {
    const unsigned long long tagw = (unsigned long long)key_tag(key_hash(meta, s)) << 32;
    prev = atomicMin(&sl->w[0], tagw | (uint32_t)i);
  } else {
    bool created;
    const uint32_t r = table_upsert(a.table, a.mask, meta, s, (uint32_t)i, a.epoch, true, &created, &prev);
    if (r == SID_FULL) {
      ((uint8_t*)(a.ent + i))[0] = (uint8_t)(CTMR_ST_PASS | (ES_FULL << 3));
      return;
    }
    if (created) return;  // stays DEFER = unknown unless a lower index joins and marks it
  }
  const uint32_t other = (uint32_t)prev;
  mark_dup(a.ent, records, other < (uint32_t)i ? (uint32_t)i : other);
}

// Fused map + pass-1 insert (variant 14): the lane that just finished walking a certificate probes the
// known-certificate table straight from its registers — the 32-byte record is not re-read (−3.2 GB per
// 100 M entries), the WAS_UNKNOWN flag is final before the record is stored (no second scattered write for
// old-batch duplicates) and the random-access latency of the CAS hides behind the walks of the other
// waves of the CU instead of being a kernel of its own.  Pass 2 (k_insert2) is unchanged.
// (Tried and dropped: loading the slot's claim word early, when the key is known but the extension block
// is still in flight, so that the CAS finds the line on-die — +0.6 ms at 100 M entries: the kernel is bound
// by memory transactions, not by the latency of the probe.)
template <int WCH, bool STRICT>
__global__ void __launch_bounds__(64) k_map_fused(MapArgs a, InsertArgs ia) {
  static_assert(WCH == 16, "cooperative fill assumes 16 chunks");
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint32_t lane = threadIdx.x;
  const uint64_t i = first + lane;
  const bool live = i < a.n;
  constexpr uint32_t STRIDE = WCH * 16 + 16;
  const uint64_t limit = map_limit(a);
  uint64_t lo = 0, hi = 0;
  if (live) cert_range(a.offsets, a.ends, i, lo, hi);
  const uint64_t g_me = live ? (lo & ~15ull) : ~0ull;
  {
    uint4 v[16];
    const uint32_t sub = lane & 15u;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint64_t g = __shfl(g_me, 4 * it + (int)(lane >> 4));
      const uint64_t at = g + 16u * sub;
      v[it] = (g != ~0ull && at + 16u <= limit) ? *(const uint4*)(a.payload + at) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 16; it++)
      *(uint4*)(smem + (4 * it + (lane >> 4)) * STRIDE + 16u * sub) = v[it];
  }
  __builtin_amdgcn_wave_barrier();
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  uint64_t claimed = ~0ull;
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
  if (live) {
    if constexpr (STRICT) {
      WinReaderS<WCH> r{{{(const uint32_t*)a.payload, lo, limit, (uint32_t*)(smem + lane * STRIDE),
                          (int32_t)(int64_t)(g_me - lo)}}, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, 0x80000000u};
      map_one(r, hi - lo, i, a, o0, o1);
      if (r.miss) {  // some access left the window: the exact reader decides (rare: hostile or odd layouts)
        GlobalReader g{(const uint32_t*)a.payload, lo};
        map_one(g, hi - lo, i, a, o0, o1);
      }
    } else {
      WinReaderC<WCH> r{{(const uint32_t*)a.payload, lo, limit, (uint32_t*)(smem + lane * STRIDE),
                         (int32_t)(int64_t)(g_me - lo)}};
      map_one(r, hi - lo, i, a, o0, o1);
    }
    const uint32_t status = o0.x & 0xffu;
    uint32_t state = ES_NONE, canon = 0;
    if (status == CTMR_ST_PASS) {
      canon = ia.canon[o0.z];
      state = insert_probe(ia, i, o0, o1, canon, claimed, q0, q1, q2, q3);
      if (state != ES_CLAIMED && state != ES_DEFER) o0.x &= ~((uint32_t)CTMR_FL_WAS_UNKNOWN << 8);
    }
    ia.ent[i] = ent_pack(status, state, canon);
  }
  store_records_wave(a, first, live, o0, o1);
  __builtin_amdgcn_wave_barrier();
  store_slots_wave(ia.table, (uint4*)smem, lane, claimed, q0, q1, q2, q3);
}

// (Tried and dropped, session 4: a software-pipelined form — one wave walks 2 or 4 batches of 64 certificates and
// fetches the next batch's front windows into registers while walking the current one.  256 VGPRs → 8 waves per CU,
// and the first vector load inside the walk (the issuerCN filter words) waits on vmcnt for the prefetch issued just
// before it, so the overlap never materialises: 26.0 ms against 23.1 ms, profiles/r01/s4/sweep_pipe.txt.)
// Wave-aggregated add: one atomic per distinct key per wave (the "match-any" loop).
__device__ __forceinline__ void wave_agg_add(bool active, uint32_t key, unsigned long long* arr) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k = __shfl(key, leader);
    const unsigned long long same = __ballot(active && key == k) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&arr[k], (unsigned long long)__popcll(same));
    todo &= ~same;
  }
}

__device__ __forceinline__ bool pair_add(PairSlot* pairs, uint64_t pmask, unsigned long long key,
                                         long long delta) {
  uint64_t j = mixk(key) & pmask;
  for (uint64_t probes = 0; probes <= pmask; probes++) {
    unsigned long long k = ld_agent(&pairs[j].key);
    if (k == 0ull) {
      const unsigned long long old = atomicCAS(&pairs[j].key, 0ull, key);
      k = old == 0ull ? key : old;
    }
    if (k == key) {
      atomicAdd(&pairs[j].count, (unsigned long long)delta);
      return true;
    }
    j = (j + 1) & pmask;
  }
  return false;
}

struct ResolveArgs {
  const uint32_t* ent;
  unsigned long long* issuer_counts;  // per canonical issuer
  DevStats* stats;
  uint32_t* blk_new;  // NEW count per 1024-entry block
  uint64_t n;
};

// One streaming pass over ent[] (4 bytes per entry; no table or record access): per-issuer unique
// counts of the entries that WERE unknown (Σ_expDate SCARD, storage-statistics.go:44-53), status
// histogram, NEW count per 1024-entry block for the compaction.
// Persistent blocks: per-issuer counts are first accumulated in an LDS histogram (issuers
// below RES_LDS_ISSUERS) and flushed with ONE global atomic per non-empty bin per block —
// hundreds of thousands of device atomics on the few cache lines of the hot issuers serialise
// at the memory side otherwise.  Issuers beyond the LDS bins use wave-aggregated global atomics.
constexpr uint32_t RES_LDS_ISSUERS = 4096;

__global__ void __launch_bounds__(1024) k_resolve(ResolveArgs a, uint64_t nb) {
  __shared__ uint32_t hist[CTMR_ST__COUNT + 4];
  __shared__ uint32_t ih[RES_LDS_ISSUERS];
  __shared__ uint32_t blk_cnt;
  if (threadIdx.x < CTMR_ST__COUNT + 4) hist[threadIdx.x] = 0;
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024) ih[k] = 0;
  __syncthreads();
  for (uint64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    if (threadIdx.x == 0) blk_cnt = 0;
    __syncthreads();
    const uint64_t i = blk * 1024 + threadIdx.x;
    bool is_new = false, is_dup = false, is_host = false, is_full = false;
    uint32_t status = CTMR_ST__COUNT, canon = 0;
    if (i < a.n) {
      const uint32_t e = a.ent[i];
      status = e & 7u;
      const uint32_t st = ent_state(e);
      canon = e >> 8;
      is_new = (st == ES_CLAIMED) | (st == ES_DEFER);
      is_dup = st == ES_DUP;
      is_host = st == ES_HOST;
      is_full = st == ES_FULL;
    }
    if (is_new && canon < RES_LDS_ISSUERS) atomicAdd(&ih[canon], 1u);
    wave_agg_add(is_new && canon >= RES_LDS_ISSUERS, canon, a.issuer_counts);
    // (the per-(expDate, issuer) cardinalities are rebuilt lazily by k_build_pairs on the first
    //  SetCardinality/KeysToChan after a mutation — they are statistics, not hot-path state)
    const unsigned long long m_new = __ballot(is_new), m_dup = __ballot(is_dup),
                             m_host = __ballot(is_host), m_full = __ballot(is_full);
    if ((threadIdx.x & 63) == 0) {
      if (m_new) {
        atomicAdd(&blk_cnt, (uint32_t)__popcll(m_new));
        atomicAdd(&hist[CTMR_ST__COUNT], (uint32_t)__popcll(m_new));
      }
      if (m_dup) atomicAdd(&hist[CTMR_ST__COUNT + 1], (uint32_t)__popcll(m_dup));
      if (m_host) atomicAdd(&hist[CTMR_ST__COUNT + 2], (uint32_t)__popcll(m_host));
      if (m_full) atomicAdd(&hist[CTMR_ST__COUNT + 3], (uint32_t)__popcll(m_full));
    }
#pragma unroll
    for (uint32_t st = 0; st < CTMR_ST__COUNT; st++) {
      const unsigned long long m = __ballot(status == st);
      if ((threadIdx.x & 63) == 0 && m) atomicAdd(&hist[st], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x == 0) a.blk_new[blk] = blk_cnt;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024)
    if (ih[k]) atomicAdd(&a.issuer_counts[k], (unsigned long long)ih[k]);
  if (threadIdx.x < CTMR_ST__COUNT) {
    if (hist[threadIdx.x]) atomicAdd(&a.stats->by_status[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
  } else if (threadIdx.x == CTMR_ST__COUNT) {
    if (hist[CTMR_ST__COUNT]) atomicAdd(&a.stats->n_new, (unsigned long long)hist[CTMR_ST__COUNT]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 1) {
    if (hist[CTMR_ST__COUNT + 1]) atomicAdd(&a.stats->n_dup, (unsigned long long)hist[CTMR_ST__COUNT + 1]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 2) {
    if (hist[CTMR_ST__COUNT + 2]) atomicAdd(&a.stats->n_host, (unsigned long long)hist[CTMR_ST__COUNT + 2]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 3) {
    if (hist[CTMR_ST__COUNT + 3]) atomicAdd(&a.stats->n_full, (unsigned long long)hist[CTMR_ST__COUNT + 3]);
  }
}

// Stream compaction of the NEW entries, ascending: wave ballot + popcount prefix inside a
// 1024-entry block, block bases from the exclusive scan of blk_new.  The NEW predicate comes
// from ent[] (local reduce) or from the record flag (exchange mode, ent == nullptr).
__global__ void __launch_bounds__(1024) k_compact(const ctmr_record* records, const uint32_t* ent, uint64_t n,
                                                  const uint64_t* blk_base, uint64_t* new_idx) {
  __shared__ uint32_t wave_cnt[16];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bool is_new = false;
  if (i < n)
    is_new = ent ? ent_is_new(ent[i]) : (((const uint8_t*)(records + i))[1] & CTMR_FL_WAS_UNKNOWN) != 0;
  const unsigned long long m = __ballot(is_new);
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (is_new) {
    uint32_t before = 0;
    for (uint32_t k = 0; k < wv; k++) before += wave_cnt[k];
    const uint32_t rank = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    new_idx[blk_base[blockIdx.x] + rank] = i;
  }
}

// exclusive scan of blk_new (u32) into blk_base (u64): single workgroup, chunked
__global__ void __launch_bounds__(1024) k_scan_blocks(const uint32_t* blk_new, uint64_t nb,
                                                      uint64_t* blk_base) {
  __shared__ unsigned long long part[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += 1024) {
    const uint64_t i = base + threadIdx.x;
    const unsigned long long v = i < nb ? blk_new[i] : 0ull;
    part[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) blk_base[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ cross-GPU key exchange
// Global dedup over G GPUs: every key has one OWNER = hash(key) mod G.  A
// rank exports the keys of its PASS entries partitioned by owner (ascending log index inside each
// partition), the partitions are exchanged (RCCL send/recv), the owner inserts what it received
// — concatenated in sender-rank order, which IS global log order because shards are contiguous
// log-index ranges — and returns one "was unknown" byte per key.
struct KeyRec {  // 64 bytes
  unsigned long long meta;  // key_meta(exp_hour, canonical issuer, serial_len)
  unsigned long long s[5];  // serial octets
  uint32_t src;             // index of the entry in the sender's batch
  uint32_t owner;
  unsigned long long pad;
};
static_assert(sizeof(KeyRec) == 64, "KeyRec");

constexpr uint32_t KEY_NO_OWNER = 0xffu;
constexpr uint32_t SID_DEFER = 0x80000000u;  // slot_id bit (owner-side kernels): candidate slot, full compare in pass 2
constexpr uint32_t MAX_WORLD = 16;

__device__ __forceinline__ bool entry_key(const InsertArgs& a, uint64_t i, unsigned long long& meta,
                                          unsigned long long s[5]) {
  const uint4* rp = (const uint4*)(a.records + i);
  const uint4 r0 = rp[0];
  if ((r0.x & 0xffu) != CTMR_ST_PASS) return false;
  const uint32_t slen = r0.x >> 16;
  if (slen > CTMR_MAX_SERIAL) return false;  // host-side set, shard-local
  const uint4 r1 = rp[1];
  record_key(a, i, r0, r1, s);
  meta = key_meta((int32_t)r0.y, a.canon[r0.z], slen);
  return true;
}

__device__ __forceinline__ uint32_t key_owner(unsigned long long meta, const unsigned long long s[5],
                                              uint32_t world) {
  return (uint32_t)(mixk(key_hash(meta, s) ^ 0x5bd1e995u) % world);
}

// pass A: owner of every entry + per-(owner, 1024-entry block) counts (owner-major layout)
__global__ void __launch_bounds__(1024) k_key_count(InsertArgs a, uint32_t world, uint64_t nb,
                                                    uint8_t* owner_out, uint32_t* cnt) {
  __shared__ uint32_t c[MAX_WORLD];
  if (threadIdx.x < MAX_WORLD) c[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  uint32_t owner = KEY_NO_OWNER;
  if (i < a.n) {
    unsigned long long meta, s[5];
    if (entry_key(a, i, meta, s)) owner = key_owner(meta, s, world);
    owner_out[i] = (uint8_t)owner;
  }
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot(owner == w);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c[w], (uint32_t)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < world) cnt[(uint64_t)threadIdx.x * nb + blockIdx.x] = c[threadIdx.x];
}

// pass B: stable scatter into the owner partitions
__global__ void __launch_bounds__(1024) k_key_scatter(InsertArgs a, uint32_t world, uint64_t nb,
                                                      const uint8_t* owner_in, const uint64_t* base,
                                                      KeyRec* out) {
  __shared__ uint32_t wc[16][MAX_WORLD];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t owner = i < a.n ? owner_in[i] : KEY_NO_OWNER;
  uint32_t my_rank = 0;
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot(owner == w);
    if (lane == 0) wc[wv][w] = (uint32_t)__popcll(m);
    if (owner == w) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (owner != KEY_NO_OWNER) {
    uint32_t before = 0;
    for (uint32_t k = 0; k < wv; k++) before += wc[k][owner];
    unsigned long long meta, s[5];
    entry_key(a, i, meta, s);
    KeyRec* o = out + base[(uint64_t)owner * nb + blockIdx.x] + before + my_rank;
    uint4* q = (uint4*)o;
    q[0] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
    q[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
    q[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
    q[3] = make_uint4((uint32_t)i, owner, 0u, 0u);
  }
}

// Owner side, pass 1 / pass 2 / resolve on received key records (same protocol as k_insert…)
__global__ void __launch_bounds__(256) k_keys_insert(const KeyRec* keys, uint64_t n, Slot* table,
                                                     uint64_t mask, uint32_t epoch, uint32_t* slot_id) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KeyRec k = keys[i];
  const unsigned long long h = key_hash(k.meta, k.s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & mask;
  uint32_t sid = SID_FULL;
  for (uint64_t probes = 0; probes <= mask; probes++) {
    Slot* sl = table + j;
    const unsigned long long old = atomicCAS(&sl->w[0], 0ull, tagw | (uint32_t)i);
    if (old == 0ull) {
      sl->w[1] = k.meta;
      uint4* q = (uint4*)&sl->w[2];
      q[0] = make_uint4(epoch, 0u, (uint32_t)k.s[0], (uint32_t)(k.s[0] >> 32));
      q[1] = make_uint4((uint32_t)k.s[1], (uint32_t)(k.s[1] >> 32), (uint32_t)k.s[2], (uint32_t)(k.s[2] >> 32));
      q[2] = make_uint4((uint32_t)k.s[3], (uint32_t)(k.s[3] >> 32), (uint32_t)k.s[4], (uint32_t)(k.s[4] >> 32));
      sid = (uint32_t)j;
      break;
    }
    if ((old & 0xffffffff00000000ull) == tagw) {
      const uint32_t ep = (uint32_t)ld_agent(&sl->w[2]);
      if (ep != 0u && ep != epoch) {
        bool eq = sl->w[1] == k.meta;
#pragma unroll
        for (int q = 0; q < 5; q++) eq = eq && sl->w[3 + q] == k.s[q];
        if (eq) {
          sid = SID_DUP_OLD;
          break;
        }
      } else {
        sid = (uint32_t)j | SID_DEFER;
        break;
      }
    }
    j = (j + 1) & mask;
  }
  slot_id[i] = sid;
}

__global__ void __launch_bounds__(256) k_keys_insert2(const KeyRec* keys, uint64_t n, Slot* table,
                                                      uint64_t mask, uint32_t epoch, uint32_t* slot_id) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t sid = slot_id[i];
  if (sid >= SID_DUP_OLD || !(sid & SID_DEFER)) return;
  sid &= ~SID_DEFER;
  const KeyRec k = keys[i];
  Slot* sl = table + sid;
  bool eq = sl->w[1] == k.meta;
#pragma unroll
  for (int q = 0; q < 5; q++) eq = eq && sl->w[3 + q] == k.s[q];
  if (eq) {
    atomicMin(&sl->w[0], ((unsigned long long)key_tag(key_hash(k.meta, k.s)) << 32) | (uint32_t)i);
  } else {
    bool created;
    sid = table_upsert(table, mask, k.meta, k.s, (uint32_t)i, epoch, true, &created);
  }
  slot_id[i] = sid;
}

__global__ void __launch_bounds__(1024) k_keys_resolve(const KeyRec* keys, uint64_t n, uint64_t nb,
                                                       const Slot* table, uint32_t epoch,
                                                       const uint32_t* slot_id, uint8_t* flags,
                                                       unsigned long long* issuer_counts, DevStats* stats) {
  __shared__ uint32_t ih[RES_LDS_ISSUERS];
  __shared__ uint32_t cnt[2];
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024) ih[k] = 0;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const uint64_t i = blk * 1024 + threadIdx.x;
    bool is_new = false, is_full = false;
    uint32_t canon = 0;
    if (i < n) {
      const uint32_t sid = slot_id[i];
      if (sid == SID_FULL) {
        is_full = true;
      } else if (sid < SID_DUP_OLD) {
        const Slot* sl = table + sid;
        const unsigned long long w0 = sl->w[0], w1 = sl->w[1], w2 = sl->w[2];
        is_new = (uint32_t)w2 == epoch && (uint32_t)w0 == (uint32_t)i;
        canon = (uint32_t)(w1 >> 32) & 0xffffffu;
      }
      flags[i] = is_new ? 1 : 0;
    }
    if (is_new && canon < RES_LDS_ISSUERS) atomicAdd(&ih[canon], 1u);
    wave_agg_add(is_new && canon >= RES_LDS_ISSUERS, canon, issuer_counts);
    const unsigned long long m_new = __ballot(is_new), m_full = __ballot(is_full);
    if ((threadIdx.x & 63) == 0) {
      if (m_new) atomicAdd(&cnt[0], (uint32_t)__popcll(m_new));
      if (m_full) atomicAdd(&cnt[1], (uint32_t)__popcll(m_full));
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024)
    if (ih[k]) atomicAdd(&issuer_counts[k], (unsigned long long)ih[k]);
  if (threadIdx.x == 0 && cnt[0]) atomicAdd(&stats->n_new, (unsigned long long)cnt[0]);
  if (threadIdx.x == 1 && cnt[1]) atomicAdd(&stats->n_full, (unsigned long long)cnt[1]);
}

// Sender side: apply the returned flags to the local records, count NEW per 1024-entry block
__global__ void __launch_bounds__(256) k_apply_flags(const KeyRec* sent, const uint8_t* flags, uint64_t n_keys,
                                                     ctmr_record* records, uint32_t* blk_new) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool is_new = k < n_keys && flags[k] != 0;
  uint32_t src = 0;
  if (is_new) {
    src = sent[k].src;
    uint8_t* fl = (uint8_t*)(records + src) + 1;
    *fl = (uint8_t)(*fl | CTMR_FL_WAS_UNKNOWN);
  }
  // per-1024-entry NEW counts for the compaction: keys of one partition are in ascending log order, so the lanes
  // of a wave nearly always share one counter — one atomic per distinct counter per wave (the per-lane form spent
  // 7.6 ms per 47 M keys serialising on single words)
  unsigned long long todo = __ballot(is_new);
  const uint32_t blk = src >> 10;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t b = __shfl(blk, leader);
    const unsigned long long same = __ballot(is_new && blk == b) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&blk_new[b], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

__global__ void __launch_bounds__(1024) k_status_hist(const ctmr_record* records, uint64_t n, uint64_t nb,
                                                      DevStats* stats) {
  __shared__ uint32_t hist[CTMR_ST__COUNT + 1];
  if (threadIdx.x <= CTMR_ST__COUNT) hist[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const uint64_t i = blk * 1024 + threadIdx.x;
    uint32_t status = CTMR_ST__COUNT, longs = 0;
    if (i < n) {
      const uint32_t head = *(const uint32_t*)(records + i);
      status = head & 0xffu;
      longs = status == CTMR_ST_PASS && (head >> 16) > CTMR_MAX_SERIAL;
    }
#pragma unroll
    for (uint32_t st = 0; st < CTMR_ST__COUNT; st++) {
      const unsigned long long m = __ballot(status == st);
      if ((threadIdx.x & 63) == 0 && m) atomicAdd(&hist[st], (uint32_t)__popcll(m));
    }
    const unsigned long long ml = __ballot(longs != 0);
    if ((threadIdx.x & 63) == 0 && ml) atomicAdd(&hist[CTMR_ST__COUNT], (uint32_t)__popcll(ml));
  }
  __syncthreads();
  if (threadIdx.x < CTMR_ST__COUNT && hist[threadIdx.x])
    atomicAdd(&stats->by_status[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
  if (threadIdx.x == CTMR_ST__COUNT && hist[CTMR_ST__COUNT])
    atomicAdd(&stats->n_host, (unsigned long long)hist[CTMR_ST__COUNT]);
}

// ------------------------------------------------------------------ cross-GPU dedup, Bloom pre-filter variant
// The all-gather of per-GPU Bloom fingerprints `north_star` names, made exact.  Every rank keeps
// its OWN known-certificate table (the ordinary fused map + insert runs unchanged) and a cumulative Bloom filter of
// every key it ever found locally new.  Per round: the filters are all-gathered; a rank probes its locally-new keys
// against the other ranks' filters — a Bloom filter has no false negatives, so a key that hits no peer filter exists
// on no other rank and needs no exchange at all; a key that hits peer p's filter (a real cross-rank duplicate or a
// false positive) is sent to p, which looks it up EXACTLY in its table and answers "known here before you": found
// with an older epoch, or found in this round under a lower global order (= lower log index).  Exactly one rank — the
// lowest log index — keeps WasUnknown for each key.  The asker then clears the flag, takes the key out of its
// per-issuer count and marks its slot SHADOW (known for dedup, not counted or listed: the sets of the ranks stay
// disjoint, so Σ over ranks of SCARD / per-issuer counts is the global value, as in the owner-computes variant).
//
// Filter: blocked Bloom, one 64-bit word per key, 4 bits inside it — one 8-byte atomicOr to add, one 8-byte load per
// peer to probe.  At 16 filter bits per key the false-positive rate is ≈ 0.5 % (only extra key traffic, never a wrong
// answer).
constexpr unsigned long long SLOT_SHADOW = 1ull << 63;  // Slot.w[2]: key is counted by another rank

__host__ __device__ inline void bloom_pos(unsigned long long h, uint64_t wmask, uint64_t& word,
                                          unsigned long long& bits) {
  const unsigned long long g = mixk(h ^ 0xa0761d6478bd642full);
  word = g & wmask;
  bits = (1ull << ((g >> 40) & 63)) | (1ull << ((g >> 46) & 63)) | (1ull << ((g >> 52) & 63)) |
         (1ull << ((g >> 58) & 63));
}

// key of entry i when it is a locally-new member of the device set (long serials stay shard-local on the host)
__device__ __forceinline__ bool entry_new_key(const InsertArgs& a, uint64_t i, unsigned long long& meta,
                                              unsigned long long s[5]) {
  const uint32_t head = *(const uint32_t*)(a.records + i);
  if ((head & 0xffu) != CTMR_ST_PASS || !((head >> 8) & CTMR_FL_WAS_UNKNOWN)) return false;
  return entry_key(a, i, meta, s);
}

__global__ void __launch_bounds__(256) k_bloom_add(InsertArgs a, unsigned long long* words, uint64_t wmask) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  unsigned long long meta, s[5];
  if (!entry_new_key(a, i, meta, s)) return;
  uint64_t word;
  unsigned long long bits;
  bloom_pos(key_hash(meta, s), wmask, word, bits);
  if ((ld_agent(&words[word]) & bits) != bits) atomicOr(&words[word], bits);
}

// pass A: peers whose filter holds the key (bit p of hit_out[i]) + per-(peer, 1024-entry block) counts
__global__ void __launch_bounds__(1024) k_bloom_probe(InsertArgs a, const unsigned long long* filters,
                                                      uint64_t n_words, uint32_t world, uint32_t rank, uint64_t nb,
                                                      uint16_t* hit_out, uint32_t* cnt) {
  __shared__ uint32_t c[MAX_WORLD];
  if (threadIdx.x < MAX_WORLD) c[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  uint32_t hit = 0;
  if (i < a.n) {
    unsigned long long meta, s[5];
    if (entry_new_key(a, i, meta, s)) {
      uint64_t word;
      unsigned long long bits;
      bloom_pos(key_hash(meta, s), n_words - 1, word, bits);
      for (uint32_t p = 0; p < world; p++)
        if (p != rank && (filters[(uint64_t)p * n_words + word] & bits) == bits) hit |= 1u << p;
    }
    hit_out[i] = (uint16_t)hit;
  }
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot((hit >> w) & 1u);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c[w], (uint32_t)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < world) cnt[(uint64_t)threadIdx.x * nb + blockIdx.x] = c[threadIdx.x];
}

// pass B: stable scatter of the key records into the per-peer partitions (a key goes to every peer it hit);
// KeyRec.owner = destination, KeyRec.pad = global order of the entry (order_base + batch index)
__global__ void __launch_bounds__(1024) k_bloom_scatter(InsertArgs a, uint32_t world, uint64_t nb,
                                                        const uint16_t* hit_in, const uint64_t* base,
                                                        unsigned long long order_base, KeyRec* out) {
  __shared__ uint32_t wc[16][MAX_WORLD];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t hit = i < a.n ? hit_in[i] : 0u;
  uint32_t my_rank[MAX_WORLD];
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot((hit >> w) & 1u);
    if (lane == 0) wc[wv][w] = (uint32_t)__popcll(m);
    my_rank[w] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (hit) {
    unsigned long long meta, s[5];
    entry_key(a, i, meta, s);
    const unsigned long long order = order_base + i;
    for (uint32_t w = 0; w < world; w++) {
      if (!((hit >> w) & 1u)) continue;
      uint32_t before = 0;
      for (uint32_t k = 0; k < wv; k++) before += wc[k][w];
      uint4* q = (uint4*)(out + base[(uint64_t)w * nb + blockIdx.x] + before + my_rank[w]);
      q[0] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
      q[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
      q[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
      q[3] = make_uint4((uint32_t)i, w, (uint32_t)order, (uint32_t)(order >> 32));
    }
  }
}

// read-only find (the batch that filled the table has completed: plain loads)
__device__ __forceinline__ uint32_t table_find(const Slot* table, uint64_t mask, unsigned long long meta,
                                               const unsigned long long s[5]) {
  const unsigned long long h = key_hash(meta, s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & mask;
  for (uint64_t probes = 0; probes <= mask; probes++) {
    const Slot* sl = table + j;
    const unsigned long long w0 = sl->w[0];
    if (w0 == 0ull) return SID_NONE;
    if ((w0 & 0xffffffff00000000ull) == tagw && w0 != SLOT_TOMB) {
      bool eq = sl->w[1] == meta;
#pragma unroll
      for (int k = 0; k < 5; k++) eq = eq && sl->w[3 + k] == s[k];
      if (eq) return (uint32_t)j;
    }
    j = (j + 1) & mask;
  }
  return SID_NONE;
}

// Peer side: flags[k] = 1 when the key is known here before the asker's entry — since an earlier round, or since
// this round under a lower global order.
__global__ void __launch_bounds__(256) k_keys_lookup(const KeyRec* keys, uint64_t n, const Slot* table, uint64_t mask,
                                                     uint32_t round_epoch, unsigned long long order_base,
                                                     uint8_t* flags) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KeyRec k = keys[i];
  const uint32_t sid = table_find(table, mask, k.meta, k.s);
  uint8_t f = 0;
  if (sid != SID_NONE) {
    const unsigned long long w0 = table[sid].w[0], w2 = table[sid].w[2];
    f = (uint32_t)w2 != round_epoch || order_base + (uint32_t)w0 < k.pad;
  }
  flags[i] = f;
}

// Asker side: a flagged key loses WasUnknown (once, however many peers flagged it), leaves the per-issuer count and
// its slot becomes SHADOW.
__global__ void __launch_bounds__(256) k_bloom_apply(const KeyRec* sent, const uint8_t* flags, uint64_t n_keys,
                                                     ctmr_record* records, Slot* table, uint64_t mask,
                                                     unsigned long long* issuer_counts) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool lost = false;
  uint32_t canon = 0;
  if (k < n_keys && flags[k] != 0) {
    const KeyRec kr = sent[k];
    const uint32_t old = atomicAnd((uint32_t*)(records + kr.src), ~((uint32_t)CTMR_FL_WAS_UNKNOWN << 8));
    if ((old >> 8) & CTMR_FL_WAS_UNKNOWN) {
      lost = true;
      canon = (uint32_t)(kr.meta >> 32) & 0xffffffu;
      const uint32_t sid = table_find(table, mask, kr.meta, kr.s);
      if (sid != SID_NONE) atomicOr(&table[sid].w[2], SLOT_SHADOW);
    }
  }
  unsigned long long todo = __ballot(lost);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(canon, leader);
    const unsigned long long same = __ballot(lost && canon == c) & todo;
    if ((int)(threadIdx.x & 63) == leader)
      atomicAdd(&issuer_counts[c], (unsigned long long)(-(long long)__popcll(same)));
    todo &= ~same;
  }
}

// NEW count per 1024-entry block from the record flags (compaction after k_bloom_apply)
__global__ void __launch_bounds__(1024) k_count_new_flags(const ctmr_record* records, uint64_t n, uint32_t* blk_new) {
  __shared__ uint32_t c;
  if (threadIdx.x == 0) c = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const bool is_new = i < n && (((const uint8_t*)(records + i))[1] & CTMR_FL_WAS_UNKNOWN) != 0;
  const unsigned long long m = __ballot(is_new);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) blk_new[blockIdx.x] = c;
}

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

// ------------------------------------------------------------------ CT get-entries decode
struct __attribute__((packed, aligned(1))) U4 { uint32_t a; };
struct DevBytes {  // arbitrary byte positions of the blob (gfx950 runs with unaligned access mode)
  const uint8_t* p;
  __device__ __forceinline__ uint32_t le32(uint64_t pos) const { return ((const U4*)(p + pos))->a; }
  __device__ __forceinline__ void le128(uint64_t pos, uint32_t out[4]) const {
    const U16 v = *(const U16*)(p + pos);
    out[0] = v.a; out[1] = v.b; out[2] = v.c; out[3] = v.d;
  }
  __device__ __forceinline__ uint32_t u8(uint64_t pos) const { return p[pos]; }
  __device__ __forceinline__ uint32_t be(uint64_t pos, int k) const {  // reads ≤ 3 bytes past pos+k: CTMR_PAYLOAD_PAD
    return __builtin_bswap32(le32(pos)) >> (32 - 8 * k);
  }
};

struct DecodeArgs {
  const uint8_t* blob;
  const uint64_t* bounds;  // 2n+1
  uint64_t n;
  uint64_t* cert_start;
  uint64_t* cert_end;
  uint8_t* entry_type;
  uint64_t* timestamp;     // may be null
  uint64_t* chain0_start;
  uint32_t* chain0_len;
  unsigned long long* counters;  // [0] x509 [1] precert [2] decode error [3] len(Chain) < 1
};

// ct.LogEntryFromLeaf, one raw entry per lane (entry_decode.h).  Reads ≈ 5 scattered header words per entry
// (leaf header, extensions length behind the certificate, the chain headers); the certificates themselves
// are skipped by length.
constexpr uint32_t DECODE_PER_BLOCK = 2048;  // entries per workgroup: counters reach global memory once per 2048 entries
__global__ void __launch_bounds__(256) k_entry_decode(DecodeArgs a) {
  __shared__ uint32_t cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const uint64_t base = (uint64_t)blockIdx.x * DECODE_PER_BLOCK;
  DevBytes b{a.blob};
#pragma unroll 2
  for (uint32_t k = 0; k < DECODE_PER_BLOCK / 256; k++) {
    const uint64_t i = base + k * 256u + threadIdx.x;
    if (i >= a.n) break;
    EntryDec d;
    decode_entry(b, a.bounds[2 * i], a.bounds[2 * i + 1], a.bounds[2 * i + 2], d);
    a.cert_start[i] = d.ok ? d.cert_lo : 0ull;
    a.cert_end[i] = d.ok ? d.cert_hi : 0ull;
    a.entry_type[i] = d.ok ? (uint8_t)d.entry_type : (uint8_t)CTMR_ENTRY_INVALID;
    if (a.timestamp) a.timestamp[i] = d.ok ? d.timestamp : 0ull;
    a.chain0_start[i] = d.ok ? d.chain0_lo : 0ull;
    a.chain0_len[i] = d.ok ? d.chain0_len : 0u;
    c0 += d.ok && d.entry_type == 0;
    c1 += d.ok && d.entry_type == 1;
    c2 += !d.ok;
    c3 += d.ok && d.n_chain == 0;
  }
  // hundreds of thousands of device atomics on one cache line serialise at the memory side (measured: 12 of the
  // 15 ms of the first version of this kernel at 40 M entries): LDS first, then four atomics per workgroup
  if (c0) atomicAdd(&cnt[0], c0);
  if (c1) atomicAdd(&cnt[1], c1);
  if (c2) atomicAdd(&cnt[2], c2);
  if (c3) atomicAdd(&cnt[3], c3);
  __syncthreads();
  if (threadIdx.x < 4 && cnt[threadIdx.x]) atomicAdd(&a.counters[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// Chain[0] → issuer table index: replaces, per entry, x509.ParseCertificate(Chain[0]) + NewIssuer
// (ct-fetch.go:221; storage/types.go:109-115) by a bytewise match against the issuer certificates registered so
// far (each of which went through exactly that parse once, k_issuer_ids).  Phase 1, per lane: candidate from a
// small hash table keyed by cert_quick_hash.  Phase 2, wave-cooperative: the 64 lanes stream the candidate's
// bytes (16 B per lane per step, 1 KiB per instruction) against the registered copy — every byte of Chain[0]
// is compared, so equal means identical.  Unregistered certificates are reported once per distinct hash
// (pend[] claims) for the host to register; `retry` re-examines only entries still marked unregistered.
constexpr uint32_t ISS_UNREGISTERED = 0xfffffffeu;
constexpr uint32_t PEND_SLOTS = 8192;  // distinct unknown Chain[0] hashes remembered per launch
#ifndef CTMR_MATCH_PER_STEP
#define CTMR_MATCH_PER_STEP 4
#endif
constexpr uint32_t MATCH_PER_STEP = CTMR_MATCH_PER_STEP;  // candidates whose loads are in flight together (4 and 8 both measure 9.2–9.3 ms per 40 M entries: the kernel is near the HBM rate once partial lines are counted)

struct MatchArgs {
  const uint8_t* blob;
  const uint64_t* chain0_start;
  const uint32_t* chain0_len;
  const uint8_t* entry_type;
  uint32_t* issuer_idx;
  uint64_t n;
  // issuer certificate store
  const uint8_t* idb_der;       // registered certificates, each at a 16-byte aligned offset, zero padded
  const uint64_t* idb_off;      // per issuer: offset into idb_der
  const uint32_t* idb_len;
  const unsigned long long* ht; // open addressing: (candidate hash & ~0xffffffff) | (issuer index + 1), 0 = empty
  uint32_t ht_mask;
  uint32_t retry;
  // unregistered report
  unsigned long long* pend;     // PEND_SLOTS claim words (zeroed by the host)
  uint32_t* unreg_list;         // entry indices, one per distinct hash
  uint32_t unreg_cap;
  unsigned long long* counters; // [0] entries left unregistered, [1] list entries, [2] pend overflow
};

__device__ __forceinline__ bool eq16_prefix(const U16& x, const uint4& y, uint32_t rem) {  // first min(rem,16) bytes equal
  const uint32_t d[4] = {x.a ^ y.x, x.b ^ y.y, x.c ^ y.z, x.d ^ y.w};
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t have = rem > 4u * k ? rem - 4u * k : 0u;
    const uint32_t mask = have >= 4u ? 0xffffffffu : (have ? (0xffffffffu >> (8 * (4 - have))) : 0u);
    eq = eq && (d[k] & mask) == 0u;
  }
  return eq;
}

__global__ void __launch_bounds__(256) k_chain0_match(MatchArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < a.n;
  uint64_t lo = 0;
  uint32_t len = 0;
  bool need = false;
  uint32_t result = CTMR_NO_ISSUER;
  if (live) {
    if (a.retry) {
      result = a.issuer_idx[i];
      need = result == ISS_UNREGISTERED;
    } else {
      need = a.entry_type[i] != CTMR_ENTRY_INVALID;
    }
    if (need) {
      lo = a.chain0_start[i];
      len = a.chain0_len[i];
      need = len != 0u;
      if (!need) result = CTMR_NO_ISSUER;
    }
  }
  DevBytes b{a.blob};
  unsigned long long qh = 0;
  uint32_t j = 0;
  if (need) {
    qh = cert_quick_hash(b, lo, len);
    j = (uint32_t)qh & a.ht_mask;
    result = ISS_UNREGISTERED;
  }
  bool searching = need;
  while (__ballot(searching)) {
    // next candidate of every searching lane: one table word carries the issuer index and the upper half of its
    // hash; length and store offset come with one more (parallel) pair of loads — no dependent load is left for
    // the cooperative phase
    uint32_t cand = 0xffffffffu;
    uint64_t db_off = 0;
    if (searching) {
      for (;;) {
        const unsigned long long v = a.ht[j];
        if (v == 0ull) {
          searching = false;
          break;
        }
        j = (j + 1u) & a.ht_mask;
        if ((v ^ qh) >> 32 == 0ull) {
          const uint32_t c = (uint32_t)v - 1u;
          const uint32_t clen = a.idb_len[c];
          db_off = a.idb_off[c];
          if (clen == len) {
            cand = c;
            break;
          }
        }
      }
    }
    // cooperative bytewise verification, MATCH_PER_STEP candidates per step so that their loads are in flight together
    // (a step costs one memory latency; certificates up to 2 KiB need no inner loop)
    unsigned long long todo = __ballot(cand != 0xffffffffu);
    while (todo) {
      int src[MATCH_PER_STEP];
      bool eq[MATCH_PER_STEP];
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        src[u] = todo ? __ffsll((long long)todo) - 1 : -1;
        todo &= todo - 1ull;  // 0 stays 0
        eq[u] = true;
      }
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        if (src[u] < 0) continue;  // wave-uniform
        const uint64_t s_lo = __shfl(lo, src[u]);
        const uint32_t s_len = __shfl(len, src[u]);
        const uint8_t* db = a.idb_der + __shfl(db_off, src[u]);
        const uint32_t off0 = lane * 16u, off1 = off0 + 1024u;
        if (off0 < s_len) {
          const U16 x = *(const U16*)(a.blob + s_lo + off0);  // ≤ 15 bytes past Chain[0]: CTMR_PAYLOAD_PAD
          const uint4 y = *(const uint4*)(db + off0);
          eq[u] = eq16_prefix(x, y, s_len - off0);
        }
        if (off1 < s_len) {
          const U16 x = *(const U16*)(a.blob + s_lo + off1);
          const uint4 y = *(const uint4*)(db + off1);
          eq[u] = eq[u] && eq16_prefix(x, y, s_len - off1);
        }
        for (uint32_t off = off0 + 2048u; off < s_len; off += 1024u) {  // > 2 KiB: rare
          const U16 x = *(const U16*)(a.blob + s_lo + off);
          const uint4 y = *(const uint4*)(db + off);
          eq[u] = eq[u] && eq16_prefix(x, y, s_len - off);
        }
      }
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        if (src[u] < 0) continue;
        const bool all = __ballot(!eq[u]) == 0ull;
        if ((int)lane == src[u] && all) {
          result = cand;
          searching = false;
        }
      }
    }
  }
  if (live) a.issuer_idx[i] = result;
  // report unregistered certificates, once per distinct hash
  const bool unreg = live && result == ISS_UNREGISTERED;
  const unsigned long long mu = __ballot(unreg);
  if (lane == 0 && mu) atomicAdd(&a.counters[0], (unsigned long long)__popcll(mu));
  // one claim per distinct hash per wave (a cold start has every lane here)
  unsigned long long todo_u = mu;
  while (todo_u) {
    const int leader = __ffsll((long long)todo_u) - 1;
    const unsigned long long lq = __shfl(qh, leader);
    const unsigned long long same = __ballot(unreg && qh == lq) & todo_u;
    todo_u &= ~same;
    if ((int)lane != leader) continue;
    uint32_t k = (uint32_t)(qh >> 32) & (PEND_SLOTS - 1u);
    bool first = false, placed = false;
    for (uint32_t probes = 0; probes < 64u && !placed; probes++) {
      const unsigned long long old = atomicCAS(&a.pend[k], 0ull, qh);
      if (old == 0ull) {
        first = true;
        placed = true;
      } else if (old == qh) {
        placed = true;
      }
      k = (k + 1u) & (PEND_SLOTS - 1u);
    }
    if (!placed) {
      atomicAdd(&a.counters[2], 1ull);
      first = true;  // overflow: report it anyway (the host dedups by bytes)
    }
    if (first) {
      const unsigned long long at = atomicAdd(&a.counters[1], 1ull);
      if (at < a.unreg_cap) a.unreg_list[at] = (uint32_t)i;
    }
  }
}

// ------------------------------------------------------------------ IssuerMetadata on device
// IssuerMetadata.Accumulate (storage/issuermetadata.go:92-138) runs for every newly unknown certificate but changes
// state only the first time an issuer meets an (expDate), a CRL distribution point or an issuer DN: its three
// per-issuer memo maps (knownExpDates :96, knownCrlDPs :113, knownIssuerDNs :97) live here as ONE device hash set of
// (kind, issuer, bytes).  k_meta_new walks the NEW list of a batch, and appends an item only for first sightings —
// the host then formats/inserts those few (addCRL :48-73, addIssuerDN :75-87, AllocateExpDateAndIssuer
// filesystemdatabase.go:189-195) instead of parsing every new certificate.
// Set semantics are exact: a slot is claimed by CAS on the 64-bit hash, its bytes are copied into an arena and
// published (write-through payload, drained, then the VALID word — the table_upsert recipe); equal hash is
// followed by a full comparison, so a hash collision only costs a probe.
struct MetaSlot {
  unsigned long long w[4];  // w0 hash (claim, never 0) | w1 VALID(63) kind(61..60) len(59..40) arena_off/8(39..0)
};                          // w2 issuer << 32 | key2 | w3 launch number that created the slot
constexpr unsigned long long META_VALID = 1ull << 63;
constexpr uint32_t MK_EXPDATE = 0, MK_CRL = 1, MK_DN = 2, MK_HOST = 3;  // item kinds; MK_HOST = parse this one on the host
constexpr uint32_t META_MAX_BYTES = 4096;
constexpr uint32_t META_MAX_URIS = 4;  // CRL distribution point URIs per certificate on the device path; more → host
struct ByteReader {  // one unaligned dword per access (k_meta_new's TLV reads)
  const uint8_t* p;
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const { return ((const U4*)(p + pos))->a; }
};

struct MetaItem {  // 32 bytes, = ctmr_meta_item
  uint64_t entry;
  uint32_t kind, issuer_idx;
  int32_t exp_hour;
  uint32_t off, len, pad;
};
static_assert(sizeof(MetaItem) == 32, "MetaItem");

struct MetaArgs {
  const uint8_t* payload;
  const uint64_t* offsets;
  const uint64_t* ends;
  const ctmr_record* records;
  const uint32_t* canon;
  const uint2* meta_loc;
  const uint64_t* new_idx;
  uint64_t n_new;
  MetaSlot* slots;
  uint64_t mask;
  uint8_t* arena;
  uint64_t arena_cap;
  unsigned long long* counters;  // [0] arena bytes used [1] items appended [2] set/arena overflow events
  MetaItem* items;
  uint64_t items_cap;
  uint32_t epoch;  // launch number (≥ 1): slots of earlier launches are immutable and read through the caches
};

// 16-byte chunks of an item, bytes past its end zeroed.  Items are hashed and compared in these chunks: the first
// version of this kernel used dwords and was bound by L2 REQUESTS (1.46 G for 18.8 M new certificates, 76 % of its
// L1 accesses missing — profiles/r01/s4/pmc_meta_20m_dword_version.txt).
__device__ __forceinline__ uint4 mask_chunk(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t rem) {
  uint32_t w[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t have = rem > 4u * q ? rem - 4u * q : 0u;
    w[q] = have >= 4u ? w[q] : (have ? (w[q] & (0xffffffffu >> (8u * (4u - have)))) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
struct GlobalSrc {  // straight from the certificate in HBM: one unaligned dwordx4 load per chunk (≤ 15 bytes past the
  const uint8_t* p; //  item, which lies inside a certificate inside the payload + CTMR_PAYLOAD_PAD)
  uint32_t len;
  __device__ __forceinline__ uint4 chunk(uint32_t k) const {
    const U16 v = *(const U16*)(p + 16u * k);
    return mask_chunk(v.a, v.b, v.c, v.d, len - 16u * k);
  }
};
struct LdsSrc {  // from this lane's staging area in LDS, at any byte offset (5 dwords, 4 alignbytes)
  const uint32_t* w;  // dword-aligned lane area
  uint32_t off;       // byte offset of the item inside it
  uint32_t len;
  __device__ __forceinline__ uint4 chunk(uint32_t k) const {
    const uint32_t at = off + 16u * k, i = at >> 2, sh = at & 3u;
    const uint32_t d0 = w[i], d1 = w[i + 1], d2 = w[i + 2], d3 = w[i + 3], d4 = w[i + 4];
    return mask_chunk(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                      __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh), len - 16u * k);
  }
};

// Plain (cacheable) load the compiler may not merge or hoist: wavefront-scope atomic.  Used for memo slots of EARLIER
// launches, which are immutable — the steady state, where the same few hundred DN/CRL slots are read by every new
// certificate and should come out of L1/L2 instead of device-coherent loads.
__device__ __forceinline__ unsigned long long ld_wave(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// true = first sighting of (kind, issuer, key2, bytes); the bytes come through a chunk source
template <class S>
__device__ __forceinline__ bool meta_upsert(const MetaArgs& a, uint32_t kind, uint32_t issuer, uint32_t key2,
                                            const S& src, uint32_t len) {
  const uint32_t nc = (len + 15u) >> 4;
  unsigned long long h = mixk(((unsigned long long)issuer << 32 | key2) + 0x9e3779b97f4a7c15ull * (kind + 1u));
  h = mixk(h ^ len);
  for (uint32_t k = 0; k < nc; k++) {
    const uint4 c = src.chunk(k);
    h = mixk(h ^ (((unsigned long long)c.y << 32 | c.x) + 0x9e3779b97f4a7c15ull * (2u * k + 2u)));
    h = mixk(h ^ (((unsigned long long)c.w << 32 | c.z) + 0x9e3779b97f4a7c15ull * (2u * k + 3u)));
  }
  if (h == 0ull) h = 1ull;
  const unsigned long long w2 = ((unsigned long long)issuer << 32) | key2;
  uint64_t j = h & a.mask;
  uint64_t probes = 0;
  for (;;) {
    MetaSlot* sl = a.slots + j;
    unsigned long long w0 = ld_wave(&sl->w[0]);  // a stale 0 only sends us to the CAS, which tells the truth
    if (w0 == 0ull) {
      const unsigned long long old = atomicCAS(&sl->w[0], 0ull, h);
      if (old == 0ull) {  // claimed: copy the bytes, publish
        const unsigned long long need = (unsigned long long)nc * 16ull;
        unsigned long long at = need ? atomicAdd(&a.counters[0], need) : 0ull;
        uint32_t pk = kind;
        if (at + need > a.arena_cap) {  // arena exhausted: a dead slot (never equal to anything); always "new"
          atomicAdd(&a.counters[2], 1ull);
          pk = MK_HOST;
          at = 0;
        } else {
          unsigned long long* dst = (unsigned long long*)(a.arena + at);
          for (uint32_t k = 0; k < nc; k++) {
            const uint4 c = src.chunk(k);
            st_agent(dst + 2 * k, (unsigned long long)c.y << 32 | c.x);
            st_agent(dst + 2 * k + 1, (unsigned long long)c.w << 32 | c.z);
          }
        }
        st_agent(&sl->w[2], w2);
        st_agent(&sl->w[3], (unsigned long long)a.epoch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(&sl->w[1], META_VALID | ((unsigned long long)pk << 60) | ((unsigned long long)len << 40) | (at >> 3));
        return true;
      }
      w0 = old;
    }
    if (w0 == h) {
      unsigned long long m = ld_wave(&sl->w[1]);
      const unsigned long long ep = ld_wave(&sl->w[3]);
      const bool settled = (m & META_VALID) && ep != 0ull && ep < a.epoch;  // published by an earlier launch: immutable
      if (!settled) {
        m = ld_agent(&sl->w[1]);
        if (!(m & META_VALID)) continue;  // the claimer has not published yet: poll again (as table_upsert does)
      }
      bool eq = ((m >> 60) & 3ull) == kind && ((m >> 40) & 0xfffffull) == len &&
                (settled ? ld_wave(&sl->w[2]) : ld_agent(&sl->w[2])) == w2;
      if (eq) {
        const unsigned long long* asrc = (const unsigned long long*)(a.arena + ((m & 0xffffffffffull) << 3));
        for (uint32_t k = 0; (k < nc) & eq; k++) {
          const uint4 c = src.chunk(k);
          unsigned long long s0, s1;
          if (settled) {
            const uint4 v = *(const uint4*)(asrc + 2 * k);  // immutable: plain 16-byte load
            s0 = (unsigned long long)v.y << 32 | v.x;
            s1 = (unsigned long long)v.w << 32 | v.z;
          } else {
            s0 = ld_agent(asrc + 2 * k);
            s1 = ld_agent(asrc + 2 * k + 1);
          }
          eq = s0 == ((unsigned long long)c.y << 32 | c.x) && s1 == ((unsigned long long)c.w << 32 | c.z);
        }
      }
      if (eq) return false;
    }
    j = (j + 1) & a.mask;
    if (++probes > a.mask) break;
  }
  atomicAdd(&a.counters[2], 1ull);  // set full: report every time (the host's sets dedup)
  return true;
}

__device__ __forceinline__ void meta_emit(const MetaArgs& a, uint64_t entry, uint32_t kind, uint32_t issuer_idx,
                                          int32_t exp_hour, uint32_t off, uint32_t len) {
  const unsigned long long at = atomicAdd(&a.counters[1], 1ull);
  if (at < a.items_cap) a.items[at] = MetaItem{entry, kind, issuer_idx, exp_hour, off, len, 0u};
}

// Per-lane LDS staging: the issuer Name (≤ META_LDS_DN bytes) and the cRLDistributionPoints value (≤ META_LDS_CRL) of
// the lane's certificate are fetched with up to 12 independent 16-byte loads issued together — ONE memory latency —
// and everything after that (the DistributionPoint walk, hashing, comparing) reads LDS.  The dependent chain per
// certificate drops from ≈35 global round trips to the three memo probes.  Longer items take the global path.
constexpr uint32_t META_LDS_DN = 128, META_LDS_CRL = 64, META_LDS_STRIDE = META_LDS_DN + META_LDS_CRL + 16;

struct LdsTlvReader {  // rd_hdr over the staged cRLDistributionPoints value: positions are certificate offsets
  const uint32_t* w;   // lane area (dwords) of the value
  uint32_t s;          // certificate offset of its first byte
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - s;  // callers stay within [s, e + 3]; the area has 16 bytes of slack
    const uint32_t i = rel >> 2;
    return __builtin_amdgcn_alignbyte(w[i + 1], w[i], rel & 3u);
  }
};

// DistributionPoint walk (RFC 5280 §4.2.1.13) over [cs, e): collects up to META_MAX_URIS URI ranges (certificate
// offsets); returns false when the value is malformed.  `host` is set when a URI is too long or there are too many.
template <class R>
__device__ __forceinline__ bool walk_crl_dps(const R& g, uint32_t L, uint32_t cs, uint32_t e, uint32_t uo[META_MAX_URIS],
                                             uint32_t ul[META_MAX_URIS], uint32_t& nu, bool& host) {
  bool ok = true;
  uint32_t p = cs;
  while (ok && p < e) {
    uint32_t t1, f, f_end;
    rd_hdr(g, L, p, e, ok, t1, f, f_end);
    ok = ok && t1 == 0x30u;
    while (ok && f < f_end) {
      uint32_t t2, n, n_end;
      rd_hdr(g, L, f, f_end, ok, t2, n, n_end);
      if (ok && t2 == 0xa0u) {
        while (ok && n < n_end) {
          uint32_t t3, q, q_end;
          rd_hdr(g, L, n, n_end, ok, t3, q, q_end);
          if (ok && t3 == 0xa0u) {
            while (ok && q < q_end) {
              uint32_t t4, u, u_end;
              rd_hdr(g, L, q, q_end, ok, t4, u, u_end);
              if (ok && t4 == 0x86u) {
                if (u_end - u > META_MAX_BYTES || nu >= META_MAX_URIS) host = true;
#pragma unroll
                for (uint32_t k = 0; k < META_MAX_URIS; k++) {  // register array: no dynamic indexing
                  uo[k] = k == nu ? u : uo[k];
                  ul[k] = k == nu ? u_end - u : ul[k];
                }
                nu++;
              }
              q = u_end;
            }
          }
          n = q_end;
        }
      }
      f = n_end;
    }
    p = f_end;
  }
  return ok;
}

__global__ void __launch_bounds__(256) k_meta_new(MetaArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[256 * META_LDS_STRIDE];
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n_new) return;
  const uint64_t i = a.new_idx[r];
  const uint4 r0 = *(const uint4*)(a.records + i);
  const int32_t exp_hour = (int32_t)r0.y;
  const uint32_t iss = r0.z, canon = a.canon[iss];
  uint64_t lo, hi;
  cert_range(a.offsets, a.ends, i, lo, hi);
  const uint32_t L = (uint32_t)(hi - lo);
  const uint8_t* cert = a.payload + lo;
  const uint2 ml = a.meta_loc[i];
  bool host = ml.x == META_HOST || ml.y == META_HOST || ml.x == META_NONE;
  uint32_t dn_off = 0, dn_len = 0, cr_s = 0, cr_len = 0;
  if (!host) {
    dn_off = ml.x & 0xffffu;
    dn_len = ml.x >> 16;
    host = dn_len > META_MAX_BYTES || dn_off + dn_len > L;
    if (ml.y != META_NONE) {
      cr_s = ml.y & 0xffffu;
      cr_len = ml.y >> 16;
      host = host || cr_s + cr_len > L;
    }
  }
  // ---- stage: every load of this lane is in flight before the first one is needed
  uint8_t* my = stage + threadIdx.x * META_LDS_STRIDE;
  const bool dn_lds = !host && dn_len <= META_LDS_DN, cr_lds = !host && cr_len != 0u && cr_len <= META_LDS_CRL;
  {
    U16 d[META_LDS_DN / 16], c[META_LDS_CRL / 16];
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      if (dn_lds && 16u * k < dn_len) d[k] = *(const U16*)(cert + dn_off + 16u * k);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      if (cr_lds && 16u * k < cr_len) c[k] = *(const U16*)(cert + cr_s + 16u * k);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      if (dn_lds && 16u * k < dn_len) *(uint4*)(my + 16u * k) = make_uint4(d[k].a, d[k].b, d[k].c, d[k].d);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      if (cr_lds && 16u * k < cr_len) *(uint4*)(my + META_LDS_DN + 16u * k) = make_uint4(c[k].a, c[k].b, c[k].c, c[k].d);
  }
  // knownExpDates → seenExpDateBefore (issuermetadata.go:96-108): no bytes, probes while the loads fly
  if (meta_upsert(a, MK_EXPDATE, canon, (uint32_t)exp_hour, GlobalSrc{cert, 0}, 0)) meta_emit(a, i, MK_EXPDATE, iss, exp_hour, 0, 0);
  // knownCrlDPs (:111-127): CRLDistributionPoints ::= SEQUENCE OF DistributionPoint { [0] { [0] GeneralNames { [6] URI }}}
  // One validating pass collects the URI ranges (a malformed value yields NO URIs, as the oracle defines; more than
  // META_MAX_URIS → host), then the memo is consulted.
  if (!host && cr_len != 0u) {
    const uint32_t e = cr_s + cr_len;
    uint32_t uo[META_MAX_URIS], ul[META_MAX_URIS], nu = 0;
#pragma unroll
    for (uint32_t k = 0; k < META_MAX_URIS; k++) uo[k] = ul[k] = 0;
    bool ok = true;
    uint32_t tag, cs, ce;
    const uint32_t* cw = (const uint32_t*)(my + META_LDS_DN);
    if (cr_lds) {
      LdsTlvReader g{cw, cr_s};
      rd_hdr(g, e, cr_s, e, ok, tag, cs, ce);  // L = e: reads clamp to the staged value, not to the certificate
      ok = ok && tag == 0x30u && ce == e;
      ok = ok && walk_crl_dps(g, e, cs, e, uo, ul, nu, host);
    } else {
      ByteReader g{cert};
      rd_hdr(g, L, cr_s, e, ok, tag, cs, ce);
      ok = ok && tag == 0x30u && ce == e;
      ok = ok && walk_crl_dps(g, L, cs, e, uo, ul, nu, host);
    }
    if (ok && !host) {
#pragma unroll
      for (uint32_t k = 0; k < META_MAX_URIS; k++) {
        if (k >= nu) continue;
        const bool first = cr_lds ? meta_upsert(a, MK_CRL, canon, 0, LdsSrc{cw, uo[k] - cr_s, ul[k]}, ul[k])
                                  : meta_upsert(a, MK_CRL, canon, 0, GlobalSrc{cert + uo[k], ul[k]}, ul[k]);
        if (first) meta_emit(a, i, MK_CRL, iss, exp_hour, uo[k], ul[k]);
      }
    }
  }
  // knownIssuerDNs (:97,:130-135): keyed by the Name's DER bytes (Issuer.String() is a function of them)
  if (!host) {
    const bool first = dn_lds ? meta_upsert(a, MK_DN, canon, 0, LdsSrc{(const uint32_t*)my, 0u, dn_len}, dn_len)
                              : meta_upsert(a, MK_DN, canon, 0, GlobalSrc{cert + dn_off, dn_len}, dn_len);
    if (first) meta_emit(a, i, MK_DN, iss, exp_hour, dn_off, dn_len);
  }
  if (host) meta_emit(a, i, MK_HOST, iss, exp_hour, 0, L);
}

// ------------------------------------------------------------------ whole-certificate SHA-256 (auxiliary)
// NOT on the reference's path — it never hashes a leaf certificate (the only SHA-256 is Issuer.ID's,
// storage/types.go:155-159).  This is the kernel BASELINE.json's north_star names literally ("one-cert-per-lane
// SHA-256 with round constants in LDS"): the fingerprint CT tooling identifies certificates by.  VALU-bound
// (≈2 000 instructions per 64-byte block), not HBM-bound: reported against its own roofline.
// Full blocks are fetched as four unaligned 16-byte loads per lane; the padded tail goes through the byte path.
__global__ void __launch_bounds__(256) k_fingerprint(const uint8_t* payload, const uint64_t* offsets,
                                                     const uint64_t* ends, uint64_t n, uint32_t* digests) {
  __shared__ uint32_t kc[64];
  if (threadIdx.x < 64) kc[threadIdx.x] = K256[threadIdx.x];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t lo, hi;
  cert_range(offsets, ends, i, lo, hi);
  const uint64_t len64 = hi - lo;
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const uint8_t* p = payload + lo;
  const uint64_t full = len64 >> 6;
  for (uint64_t b = 0; b < full; b++) {
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const U16 v = *(const U16*)(p + b * 64 + q * 16);
      w[4 * q] = __builtin_bswap32(v.a); w[4 * q + 1] = __builtin_bswap32(v.b);
      w[4 * q + 2] = __builtin_bswap32(v.c); w[4 * q + 3] = __builtin_bswap32(v.d);
    }
    sha256_compress(h, w, kc);
  }
  // tail: 0..63 message bytes, 0x80, zeros, 64-bit bit length — one or two blocks
  const uint32_t rem = (uint32_t)(len64 & 63u);
  const uint32_t nt = rem + 9 > 64 ? 2u : 1u;
  const uint8_t* t = p + full * 64;
  for (uint32_t b = 0; b < nt; b++) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t pos = b * 64 + k * 4;
      uint32_t v = 0;
      if (pos + 4 <= rem) {
        v = __builtin_bswap32(((const U4*)(t + pos))->a);
      } else if (pos <= rem) {
        const uint32_t r = rem - pos;  // 0..3 message bytes in this word
        const uint32_t raw = r ? ((const U4*)(t + pos))->a : 0u;  // ≤ 3 bytes past the certificate: CTMR_PAYLOAD_PAD
        const uint32_t m = r ? (raw & (0xffffffffu >> (8 * (4 - r)))) : 0u;
        v = __builtin_bswap32(m | (0x80u << (8 * r)));
      }
      w[k] = v;
    }
    if (b == nt - 1) {
      w[14] = (uint32_t)((len64 * 8ull) >> 32);
      w[15] = (uint32_t)(len64 * 8ull);
    }
    sha256_compress(h, w, kc);
  }
  uint4* out = (uint4*)(digests + i * 8);  // big-endian digest bytes
  out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
  out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
}

// ------------------------------------------------------------------ RemoteCache point ops
// op: 0 = SetInsert, 1 = SetContains, 2 = SetRemove.  result[0] = 1 when inserted / present /
// removed; result[1] = SID_FULL marker on a full table.
__global__ void k_set_op(Slot* table, uint64_t mask, unsigned long long meta, unsigned long long s0,
                         unsigned long long s1, unsigned long long s2, unsigned long long s3,
                         unsigned long long s4, int op, uint32_t epoch,
                         unsigned long long* issuer_counts, PairSlot* pairs, uint64_t pmask,
                         uint32_t* result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long s[5] = {s0, s1, s2, s3, s4};
  bool created = false;
  const uint32_t sid = table_upsert(table, mask, meta, s, 0xffffffffu, epoch, op == 0, &created);
  result[0] = 0;
  result[1] = sid == SID_FULL;
  if (sid == SID_FULL) return;
  const uint32_t canon = (uint32_t)(meta >> 32) & 0xffffffu;
  if (op == 0) {
    result[0] = created;
    if (created) atomicAdd(&issuer_counts[canon], 1ull);
  } else if (op == 1) {
    result[0] = sid != SID_NONE;
  } else if (sid != SID_NONE) {
    const bool shadow = (table[sid].w[2] & SLOT_SHADOW) != 0;  // counted by another rank: nothing to take off here
    table[sid].w[0] = SLOT_TOMB;
    table[sid].w[1] = 0;
    if (!shadow) atomicAdd(&issuer_counts[canon], (unsigned long long)-1ll);
    result[0] = 1;
  }
}

// Drop every member whose (exp_hour, canonical issuer) matches, or — with any_key — every
// member with exp_hour*3600 <= now (Redis EXPIREAT set by knowncertificates.go:98-104).
__global__ void __launch_bounds__(256) k_sweep(Slot* table, uint64_t nslots, int any_key,
                                               long long now, uint32_t exp_hour_key, uint32_t canon_key,
                                               unsigned long long* issuer_counts, PairSlot* pairs,
                                               uint64_t pmask, unsigned long long* removed) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nslots) return;
  const unsigned long long w0 = table[j].w[0], w1 = table[j].w[1];
  if (w0 == 0ull || w0 == SLOT_TOMB || !(w1 & SLOT_VALID)) return;
  const int32_t eh = (int32_t)(uint32_t)w1;
  const uint32_t canon = (uint32_t)(w1 >> 32) & 0xffffffu;
  const bool hit = any_key ? ((long long)eh * 3600 <= now) : ((uint32_t)eh == exp_hour_key && canon == canon_key);
  if (!hit) return;
  const bool shadow = (table[j].w[2] & SLOT_SHADOW) != 0;  // counted by another rank (Bloom-variant global dedup)
  table[j].w[0] = SLOT_TOMB;
  table[j].w[1] = 0;
  if (shadow) return;
  atomicAdd(&issuer_counts[canon], (unsigned long long)-1ll);
  atomicAdd(removed, 1ull);
}

// Rebuild the (expDate, issuer) → SCARD table from the known-certificate table (lazy: only the
// statistics-style queries SetCardinality / Exists / KeysToChan need it).
__global__ void __launch_bounds__(256) k_build_pairs(const Slot* table, uint64_t nslots, PairSlot* pairs,
                                                     uint64_t pmask, unsigned long long* full) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nslots) return;
  const unsigned long long w0 = table[j].w[0], w1 = table[j].w[1];
  if (w0 == 0ull || w0 == SLOT_TOMB || !(w1 & SLOT_VALID) || (table[j].w[2] & SLOT_SHADOW)) return;
  const uint32_t canon = (uint32_t)(w1 >> 32) & 0xffffffu;
  if (!pair_add(pairs, pmask, ((unsigned long long)(canon + 1) << 32) | (uint32_t)w1, 1)) atomicAdd(full, 1ull);
}

// SetList / SetToChan: gather the serials of one set.  out entries are 48 bytes:
// [u32 len][40 bytes serial][u32 pad].
__global__ void __launch_bounds__(256) k_list(const Slot* table, uint64_t nslots, uint32_t exp_hour_key,
                                              uint32_t canon_key, uint8_t* out, uint64_t cap,
                                              unsigned long long* count) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nslots) return;
  const unsigned long long w0 = table[j].w[0], w1 = table[j].w[1];
  if (w0 == 0ull || w0 == SLOT_TOMB || !(w1 & SLOT_VALID) || (table[j].w[2] & SLOT_SHADOW)) return;
  if ((uint32_t)w1 != exp_hour_key || ((uint32_t)(w1 >> 32) & 0xffffffu) != canon_key) return;
  const unsigned long long k = atomicAdd(count, 1ull);
  if (k >= cap) return;
  unsigned long long* o = (unsigned long long*)(out + k * 48);
  o[0] = (w1 >> 56) & 0x7full;
#pragma unroll
  for (int q = 0; q < 5; q++) o[1 + q] = table[j].w[3 + q];
}

// KeysToChan: dump the non-empty (expDate, issuer) pairs.
__global__ void __launch_bounds__(256) k_pairs(const PairSlot* pairs, uint64_t npairs,
                                               unsigned long long* out, uint64_t cap,
                                               unsigned long long* count) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= npairs) return;
  const unsigned long long k = pairs[j].key, c = pairs[j].count;
  if (k == 0ull || c == 0ull) return;
  const unsigned long long at = atomicAdd(count, 1ull);
  if (at >= cap) return;
  out[2 * at] = k;
  out[2 * at + 1] = c;
}

// ------------------------------------------------------------------ synthetic generator
__global__ void __launch_bounds__(256) k_synth_len(SynthCfg c, uint64_t first, uint64_t n,
                                                   uint64_t* offsets) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  BackWriter w{nullptr, SYNTH_MAX_LEN};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  offsets[i + 1] = SYNTH_MAX_LEN - w.pos;
  if (i == 0) offsets[0] = 0;
}

__global__ void __launch_bounds__(256) k_synth_emit(SynthCfg c, uint64_t first, uint64_t n,
                                                    const uint64_t* offsets, uint8_t* payload,
                                                    uint32_t* issuer_idx, uint8_t* entry_type) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
  BackWriter w{payload + offsets[i], len};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  issuer_idx[i] = iss;
  entry_type[i] = et;
}


// raw get-entries form: lens[2i] = leaf_input bytes, lens[2i+1] = extra_data bytes (scanned into bounds by the host)
__global__ void __launch_bounds__(256) k_synth_entries_len(SynthCfg c, uint64_t first, uint64_t n, uint64_t* bounds) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  BackWriter w{nullptr, SYNTH_ENTRY_MAX};
  const uint32_t leaf = synth_entry_emit(c, first + i, w);
  bounds[2 * i + 1] = leaf;
  bounds[2 * i + 2] = SYNTH_ENTRY_MAX - w.pos - leaf;
  if (i == 0) bounds[0] = 0;
}

__global__ void __launch_bounds__(256) k_synth_entries_emit(SynthCfg c, uint64_t first, uint64_t n,
                                                            const uint64_t* bounds, uint8_t* blob) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = (uint32_t)(bounds[2 * i + 2] - bounds[2 * i]);
  BackWriter w{blob + bounds[2 * i], len};
  synth_entry_emit(c, first + i, w);
}

}  // namespace ctmr
