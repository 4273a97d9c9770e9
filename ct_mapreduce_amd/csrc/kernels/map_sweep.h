// kernels/map_sweep.h — the two baseline designs of the map the default kernel is measured against:
// the literal north-star form (whole certificates copied into an LDS tile) and the naive one (8-byte loads straight
// from global memory).  Compiled only into libctmr_sweep.so (-DCTMR_SWEEP, `python -m ct_mapreduce_amd.build --sweep`,
// used by scripts/sweep.py); the shipped library does not contain them.
#pragma once
#include "map.h"

namespace ctmr {

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

}  // namespace ctmr
