// calib_fetch.hip — calibration of rocprofv3's FETCH_SIZE on the map kernel's access pattern, and
// the bandwidth ceiling of that pattern.  Not part of the product: a measurement tool.
//
// The map kernel (k_map_win*) reads per-lane windows: W consecutive 16-byte chunks starting at an
// arbitrary 16-byte-aligned address, lanes ≈1.5 KB apart, a few windows per certificate.  This
// program issues exactly that pattern on a buffer with known geometry, so that the unique bytes
// touched at 64-B and at 128-B granularity are known in closed form, and reports the time per
// launch.  Run it under `rocprofv3 --pmc FETCH_SIZE` to get the counter for each launch.
//
//   calib_fetch [lanes=8388608] [stride=1536]
// kernels (one launch each, after a warm-up):
//   k_stream            plain coalesced float4 copy-read of the whole buffer (the guide's 2x case)
//   k_win<off,nwin>     per lane: nwin windows of 16 chunks at lane*stride + off + w*512
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

extern __shared__ uint8_t smem[];

__global__ void __launch_bounds__(256) k_stream(const uint4* p, uint64_t nvec, uint32_t* out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// one lane = one "certificate": nwin windows of 16 chunks, each window one burst of 16 independent
// global_load_dwordx4, staged through LDS exactly like WinReader::refill (lane stride 272 B)
template <int NWIN>
__global__ void __launch_bounds__(64) k_win(const uint8_t* buf, uint64_t lanes, uint32_t stride, uint32_t off,
                                            uint32_t wstep, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= lanes) return;
  uint4* win = (uint4*)(smem + threadIdx.x * 272);
  uint32_t acc = 0;
#pragma unroll 1
  for (int w = 0; w < NWIN; w++) {
    const uint4* src = (const uint4*)(buf + i * stride + off + (uint64_t)w * wstep);
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = src[k];
#pragma unroll
    for (int k = 0; k < 16; k++) win[k] = v[k];
    // dependent chain through the window, like a TLV walk: next position from the data
    uint32_t pos = 0;
#pragma unroll 1
    for (int s = 0; s < 8; s++) {
      const uint32_t x = ((const uint32_t*)win)[pos & 63];
      acc ^= x;
      pos = (pos + 5 + (x & 3)) & 63;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// streaming write, 16 B per lane, coalesced (WRITE_SIZE calibration)
__global__ void __launch_bounds__(256) k_fill(uint4* p, uint64_t nvec) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * 256)
    p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

// the reduce's access pattern: one 64-bit atomicCAS on a random 64-byte slot of a large table, then the
// whole 64-byte slot written by four adjacent lanes (k_insert's cooperative store).  mode 0: CAS only,
// 1: CAS + slot store, 2: plain 16-byte load of the slot instead of the CAS (random read ceiling)
__global__ void __launch_bounds__(256) k_rand_rmw(unsigned long long* table, uint64_t mask, uint64_t n, int mode,
                                                  uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  unsigned long long z = (i + 1) * 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  const uint64_t j = z & mask;
  unsigned long long old = 1;
  if (i < n) {
    if (mode == 2) {
      const uint4 v = *(const uint4*)(table + j * 8);
      old = v.x | v.y;
    } else {
      old = atomicCAS(table + j * 8, 0ull, z | 1ull);
    }
  }
  if (mode == 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t src = 16u * r + (lane >> 2);
      const uint64_t sj = __shfl(j, src);
      const unsigned long long so = __shfl(old, src);
      if (so == 0ull) ((uint4*)(table + sj * 8))[lane & 3u] = make_uint4((uint32_t)z | 1u, 7u, (uint32_t)sj, lane);
    }
  }
  if (old == 0x1234567ull) out[0] = 1;
}

static double unique_bytes(uint64_t lanes, uint32_t stride, uint32_t off, uint32_t wstep, int nwin, uint32_t gran) {
  // windows of one lane do not overlap windows of another when stride >= off + nwin*wstep + 256
  double tot = 0;
  uint64_t sample = lanes < 65536 ? lanes : 65536;
  for (uint64_t i = 0; i < sample; i++) {
    uint64_t prev_hi = 0;
    for (int w = 0; w < nwin; w++) {
      uint64_t a = i * (uint64_t)stride + off + (uint64_t)w * wstep;
      uint64_t lo = a / gran, hi = (a + 255) / gran;
      if (w && lo <= prev_hi) lo = prev_hi + 1;
      if (hi >= lo) tot += (double)(hi - lo + 1) * gran;
      prev_hi = hi;
    }
  }
  return tot * ((double)lanes / sample);
}

int main(int argc, char** argv) {
  const uint64_t lanes = argc > 1 ? strtoull(argv[1], 0, 10) : 8388608ull;
  const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 1536u;
  const uint64_t bytes = lanes * stride + 4096;
  uint8_t* buf;
  uint32_t* out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0x5a, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  printf("{\"lanes\": %llu, \"stride\": %u, \"buffer_bytes\": %llu}\n", (unsigned long long)lanes, stride,
         (unsigned long long)bytes);
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep)
      printf("{\"kernel\": \"k_stream\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", (double)bytes, ms,
             bytes / ms / 1e6);
  }
  struct Case { int nwin; uint32_t off, wstep; } cases[] = {
      {1, 0, 512}, {1, 64, 512}, {1, 16, 512}, {2, 0, 512}, {2, 16, 512}, {3, 0, 512}, {3, 16, 512}, {3, 48, 400}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      const unsigned grid = (unsigned)((lanes + 63) / 64);
      if (c.nwin == 1) hipLaunchKernelGGL(k_win<1>, dim3(grid), dim3(64), 64 * 272, 0, buf, lanes, stride, c.off, c.wstep, out);
      if (c.nwin == 2) hipLaunchKernelGGL(k_win<2>, dim3(grid), dim3(64), 64 * 272, 0, buf, lanes, stride, c.off, c.wstep, out);
      if (c.nwin == 3) hipLaunchKernelGGL(k_win<3>, dim3(grid), dim3(64), 64 * 272, 0, buf, lanes, stride, c.off, c.wstep, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) {
        const double u64 = unique_bytes(lanes, stride, c.off, c.wstep, c.nwin, 64);
        const double u128 = unique_bytes(lanes, stride, c.off, c.wstep, c.nwin, 128);
        printf("{\"kernel\": \"k_win<%d>\", \"off\": %u, \"wstep\": %u, \"requested_bytes\": %.0f, \"unique64\": %.0f, "
               "\"unique128\": %.0f, \"ms\": %.4f, \"GBps_requested\": %.1f, \"GBps_unique128\": %.1f}\n",
               c.nwin, c.off, c.wstep, (double)lanes * c.nwin * 256, u64, u128, ms, lanes * c.nwin * 256.0 / ms / 1e6,
               u128 / ms / 1e6);
      }
    }
  }
  // ---- WRITE_SIZE calibration: coalesced fill of the whole buffer
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep)
      printf("{\"kernel\": \"k_fill\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", (double)bytes, ms,
             bytes / ms / 1e6);
  }
  // ---- random 64-byte-slot RMW ceiling (the known-certificate insert), table = 2^27 slots (8.6 GB), load 0.35
  {
    const uint64_t slots = 1ull << 27, nkeys = 47000000ull;
    unsigned long long* table;
    CK(hipMalloc(&table, slots * 64));
    for (int mode = 0; mode < 3; mode++) {
      CK(hipMemset(table, 0, slots * 64));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rand_rmw, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, 0, table, slots - 1, nkeys, mode, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("{\"kernel\": \"k_rand_rmw\", \"mode\": %d, \"keys\": %llu, \"slots\": %llu, \"ms\": %.4f, \"Gkeys_per_s\": %.2f}\n",
             mode, (unsigned long long)nkeys, (unsigned long long)slots, ms, nkeys / ms / 1e6);
    }
    CK(hipFree(table));
  }
  CK(hipDeviceSynchronize());
  return 0;
}
