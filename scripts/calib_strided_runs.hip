// calib_strided_runs.hip — what does the memory system deliver for the raw path's access pattern?
// k_decode_match reads, of every ≈ 3 023-byte get-entries record, one contiguous run of ≈ 838 bytes at an arbitrary byte
// offset (Chain[0], compared with the registered copy) plus a header line; the records of a launch lie back to back, so the
// device sees runs of ≈ 1 KiB separated by gaps of ≈ 2 KiB.  The streaming ceiling (≈ 6.3 TB/s for whole-buffer copies) is
// the wrong yardstick if that pattern itself is slower.  This tool measures it: E records of `stride` bytes, of each the run
// [off, off + run) is read by one wave (64 lanes × 16 bytes per instruction, `per_step` records in flight together like the
// kernel's MATCH_PER_STEP) and xor-reduced; variants: the run's start aligned to 16 bytes or not (unaligned dwordx4, as the
// kernel issues them), non-temporal or default loads, 8 or 4 waves per SIMD.  Reported: useful GB/s (run bytes) and line GB/s
// (the 128-byte lines the runs touch).  Not part of the product.
//   usage: calib_strided_runs [records = 20000000] [stride = 3023] [run = 838]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));

template <bool ALIGNED, bool NT, int PER_STEP, int HDR = 0>
__global__ void __launch_bounds__(256) k_runs(const uint8_t* buf, uint64_t n, uint32_t stride, uint32_t run, uint32_t off, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  uint32_t acc = 0;
  if (HDR) {  // what the decode reads besides: HDR isolated words per record, one record per lane (a line each: the leaf header
    // at the record's start, the CtExtensions length behind the certificate, the chain header of a precertificate entry);
    // HDR = 3 makes the third DEPEND on the first, as the decoder's position does
    const uint64_t r = wave * 64 + lane;
    if (r < n) {
      const uint32_t h0 = *(const uint32_t*)(buf + r * stride + 2);
      acc ^= h0;
      if (HDR >= 2) acc ^= *(const uint32_t*)(buf + r * stride + 1540);
      if (HDR >= 3) acc ^= *(const uint32_t*)(buf + r * stride + 1555 + (h0 & 1u));
    }
  }
  for (uint32_t k = 0; k < 64; k += PER_STEP) {  // one wave: 64 consecutive records, PER_STEP at a time
    u32x4 v[PER_STEP][2];
#pragma unroll
    for (int u = 0; u < PER_STEP; u++) {
      const uint64_t r = wave * 64 + k + u;
      uint64_t lo = r * stride + off;
      if (ALIGNED) lo &= ~15ull;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t o = lane * 16u + 1024u * h;
        v[u][h] = (u32x4){0, 0, 0, 0};
        if (r < n && o < run) {
          const uint8_t* p = buf + lo + o;
          if (ALIGNED) v[u][h] = NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p;
          else v[u][h] = NT ? (u32x4)__builtin_nontemporal_load((const u32x4u*)p) : (u32x4)*(const u32x4u*)p;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PER_STEP; u++)
#pragma unroll
      for (int h = 0; h < 2; h++) acc ^= v[u][h].x ^ v[u][h].y ^ v[u][h].z ^ v[u][h].w;
  }
  if (acc == 0x12345677u) out[0] = 1;
}

template <bool ALIGNED, bool NT, int PER_STEP, int HDR = 0>
static void run_one(const uint8_t* buf, uint64_t n, uint32_t stride, uint32_t run, uint32_t off, uint32_t* out, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  const uint64_t waves = (n + 63) / 64;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_runs<ALIGNED, NT, PER_STEP, HDR>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, buf, n, stride, run, off, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // 128-byte lines a run touches, averaged over the records' alignments
  double lines = 0;
  for (uint64_t r = 0; r < 4096; r++) {
    uint64_t lo = r * stride + off;
    if (ALIGNED) lo &= ~15ull;
    lines += (double)((lo + run + 127) / 128 - lo / 128);
  }
  lines = lines / 4096 + HDR;
  printf("{\"what\": \"%s\", \"records\": %llu, \"stride\": %u, \"run\": %u, \"ms\": %.3f, \"useful_GB_per_s\": %.0f, \"line_GB_per_s\": %.0f, "
         "\"lines_per_run\": %.2f}\n", what, (unsigned long long)n, stride, run, best, n * (double)run / best / 1e6, n * lines * 128 / best / 1e6, lines);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000ull;
  const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 3023u, run = argc > 3 ? (uint32_t)atoi(argv[3]) : 838u;
  uint8_t* buf;
  uint32_t* out;
  const uint64_t bytes = n * stride + 4096;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 0x5a, bytes));
  CK(hipMemset(out, 0, 4));
  CK(hipDeviceSynchronize());
  const uint32_t off = 1583u;  // where Chain[0] begins in a synthetic X509 entry's record, more or less: any odd offset will do
  run_one<false, true, 4>(buf, n, stride, run, off, out, "unaligned dwordx4, non-temporal, 4 records in flight per wave (the kernel's loads)");
  run_one<false, false, 4>(buf, n, stride, run, off, out, "unaligned dwordx4, default policy, 4 in flight");
  run_one<true, true, 4>(buf, n, stride, run, off, out, "16-byte aligned dwordx4, non-temporal, 4 in flight");
  run_one<true, false, 4>(buf, n, stride, run, off, out, "16-byte aligned dwordx4, default policy, 4 in flight");
  run_one<false, true, 8>(buf, n, stride, run, off, out, "unaligned dwordx4, non-temporal, 8 in flight");
  run_one<true, true, 8>(buf, n, stride, run, off, out, "16-byte aligned dwordx4, non-temporal, 8 in flight");
  run_one<false, true, 2>(buf, n, stride, run, off, out, "unaligned dwordx4, non-temporal, 2 in flight");
  run_one<false, true, 4, 1>(buf, n, stride, run, off, out, "unaligned, non-temporal, 4 in flight + 1 isolated header word per record");
  run_one<false, true, 4, 2>(buf, n, stride, run, off, out, "unaligned, non-temporal, 4 in flight + 2 isolated header words per record");
  run_one<false, true, 4, 3>(buf, n, stride, run, off, out, "unaligned, non-temporal, 4 in flight + 3 isolated header words per record (one dependent)");
  // the same bytes without gaps: a contiguous stream read the same way (stride = run rounded up to 16)
  run_one<true, true, 4>(buf, n, (run + 15u) & ~15u, run, 0u, out, "no gaps: records back to back, aligned, non-temporal, 4 in flight");
  // the packed map's pattern: ≈ 930 of every 1 523 bytes
  run_one<false, true, 4>(buf, n, 1523u, 930u, 0u, out, "930 of every 1523 bytes (the packed map's reads), unaligned, non-temporal");
  return 0;
}
