// calib_atomics_regions.hip — VERDICT r04 #3b: would a REGION-PARTITIONED insert pay for the 8-byte index?
// The known-certificate index is one 8-byte word per slot (2–4 GB for the 100 M batch); pass 1 of the insert is one
// agent-scope atomicCAS per new key on a random word of it.  If the keys were first partitioned by index
// region (the XM_OWNER staging + k_key_gather machinery could do that), every phase of the insert would hit a window of the
// index small enough for the Infinity Cache (256 MB) or an XCD's L2 (4 MB).  This tool measures what that buys: the rate of
// random agent-scope CAS (result used, as the insert uses it) when all keys of a launch fall into a window of W bytes, for W
// from the whole index down to 1 MB — plus workgroup-scope CAS (executed in the issuing XCD's L2; NOT coherent across XCDs:
// an upper bound for a design that also routes every region to one XCD) and plain loads for reference.
// Not part of the product.   usage: calib_atomics_regions [keys = 94000000]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

template <int MODE>  // 0 agent CAS, 1 workgroup CAS, 2 plain load
__global__ void __launch_bounds__(256) k_rand(unsigned long long* table, uint64_t mask, uint64_t n, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long z = (i + 1) * 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  unsigned long long* p = table + (z & mask);
  unsigned long long old = 1;
  if (i < n) {
    if (MODE == 2) {
      old = *(volatile unsigned long long*)p;
    } else {
      unsigned long long e = 0;
      if (MODE == 0) __hip_atomic_compare_exchange_strong(p, &e, z | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_compare_exchange_strong(p, &e, z | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = e;
    }
  }
  if (old == 0x1234567ull) out[0] = 1;
}

template <int MODE>
static void run(unsigned long long* table, uint64_t words, uint64_t n, uint32_t* out, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(table, 0, words * 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rand<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, table, words - 1, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("{\"what\": \"%s\", \"window_MB\": %.0f, \"keys\": %llu, \"keys_per_word\": %.2f, \"ms\": %.4f, \"Gops_per_s\": %.2f}\n", what,
         words * 8 / 1048576.0, (unsigned long long)n, (double)n / words, best, n / best / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 94000000ull;
  unsigned long long* table;
  uint32_t* out;
  CK(hipMalloc(&table, 1ull << 32));  // 4 GB: 2^29 words
  CK(hipMalloc(&out, 64));
  // load factor matters (a CAS on a taken word fails: same memory transaction): the window sweep keeps n fixed, so small
  // windows see mostly failing CAS — also run each window with n = words / 4 (the bench's load)
  for (int lg = 29; lg >= 17; lg -= 2) {
    const uint64_t words = 1ull << lg;
    run<0>(table, words, n, out, "agent CAS, all keys of the batch into the window");
    const uint64_t nq = words / 4 > n ? n : words / 4;
    if (nq >= (1u << 16)) run<0>(table, words, nq, out, "agent CAS, window at load 1/4");
  }
  for (int lg = 29; lg >= 17; lg -= 4) {
    run<1>(table, 1ull << lg, n, out, "workgroup CAS (XCD L2, not coherent), all keys into the window");
    run<2>(table, 1ull << lg, n, out, "plain 8-byte load, all keys into the window");
  }
  return 0;
}
