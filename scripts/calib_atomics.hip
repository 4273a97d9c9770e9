// calib_atomics.hip — what a random 8-byte atomic on a large table costs by SCOPE and by whether its result is used.
// Not part of the product: a measurement tool for DESIGN.md §7 (the set insert is bound by device-scope atomicCAS).
// Workgroup-scope atomics execute in the issuing XCD's L2 and are NOT coherent across XCDs: the numbers say what an
// insert whose keys were first routed to the XCD that owns their table region could gain, nothing more.
//   modes  0 agent CAS (the product's claim)      1 workgroup CAS        2 agent fetch_or, result unused
//          3 workgroup fetch_or, result unused    4 plain 8-byte load    5 plain 8-byte store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_rand(unsigned long long* table, uint64_t mask, uint64_t n, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long z = (i + 1) * 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  unsigned long long* p = table + (z & mask) * 8;  // 64-byte slots
  unsigned long long old = 1;
  if (i < n) {
    if (MODE == 0) {
      unsigned long long e = 0;
      __hip_atomic_compare_exchange_strong(p, &e, z | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = e;
    } else if (MODE == 1) {
      unsigned long long e = 0;
      __hip_atomic_compare_exchange_strong(p, &e, z | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = e;
    } else if (MODE == 2) {
      (void)__hip_atomic_fetch_or(p, z | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 3) {
      (void)__hip_atomic_fetch_or(p, z | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 4) {
      old = *(volatile unsigned long long*)p;
    } else {
      *p = z | 1ull;
    }
  }
  if (old == 0x1234567ull) out[0] = 1;
}

template <int MODE>
static void run(unsigned long long* table, uint64_t slots, uint64_t n, uint32_t* out, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(table, 0, slots * 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rand<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, table, slots - 1, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("{\"mode\": %d, \"what\": \"%s\", \"keys\": %llu, \"slots\": %llu, \"ms\": %.4f, \"Gops_per_s\": %.2f}\n", MODE, what,
         (unsigned long long)n, (unsigned long long)slots, best, n / best / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t slots = 1ull << (argc > 1 ? atoi(argv[1]) : 27), n = argc > 2 ? strtoull(argv[2], 0, 10) : 47000000ull;
  unsigned long long* table;
  uint32_t* out;
  CK(hipMalloc(&table, slots * 64));
  CK(hipMalloc(&out, 64));
  run<0>(table, slots, n, out, "atomicCAS, agent scope, result used");
  run<1>(table, slots, n, out, "atomicCAS, workgroup scope, result used");
  run<2>(table, slots, n, out, "fetch_or, agent scope, result unused");
  run<3>(table, slots, n, out, "fetch_or, workgroup scope, result unused");
  run<4>(table, slots, n, out, "plain 8-byte load");
  run<5>(table, slots, n, out, "plain 8-byte store");
  return 0;
}
