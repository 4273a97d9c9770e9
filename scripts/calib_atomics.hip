// calib_atomics.hip — what a random 8-byte atomic on a large table costs by SCOPE and by whether its result is used.
// Not part of the product: a measurement tool for DESIGN.md §7 (the set insert is bound by device-scope atomicCAS).
// Workgroup-scope atomics execute in the issuing XCD's L2 and are NOT coherent across XCDs: the numbers say what an
// insert whose keys were first routed to the XCD that owns their table region could gain, nothing more.
//   modes  0 agent CAS (the product's claim)      1 workgroup CAS        2 agent fetch_or, result unused
//          3 workgroup fetch_or, result unused    4 plain 8-byte load    5 plain 8-byte store
//          6..9 the insert's pair: agent CAS on word 0 of the slot, then the 64-byte slot image (4 x 16 B from the claiming
//               lane) — 6 plain stores, 7 non-temporal (nt), 8 sc1, 9 sc0 sc1: does a store that leaves the L2 at once
//               meet the line the atomic has just dirtied while the Infinity Cache still holds it?
//          10   the same pair with the image stored by four adjacent lanes per slot (the product's store_slots_wave)
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
    } else if (MODE == 5) {
      *p = z | 1ull;
    } else {
      unsigned long long e = 0;
      __hip_atomic_compare_exchange_strong(p, &e, z | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = e;
      if (MODE != 10 && e == 0ull) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
        v.x = (uint32_t)z | 1u; v.y = (uint32_t)(z >> 32); v.z = 7u; v.w = (uint32_t)i;
        u32x4* q = (u32x4*)p;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (MODE == 6) q[k] = v;
          else if (MODE == 7) __builtin_nontemporal_store(v, q + k);
          else if (MODE == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q + k), "v"(v) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q + k), "v"(v) : "memory");
        }
      }
    }
  }
  if (MODE == 10) {  // the product's form (store_slots_wave): four adjacent lanes emit one whole slot, 16 slots per instruction
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t j = z & mask;
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
  run<6>(table, slots, n, out, "agent CAS + 64-byte slot image, plain stores");
  run<7>(table, slots, n, out, "agent CAS + 64-byte slot image, nt stores");
  run<8>(table, slots, n, out, "agent CAS + 64-byte slot image, sc1 stores");
  run<9>(table, slots, n, out, "agent CAS + 64-byte slot image, sc0 sc1 stores");
  run<10>(table, slots, n, out, "agent CAS + 64-byte slot image, four lanes per slot (the product's store_slots_wave)");
  run<6>(table, slots, n, out, "agent CAS + 64-byte slot image, plain stores (again)");
  return 0;
}
