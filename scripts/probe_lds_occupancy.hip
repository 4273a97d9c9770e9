// How many one-wave workgroups fit a CU for a given dynamic LDS size?  (The map kernels are LDS-bound: this is their occupancy.)
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char smem[];
__global__ void __launch_bounds__(64) k(unsigned* out) { smem[threadIdx.x] = 1; __syncthreads(); out[threadIdx.x] = smem[63 - threadIdx.x]; }
int main() {
  int prev = -1;
  for (int s = 8192; s <= 20480; s += 64) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, s) != hipSuccess) { printf("error at %d\n", s); return 1; }
    if (n != prev) printf("dynamic LDS %6d B and up: %d workgroups of one wave per CU\n", s, n);
    prev = n;
  }
  return 0;
}
