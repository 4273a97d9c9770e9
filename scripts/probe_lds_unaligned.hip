#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;
extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
__global__ void __launch_bounds__(64) k(const uint32_t* in, uint32_t* out, uint32_t off) {
  ((uint32_t*)smem)[threadIdx.x] = in[threadIdx.x];
  ((uint32_t*)smem)[threadIdx.x + 64] = in[threadIdx.x + 64];
  __syncthreads();
  const uint32_t a = *(const u32u*)(smem + threadIdx.x * 3 + off);
  const uint32_t b = *(const u16u*)(smem + threadIdx.x * 5 + off);
  out[threadIdx.x] = a;
  out[threadIdx.x + 64] = b;
}
int main() {
  uint32_t h[128], *d_in, *d_out, r[128];
  for (int i = 0; i < 128; i++) h[i] = 0x03020100u + 0x04040404u * i;
  hipMalloc(&d_in, 512); hipMalloc(&d_out, 512);
  hipMemcpy(d_in, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d_in, d_out, 1u);
  hipMemcpy(r, d_out, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  const uint8_t* hb = (const uint8_t*)h;
  for (int t = 0; t < 64; t++) {
    uint32_t wa, wb = 0; memcpy(&wa, hb + t * 3 + 1, 4); memcpy(&wb, hb + t * 5 + 1, 2);
    if (r[t] != wa || r[t + 64] != wb) bad++;
  }
  printf("unaligned LDS reads: %d mismatches (first got %08x)\n", bad, r[0]);
  return 0;
}
