// Practical fp32 MFMA ceiling for the conv kernel's issue pattern: NACC independent 32x32x2 accumulators per wave,
// WPS waves per SIMD, optional filler instructions per MFMA (LDS reads / VALU) to mimic the real loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int FILL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  f32x16 acc[NACC];
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0 + i;
  __syncthreads();
  float a = a0 + threadIdx.x, b = b0;
  int addr = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int m = 0; m < NACC; ++m) {
        if (FILL >= 1) { a = lds[(addr ^ (u * 2 + m)) & 4095]; }
        if (FILL >= 2) { addr = (addr * 3 + it) & 4095; }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 16; ++j) s += acc[m][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int FILL>
void run(int blocks_per_cu, const char* name) {
  int iters = 2000;
  int nblk = 256 * blocks_per_cu;
  float* out; hipMalloc(&out, nblk * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, FILL>), dim3(nblk), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, FILL>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)nblk * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("%-34s blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4, 0>(1, "4 acc, no filler");
  run<2, 0>(1, "2 acc, no filler");
  run<2, 0>(2, "2 acc, no filler");
  run<2, 0>(3, "2 acc, no filler");
  run<1, 0>(2, "1 acc, no filler");
  run<2, 1>(2, "2 acc, ds_read+xor per mfma");
  run<2, 2>(2, "2 acc, ds_read+xor+valu per mfma");
  run<2, 1>(1, "2 acc, ds_read+xor per mfma");
  run<4, 1>(1, "4 acc, ds_read+xor per mfma");
  return 0;
}
