// Chip-wide sustained rate of v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32 (one wave per SIMD, independent accumulators),
// for durations long enough (several ms) that power management settles: gives the PRACTICAL fp32 MFMA ceiling of this part.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
  }
  float s = 0.f;
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 4; ++j) s += acc[m][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
  }
  float s = 0.f;
  for (int m = 0; m < NACC; ++m) for (int j = 0; j < 16; ++j) s += acc[m][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void timeit(const char* name, F launch, double flops_per_iter, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s iters=%d: %.3f ms  %.1f TFLOP/s\n", name, iters, ms, flops_per_iter * iters / ms / 1e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  for (int cu : {256, 512}) {
    printf("workgroups = %d\n", cu);
    timeit("16x16x4, 54 acc", [&](int it) { hipLaunchKernelGGL(k16<54>, dim3(cu), dim3(256), 0, 0, out, it, 1.f, 2.f); }, (double)cu * 4 * 54 * 2.0 * 16 * 16 * 4, 4000);
    timeit("16x16x4, 54 acc (long)", [&](int it) { hipLaunchKernelGGL(k16<54>, dim3(cu), dim3(256), 0, 0, out, it, 1.f, 2.f); }, (double)cu * 4 * 54 * 2.0 * 16 * 16 * 4, 40000);
    timeit("32x32x2, 8 acc", [&](int it) { hipLaunchKernelGGL(k32<8>, dim3(cu), dim3(256), 0, 0, out, it, 1.f, 2.f); }, (double)cu * 4 * 8 * 2.0 * 32 * 32 * 2, 16000);
    timeit("32x32x2, 8 acc (long)", [&](int it) { hipLaunchKernelGGL(k32<8>, dim3(cu), dim3(256), 0, 0, out, it, 1.f, 2.f); }, (double)cu * 4 * 8 * 2.0 * 32 * 32 * 2, 160000);
  }
  return 0;
}
