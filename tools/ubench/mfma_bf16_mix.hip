// Gate of the "Winograd domain in 3 x bf16" experiment: does a bf16 MFMA stream on one wave of a SIMD keep its
// rate while the SIMD's OTHER wave executes vector-ALU + LDS-store work (the input transform of the Winograd kernels)?  For fp32
// MFMA the answer is no (every VALU instruction on the SIMD is paid in matrix time).
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run the matrix stream, waves 4-7 (their SIMD partners) the vector stream.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_mix mfma_bf16_mix.hip ;  run: ./mfma_bf16_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MM: 0 no matrix stream, 1 fp32 32x32x2, 2 bf16 32x32x16.   VM: 0 no vector stream, 1 VALU fma chain, 2 VALU + ds_write_b32,
// 3 VALU + ds_write_b128, 4 ds_write_b32 only
template <int MM, int VM>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float res = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (MM == 1) {
      f32x16 acc[4];
      for (int m = 0; m < 4; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
      const float a = 1.f + lane, b = 2.f - lane;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
      for (int m = 0; m < 4; ++m) res += acc[m][3];
    } else if (MM == 2) {
      f32x16 acc[4];
      for (int m = 0; m < 4; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane + e); b[e] = (__bf16)(0.5f * e); }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      for (int m = 0; m < 4; ++m) res += acc[m][3];
    }
  } else if (VM != 0) {
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = 1.f + lane * 0.001f + e;
    float* lp = lds + (wave - 4) * 4096 + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (VM != 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 1.0001f, 0.5f);        // 8 VALU
        }
        if (VM == 2 || VM == 4) { lp[(u & 7) * 256] = x[u & 7]; lp[(u & 7) * 256 + 1] = x[(u + 1) & 7]; }      // 2 ds_write_b32
        if (VM == 3) *(float4*)(lp + (u & 7) * 256) = make_float4(x[0], x[1], x[2], x[3]);                       // 1 ds_write_b128
      }
    }
    for (int e = 0; e < 8; ++e) res += x[e];
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = res + lds[threadIdx.x];
}

template <int MM, int VM>
void run(const char* name) {
  const int iters = 2000, nblk = 256;
  float* out; long long* cyc;
  hipMalloc(&out, nblk * 512 * 4); hipMalloc(&cyc, 8 * 8);
  hipMemset(cyc, 0, 64);
  hipLaunchKernelGGL((k<MM, VM>), dim3(nblk), dim3(512), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MM, VM>), dim3(nblk), dim3(512), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double per = 16.0 * iters;
  printf("%-58s %8.3f ms | cycles per step: matrix wave %7.1f (per MFMA), vector wave %7.1f (per 8 VALU [+ LDS stores])\n", name, ms,
         MM ? (double)h[0] / per : 0.0, VM ? (double)h[4] / per : 0.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("fp32 32x32x2 MFMA stream alone");
  run<2, 0>("bf16 32x32x16 MFMA stream alone");
  run<0, 1>("vector stream alone: 8 VALU");
  run<0, 2>("vector stream alone: 8 VALU + 2 ds_write_b32");
  run<0, 3>("vector stream alone: 8 VALU + 1 ds_write_b128");
  run<0, 4>("vector stream alone: 2 ds_write_b32");
  run<1, 1>("fp32 MFMA | partner: 8 VALU");
  run<1, 2>("fp32 MFMA | partner: 8 VALU + 2 ds_write_b32");
  run<2, 1>("bf16 MFMA | partner: 8 VALU");
  run<2, 2>("bf16 MFMA | partner: 8 VALU + 2 ds_write_b32");
  run<2, 3>("bf16 MFMA | partner: 8 VALU + 1 ds_write_b128");
  run<2, 4>("bf16 MFMA | partner: 2 ds_write_b32");
  return 0;
}
