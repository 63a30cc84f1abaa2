// Find which ingredient of the conv inner loop costs matrix-pipe throughput.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE bits: 1 = LDS read per MFMA (xor swizzled addr), 2 = global load per 2 MFMA (ping-pong regs, consumed 8 steps later),
//            4 = extra VALU (4 per 2 MFMA), 8 = SALU walk per tap, 16 = sched_barrier per step
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, int iters, int wstride) {
  __shared__ float lds[8192];
  f32x16 acc[2];
  for (int m = 0; m < 2; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.f + i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a0[2][8], b0[8], a1[2][8], b1[8];
  for (int kp = 0; kp < 8; ++kp) { b0[kp] = w[kp * 64 + lane]; a0[0][kp] = lds[lane + kp]; a0[1][kp] = lds[lane + 64 + kp]; b1[kp] = b0[kp]; a1[0][kp] = a0[0][kp]; a1[1][kp] = a0[1][kp]; }
  int off = 0, kw = 0, kh = 0;
  const float* wq = w + lane;
  int base0 = lane, base1 = lane + 1000;
#define STEP(NXT_A, NXT_B, CUR_A, CUR_B)                                              \
  _Pragma("unroll") for (int kp = 0; kp < 8; ++kp) {                                  \
    if (MODE & 2) NXT_B[kp] = wq[kp * 64];                                             \
    if (MODE & 1) {                                                                   \
      int lv0 = base0 + off, lv1 = base1 + off;                                       \
      if (MODE & 4) { lv0 = (lv0 * 16 + ((lv0 >> 1) & 15)) & 8191; lv1 = (lv1 * 16 + ((lv1 >> 1) & 15)) & 8191; } \
      NXT_A[0][kp] = lds[(lv0 ^ (2 * kp)) & 8191];                                    \
      NXT_A[1][kp] = lds[(lv1 ^ (2 * kp)) & 8191];                                    \
    }                                                                                 \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR_A[0][kp], CUR_B[kp], acc[0], 0, 0, 0); \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR_A[1][kp], CUR_B[kp], acc[1], 0, 0, 0); \
    if (MODE & 16) __builtin_amdgcn_sched_barrier(0);                                 \
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE & 8) { off += 1; if (++kw == 3) { kw = 0; off += 31; if (++kh == 3) { kh = 0; off += 100; } } off &= 255; wq += wstride; if (it % 27 == 26) wq = w + lane; }
    STEP(a1, b1, a0, b0)
    if (MODE & 8) { off += 1; if (++kw == 3) { kw = 0; off += 31; if (++kh == 3) { kh = 0; off += 100; } } off &= 255; wq += wstride; }
    STEP(a0, b0, a1, b1)
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m) for (int j = 0; j < 16; ++j) s += acc[m][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int bpc, int wstride, const char* name) {
  int iters = 1000, nblk = 256 * bpc;
  float *out, *w; hipMalloc(&out, nblk * 256 * 4); hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(256), 0, 0, out, w, 10, wstride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(256), 0, 0, out, w, iters, wstride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)nblk * 4 * iters * 2 * 16 * 2.0 * 32 * 32 * 2;
  printf("%-52s bpc=%d: %.3f ms  %.1f TFLOP/s\n", name, bpc, ms, flops / ms / 1e9);
  hipFree(out); hipFree(w);
}
int main() {
  run<0>(2, 0, "mfma only");
  run<1>(2, 0, "+lds read/mfma");
  run<1 | 4>(2, 0, "+lds read + swizzle valu");
  run<2>(2, 0, "+global load (L1 hit)");
  run<2>(2, 512, "+global load (streaming 2KB/tap)");
  run<1 | 2 | 4>(2, 512, "+lds +valu +gload stream");
  run<1 | 2 | 4 | 8>(2, 512, "+lds +valu +gload +salu");
  run<1 | 2 | 4 | 8 | 16>(2, 512, "+lds +valu +gload +salu +sched_barrier");
  run<1 | 2 | 4 | 8 | 16>(1, 512, "same, 1 block/CU");
  run<1 | 2 | 4 | 8 | 16>(3, 512, "same, 3 blocks/CU");
  return 0;
}
