// Cost of filler instructions between v_mfma_f32_32x32x2_f32 on one wave per SIMD (inline asm, exact streams).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#define VALU1 asm volatile("v_xor_b32 %0, %0, %1" : "+v"(t0) : "v"(one));
#define VALU2 asm volatile("v_xor_b32 %0, %0, %1" : "+v"(t1) : "v"(one));
#define DSRD  asm volatile("ds_read_b32 %0, %1" : "=v"(d0) : "v"(ldsaddr));
#define VMEMRD asm volatile("global_load_dword %0, %1, off" : "=v"(g0) : "v"(gp));
#define SALU  asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc));
#define WAITL asm volatile("s_waitcnt lgkmcnt(8)");
#define NOP   asm volatile("s_nop 0");

template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, const float* w, int iters, long long* cyc) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int j = 0; j < 16; ++j) { acc0[j] = 0; acc1[j] = 0; }
  float a = threadIdx.x, b = 2.f, d0 = 0, g0 = 0;
  int t0 = threadIdx.x, t1 = 3, one = 1, sc = 0;
  int ldsaddr = (threadIdx.x & 255) * 4;
  const float* gp = w + (threadIdx.x & 63);
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      MFMA(acc0)
      if (VAR == 1) { VALU1 }
      if (VAR == 2) { VALU1 VALU2 }
      if (VAR == 3) { VALU1 VALU2 VALU1 VALU2 }
      if (VAR == 4) { VALU1 VALU2 VALU1 VALU2 VALU1 VALU2 VALU1 VALU2 }
      if (VAR == 5) { DSRD }
      if (VAR == 6) { DSRD VALU1 VALU2 }
      if (VAR == 7) { VMEMRD }
      if (VAR == 8) { SALU SALU SALU SALU }
      if (VAR == 9) { DSRD VALU1 VALU2 WAITL }
      if (VAR == 10) { NOP NOP NOP NOP }
      if (VAR == 11) { DSRD DSRD VALU1 VALU2 VALU1 VALU2 VMEMRD }
      MFMA(acc1)
      if (VAR == 1) { VALU2 }
      if (VAR == 2) { VALU1 VALU2 }
      if (VAR == 3) { VALU1 VALU2 VALU1 VALU2 }
      if (VAR == 4) { VALU1 VALU2 VALU1 VALU2 VALU1 VALU2 VALU1 VALU2 }
      if (VAR == 5) { DSRD }
      if (VAR == 6) { DSRD VALU1 VALU2 }
      if (VAR == 8) { SALU SALU SALU SALU }
      if (VAR == 9) { DSRD VALU1 VALU2 WAITL }
      if (VAR == 10) { NOP NOP NOP NOP }
      if (VAR == 11) { DSRD DSRD VALU1 VALU2 VALU1 VALU2 }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  long long c1 = __builtin_readcyclecounter();
  float s = d0 + g0 + t0 + t1 + sc;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int VAR> void run(const char* name) {
  int iters = 500;
  float *out, *w; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&w, 1 << 20); hipMalloc(&cyc, 8);
  hipLaunchKernelGGL((k<VAR>), dim3(256), dim3(256), 0, 0, out, w, iters, cyc);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.2f cycles per MFMA\n", name, (double)c / (iters * 16.0));
}
int main() {
  run<0>("bare"); run<1>("+1 VALU"); run<2>("+2 VALU"); run<3>("+4 VALU"); run<4>("+8 VALU");
  run<5>("+1 DS read"); run<6>("+1 DS read +2 VALU"); run<7>("+1 VMEM per 2 MFMA"); run<8>("+4 SALU");
  run<9>("+1 DS +2 VALU + s_waitcnt lgkmcnt(8)"); run<10>("+4 s_nop"); run<11>("+2 DS +4 VALU (+1 VMEM/2)");
  return 0;
}
