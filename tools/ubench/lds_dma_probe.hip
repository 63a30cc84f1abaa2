// LDS-DMA (buffer_load_dwordx4 ... lds) semantics + issue cost on gfx950, for the Winograd patch staging (conv_wino.inc):
//  (1) does the LDS destination reach beyond 64 KiB (M0 is documented with a 16-bit base)?  (2) 8-byte / 4-byte aligned global
//  sources;  (3) lanes whose offset is out of range (bit 31): zeros written?  (4) cycles a wave spends ISSUING six gather
//  instructions (32 voxels x 2 quads, 120-byte voxel pitch = the 8-channel chunk of a 30-channel NDHWC tensor) as LDS-DMA vs as
//  VGPR loads, with 4 issuing waves per workgroup and one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__global__ void sem_kernel(const float* src, int nbytes, float* out, int lds_float_off, int src_float_off) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 40960; i += 64) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lane l reads 16 B at float offset src_float_off + 30 * (l >> 1) + 4 * (l & 1); lanes 60-63 out of range
  int off = (src_float_off + 30 * (threadIdx.x >> 1) + 4 * (threadIdx.x & 1)) * 4;
  if (threadIdx.x >= 60) off |= 0x80000000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + lds_float_off), 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = lds[lds_float_off + threadIdx.x * 4 + e];
}

template <int DMA>
__global__ __launch_bounds__(512) void issue_kernel(const float* src, long nfloats, unsigned long long* cyc, float* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(nfloats * 4), 0x00020000);
  unsigned long long total = 0;
  f32x4 acc = {0, 0, 0, 0};
  if (wave >= 4) {
    const int w4 = wave - 4;
    for (int it = 0; it < reps; ++it) {
      // a 6x6x18 patch somewhere in a 48x192x192x30 tensor; chunk = it & 3
      const int tile = (blockIdx.x * 131 + it * 7) % (12 * 48 * 12);
      const int od = (tile % 12) * 4, oh = ((tile / 12) % 48) * 4, ow = (tile / 576) * 16;
      const int c0 = (it & 3) * 8;
      int offs[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int s = (j * 4 + w4) * 64 + lane, p = s >> 1, q = s & 1;
        const int v = p < 324 ? 2 * p : 2 * (p - 324) + 1;
        const int lw = v % 18, lh = (v / 18) % 6, ld = v / 108;
        const int ud = od + ld, uh = oh + lh, uw = ow + lw;
        const bool ok = p < 648 && ud < 48 && uh < 192 && uw < 192;
        offs[j] = ok ? (((ud * 192 + uh) * 192 + uw) * 30 + c0 + q * 4) * 4 : (int)0x80000000;
      }
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      if (DMA) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + ((it % 3) * 5376 + (j * 4 + w4) * 256)), 16, offs[j], 0, 0, 0);
      } else {
        f32x4 v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, offs[j], 0, 0));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        total += t1 - t0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += v[j];
      }
      if (DMA) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); total += t1 - t0; }
      __builtin_amdgcn_s_waitcnt(0);
    }
    if (lane == 0) cyc[blockIdx.x * 4 + w4] = total;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) sink[0] = lds[tid];
  }
}

int main() {
  const long n = 48L * 192 * 192 * 30;
  float* h = (float*)malloc(n * 4);
  for (long i = 0; i < n; ++i) h[i] = (float)(i % 100003);
  float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)sem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int ldsoffs[4] = {0, 15000, 17600, 36000};     // bytes 0, 60000, 70400 (> 64 KiB), 144000
  const int srcoffs[3] = {0, 2, 1};                      // 16-, 8-, 4-byte aligned sources
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 3; ++b) {
      hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 163840, 0, d, (int)(4096 * 4), o, ldsoffs[a], srcoffs[b]);
      float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
      int bad = 0, zeros = 0;
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
          const float want = l >= 60 ? 0.f : h[srcoffs[b] + 30 * (l >> 1) + 4 * (l & 1) + e];
          if (r[l * 4 + e] != want) ++bad;
          if (l >= 60 && r[l * 4 + e] == 0.f) ++zeros;
        }
      printf("lds byte offset %6d, source aligned to %2d B: %d wrong of 256 (out-of-range lanes wrote zeros: %d of 16; lane 60 got %g)\n",
             ldsoffs[a] * 4, b == 0 ? 16 : (b == 1 ? 8 : 4), bad, zeros, r[240]);
    }
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 4 * 8);
  hipFuncSetAttribute((const void*)issue_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)issue_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int reps = 200;
  for (int dma = 0; dma < 2; ++dma) {
    for (int rep = 0; rep < 2; ++rep) {
      if (dma) hipLaunchKernelGGL(issue_kernel<1>, dim3(256), dim3(512), 131072, 0, d, n, cyc, o, reps);
      else hipLaunchKernelGGL(issue_kernel<0>, dim3(256), dim3(512), 131072, 0, d, n, cyc, o, reps);
      hipDeviceSynchronize();
    }
    unsigned long long hc[1024]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)hc[i];
    printf("%s: %.0f cycles per wave to issue the 6 gathers of one patch chunk (mean over 1024 waves x %d patches)\n",
           dma ? "LDS-DMA  " : "VGPR load", s / 1024 / reps, reps);
  }
  return 0;
}
