// Does a raw buffer_load_dwordx4 that straddles num_records return the in-range dwords (per-dword range check) or zeros for all four?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* p, int nrec_bytes, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec_bytes, 0x00020000);
  const int off = threadIdx.x * 8;      // lane i reads floats 2i .. 2i+3
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
__global__ void k4(const float* p, int nrec_bytes, float* out) {     // 4-byte aligned (not 8, not 16) dwordx4 loads: rows of 47 floats
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec_bytes, 0x00020000);
  const int off = (threadIdx.x * 47 + 1) * 4;
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int nrec = 30 * 4;               // 30 floats in range
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nrec, o);
  float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 12; i < 17; ++i) printf("lane %d (floats %d..%d, num_records = 30 floats): %g %g %g %g\n", i, 2 * i, 2 * i + 3, r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
  {
    const int n = 64 * 47 + 8;
    float* hh = (float*)malloc(n * 4); for (int i = 0; i < n; ++i) hh[i] = (float)i;
    float* dd; hipMalloc(&dd, n * 4); hipMemcpy(dd, hh, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dd, n * 4, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) if (r[4 * i + e] != (float)(i * 47 + 1 + e)) ++bad;
    printf("4-byte-aligned dwordx4 loads: %d wrong values of 256 (lane 3: %g %g %g %g, expected 142..145)\n", bad, r[12], r[13], r[14], r[15]);
  }
  return 0;
}
