#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o, int nbytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  int off = threadIdx.x * 120 + 8;   // like voxel stride 120 B, channel 2
  if (threadIdx.x == 5) off = (int)0x80000000 + 64;
  auto t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  float2 f2 = __builtin_bit_cast(float2, t); o[threadIdx.x * 2] = f2.x;
  o[threadIdx.x * 2 + 1] = f2.y;
}
int main() {
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 8); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024);
  float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 10; ++i) printf("lane %d: %g %g (expect %d %d)\n", i, r[2 * i], r[2 * i + 1], i * 30 + 2, i * 30 + 3);
  return 0;
}
