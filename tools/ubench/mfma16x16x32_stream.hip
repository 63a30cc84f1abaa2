// What does one wave per SIMD sustain on v_mfma_f32_16x16x32_bf16 with 54 independent accumulator tiles (the shape of
// conv_bwdw_tr16_kernel's step: 108 MFMAs, operands from ds_read_b64_tr_b16, one barrier per step)?
//   V: 0 bare MFMA stream (register operands)      1 + 32 transpose reads per 108 MFMAs, two groups ahead
//      2 + one s_barrier per 108                   3 + three VALU instructions after every MFMA
//      4 = 1 + 2 + 3
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma16x16x32_stream mfma16x16x32_stream.hip ; run: ./mfma16x16x32_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 operand(const char* b, int off) {
  const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(b + off));
  const s4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(b + off + 512));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7));
}

// V bit 0: transpose reads (32 per 108 MFMAs of a wave), bit 1: barrier per 108, bit 2: 3 VALU after every MFMA, bit 3: NO MFMAs,
// bit 4: the operand as ONE ds_read_b128 instead of two transpose reads, bit 5: the reads of a step as one burst at its start.
// NW waves per SIMD (NW = 2: 27 accumulator tiles per wave, 54 MFMAs per wave and step)
template <int V, int NW>
__global__ __launch_bounds__(256 * NW) void k(float* out, int steps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256 * NW) lds[i] = 0x3f803f80u + i;
  __syncthreads();
  constexpr int NA = 54 / NW, MG = 9 / NW + (NW == 2 ? 1 : 0);
  f32x4 acc[NA];
  for (int t = 0; t < NA; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  const char* base = (const char*)lds + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8 + (wave & 3) * 2176;
  bf16x8 w[12], x[3];
  for (int i = 0; i < 12; ++i) w[i] = operand(base, 32768 + i * 1024);
  x[0] = operand(base, 0); x[1] = operand(base, 32);
  float f0 = lane, f1 = 1.f, f2 = 2.f;
  const long long t0 = __builtin_readcyclecounter();
  auto rd = [&](int off) -> bf16x8 {
    if (V & 16) return *(const bf16x8*)((const char*)lds + (threadIdx.x & 63) * 16 + (off & ~15));
    return operand(base, off);
  };
  for (int s = 0; s < steps; ++s) {
    if ((V & 1) && (V & 32)) {
      bf16x8 t[12];
#pragma unroll
      for (int g = 0; g < 12; ++g) t[g] = rd((g / 3) * 1088 + (g % 3) * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 12; ++g) { x[g % 3][0] += t[g][0]; }
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      if ((V & 1) && !(V & 32)) { x[(g + 2) % 3] = rd(((g + 2) % 12 / 3) * 1088 + ((g + 2) % 3) * 32); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int m = 0; m < (NW == 1 ? 9 : (g & 1) ? 4 : 5); ++m) {
        const int t = (g * 9 + m) % NA;
        if (!(V & 8)) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[g % 3], w[m], acc[t], 0, 0, 0);
        if (V & 4) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (V & 2) __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = f0 + f1 + f2;
  for (int t = 0; t < NA; ++t) r += acc[t][0] + acc[t][3];
  r += (float)x[0][0] + (float)x[1][0] + (float)x[2][0];
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  out[blockIdx.x * 256 * NW + threadIdx.x] = r;
}

template <int V, int NW>
void run(const char* name, float* out, long long* cyc) {
  const int steps = 2000;
  hipFuncSetAttribute((const void*)k<V, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, NW><<<256, 256 * NW, 65536>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<V, NW><<<256, 256 * NW, 65536>>>(out, steps, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c[4]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const double flop = (V & 8) ? 0.0 : 256.0 * 4 * steps * 108 * 16384.0;
  printf("%-58s %7.3f ms  %6.0f TFLOP/s (%.2f of 2500)  %7.0f ticks per step of a SIMD (108 MFMAs = 1728)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500,
         (double)c[0] / steps);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  printf("one wave per SIMD (108 MFMAs per wave and step)\n");
  run<0, 1>("bare stream, 54 accumulator tiles", out, cyc);
  run<1, 1>("+ 32 transpose reads per step, two groups ahead", out, cyc);
  run<1 | 32, 1>("+ 24 transpose reads per step as one burst", out, cyc);
  run<1 | 16, 1>("+ 16 ds_read_b128 per step, two groups ahead", out, cyc);
  run<2, 1>("+ barrier per step", out, cyc);
  run<4, 1>("+ 3 VALU after every MFMA", out, cyc);
  run<7, 1>("+ reads + barrier + VALU", out, cyc);
  run<8 | 1, 1>("transpose reads alone (no MFMA)", out, cyc);
  run<8 | 4, 1>("VALU alone (no MFMA)", out, cyc);
  printf("two waves per SIMD (54 MFMAs per wave and step, each wave reads all 12 operands)\n");
  run<0, 2>("bare stream, 27 accumulator tiles per wave", out, cyc);
  run<1, 2>("+ 32 transpose reads per wave and step", out, cyc);
  run<4, 2>("+ 3 VALU after every MFMA", out, cyc);
  run<5, 2>("+ reads + VALU", out, cyc);
  run<7, 2>("+ reads + barrier + VALU", out, cyc);
  return 0;
}
