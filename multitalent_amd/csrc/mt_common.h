// Shared device/host helpers for libmtseg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mtseg.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MT_WAVE 64

void mt_set_error(const char* fmt, ...);

#define MT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      mt_set_error(__VA_ARGS__);         \
      return MT_EINVAL;                  \
    }                                    \
  } while (0)

#define MT_CHECK_LAUNCH(name)                                                  \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      mt_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return MT_EHIP;                                                          \
    }                                                                          \
  } while (0)

// XCD-aware remap: hardware places block b on XCD b%8; give every XCD a contiguous range of
// logical tiles so neighbouring tiles (shared halos, shared weights) hit the same 4 MiB L2.
// Bijective for any nblk (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ int mt_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float mt_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- 16-bit storage (MT_BF16, MT_F16): widening is exact, narrowing rounds to nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).
// ST is the storage type code of the C ABI as a compile-time parameter: MT_F32 (0), MT_BF16 (1), MT_F16 (2).
typedef __bf16 mt_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 mt_f16x2 __attribute__((ext_vector_type(2)));
typedef float mt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mt_pk_bf16(float a, float b) {          // dword = (bf16(a) low half, bf16(b) high half)
  mt_f32x2 v; v[0] = a; v[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mt_bf16x2));
}
__device__ __forceinline__ float mt_bf16_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float mt_bf16_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }
__device__ __forceinline__ float mt_round_bf16(float a) { return mt_bf16_lo(mt_pk_bf16(a, a)); }     // the value a bf16 store keeps
template <int ST> __device__ __forceinline__ unsigned mt_pk16(float a, float b) {      // two values -> one dword of 16-bit elements
  static_assert(ST == MT_BF16 || ST == MT_F16, "16-bit storage types");
  if constexpr (ST == MT_BF16) return mt_pk_bf16(a, b);
  else { mt_f32x2 v; v[0] = a; v[1] = b; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mt_f16x2)); }
}
template <int ST> __device__ __forceinline__ float mt_lo16(unsigned d) {
  if constexpr (ST == MT_BF16) return mt_bf16_lo(d);
  else return (float)__builtin_bit_cast(mt_f16x2, d)[0];
}
template <int ST> __device__ __forceinline__ float mt_hi16(unsigned d) {
  if constexpr (ST == MT_BF16) return mt_bf16_hi(d);
  else return (float)__builtin_bit_cast(mt_f16x2, d)[1];
}
template <int ST> __device__ __forceinline__ float mt_from16(unsigned short h) {         // one 16-bit element (zero-extended load)
  if constexpr (ST == MT_BF16) return __builtin_bit_cast(float, (unsigned)h << 16);
  else return (float)__builtin_bit_cast(_Float16, h);
}
template <int ST> __device__ __forceinline__ float mt_round_st(float a) {                // the value a store of type ST keeps
  if constexpr (ST == MT_F32) return a;
  else return mt_lo16<ST>(mt_pk16<ST>(a, a));
}
template <int ST> __host__ __device__ constexpr int mt_ebytes() { return ST == MT_F32 ? 4 : 2; }
// scalar element access with a compile-time storage type (generic, strided kernels)
template <int ST> __device__ __forceinline__ float mt_ld(const void* p, size_t i) {
  if constexpr (ST == MT_F32) return ((const float*)p)[i];
  else return mt_from16<ST>(((const unsigned short*)p)[i]);
}
template <int ST> __device__ __forceinline__ void mt_st(void* p, size_t i, float v) {
  if constexpr (ST == MT_F32) ((float*)p)[i] = v;
  else ((unsigned short*)p)[i] = (unsigned short)(mt_pk16<ST>(v, v) & 0xffffu);
}
// the same with the type as a run-time value (slow generic paths: a uniform branch per access)
__device__ __forceinline__ float mt_ld_rt(const void* p, size_t i, int st) {
  return st == MT_F32 ? mt_ld<MT_F32>(p, i) : st == MT_BF16 ? mt_ld<MT_BF16>(p, i) : mt_ld<MT_F16>(p, i);
}
__device__ __forceinline__ void mt_st_rt(void* p, size_t i, float v, int st) {
  if (st == MT_F32) mt_st<MT_F32>(p, i, v); else if (st == MT_BF16) mt_st<MT_BF16>(p, i, v); else mt_st<MT_F16>(p, i, v);
}
__device__ __forceinline__ float mt_round_rt(float a, int st) { return st == MT_F32 ? a : st == MT_BF16 ? mt_round_st<MT_BF16>(a) : mt_round_st<MT_F16>(a); }
// VEC consecutive elements as one 4 / 8 / 16-byte access (fp32: VEC 1, 2, 4; 16-bit: VEC 2, 4, 8); idx counts vectors
template <int VEC, int ST> __device__ __forceinline__ void mt_ldv(const void* p, size_t idx, float (&v)[VEC]) {
  if constexpr (ST == MT_F32) {
    if constexpr (VEC == 1) v[0] = ((const float*)p)[idx];
    else if constexpr (VEC == 2) { const float2 t = ((const float2*)p)[idx]; v[0] = t.x; v[1] = t.y; }
    else { static_assert(VEC == 4, "fp32 vectors: 1, 2, 4"); const float4 t = ((const float4*)p)[idx]; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  } else {
    static_assert(VEC == 2 || VEC == 4 || VEC == 8, "16-bit vectors: 2, 4, 8");
    unsigned d[VEC / 2];
    if constexpr (VEC == 2) d[0] = ((const unsigned*)p)[idx];
    else if constexpr (VEC == 4) { const uint2 t = ((const uint2*)p)[idx]; d[0] = t.x; d[1] = t.y; }
    else { const uint4 t = ((const uint4*)p)[idx]; d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
#pragma unroll
    for (int k = 0; k < VEC / 2; ++k) { v[2 * k] = mt_lo16<ST>(d[k]); v[2 * k + 1] = mt_hi16<ST>(d[k]); }
  }
}
template <int VEC, int ST> __device__ __forceinline__ void mt_stv(void* p, size_t idx, const float (&v)[VEC]) {
  if constexpr (ST == MT_F32) {
    if constexpr (VEC == 1) ((float*)p)[idx] = v[0];
    else if constexpr (VEC == 2) { float2 t; t.x = v[0]; t.y = v[1]; ((float2*)p)[idx] = t; }
    else { float4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; ((float4*)p)[idx] = t; }
  } else {
    unsigned d[VEC / 2];
#pragma unroll
    for (int k = 0; k < VEC / 2; ++k) d[k] = mt_pk16<ST>(v[2 * k], v[2 * k + 1]);
    if constexpr (VEC == 2) ((unsigned*)p)[idx] = d[0];
    else if constexpr (VEC == 4) { uint2 t; t.x = d[0]; t.y = d[1]; ((uint2*)p)[idx] = t; }
    else { uint4 t; t.x = d[0]; t.y = d[1]; t.z = d[2]; t.w = d[3]; ((uint4*)p)[idx] = t; }
  }
}
static inline size_t mt_esize(int dtype) { return dtype == MT_F32 ? 4 : 2; }
static inline bool mt_dtype_ok(int dtype) { return dtype == MT_F32 || dtype == MT_BF16 || dtype == MT_F16; }
static inline bool mt_is16(int dtype) { return dtype == MT_BF16 || dtype == MT_F16; }

__device__ __forceinline__ float mt_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double mt_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

static inline int mt_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time per-DEVICE setup (hipFuncSetAttribute is a per-device property of a kernel): `done` is a bit mask over device ids.
// Returns true when the setup still has to run for the current device; the caller marks it done with mt_mark_device_done.
// Racing threads at worst repeat an idempotent call.
#include <atomic>
static inline int mt_current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : 0; }
static inline bool mt_device_pending(const std::atomic<uint64_t>& done, int dev) { return dev >= 64 || !(done.load(std::memory_order_acquire) >> dev & 1ull); }
static inline void mt_mark_device_done(std::atomic<uint64_t>& done, int dev) { if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release); }
// compute units of the current device (cached per device id)
static inline int mt_device_cus(int dev) {
  static std::atomic<int> cus[64];
  int n = dev < 64 ? cus[dev].load(std::memory_order_relaxed) : 0;
  if (n <= 0) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (n <= 0) n = 256;
    if (dev < 64) cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
