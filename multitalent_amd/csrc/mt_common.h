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
