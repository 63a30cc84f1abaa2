// optim.hip — gradient-norm clipping + SGD-Nesterov over ONE flat fp32 parameter buffer.
// Reference: torch.nn.utils.clip_grad_norm_(params, 12) (MultiTalent_Trainer_DDP.py:352,362;
// nnUNetTrainerV2.py:254,263) and torch.optim.SGD(lr, weight_decay=3e-5, momentum=0.99, nesterov=True)
// (nnUNetTrainerV2.py:166-170).  The reference walks ~100 tensors twice with foreach kernels; here all
// parameters live in one contiguous buffer (also the RCCL all-reduce buffer), so clip = one
// reduction, step = one elementwise pass (3 reads + 2 writes per parameter).
#include "mt_common.h"

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  const long n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = x4[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; a += v * v; }
  a = mt_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* part, int nparts, float* out) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) a += (double)part[i];
  a = mt_wave_sum_d(a);
  if (threadIdx.x == 0) out[0] = (float)a;
}
#define SUMSQ_BLOCKS 1024
extern "C" size_t mt_sumsq_workspace(long n) { (void)n; return SUMSQ_BLOCKS * sizeof(float); }
extern "C" int mt_sumsq(const float* x, long n, float* out, void* ws, size_t ws_bytes, mt_stream_t stream) {
  MT_REQUIRE(x && out && n > 0, "sumsq: bad args");
  MT_REQUIRE(((uintptr_t)x & 15) == 0, "sumsq: buffer must be 16-byte aligned");
  if (ws == nullptr || ws_bytes < SUMSQ_BLOCKS * sizeof(float)) { mt_set_error("sumsq: workspace too small"); return MT_EWORKSPACE; }
  int blocks = mt_cdiv(n >> 2, 256 * 4); if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS; if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, (float*)ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, blocks, out);
  MT_CHECK_LAUNCH("sumsq");
  return MT_OK;
}

__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ buf, long n, float lr, float wd, float mom,
                                                           int first, const float* __restrict__ sumsq, float max_norm) {
  float coef = 1.f;
  if (sumsq != nullptr) {
    const float tn = sqrtf(sumsq[0]);
    coef = max_norm / (tn + 1e-6f);           // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6)
    coef = coef > 1.f ? 1.f : coef;           // clamped to 1.0
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pv = p[i];
    float gv = g[i] * coef;
    gv = fmaf(wd, pv, gv);                    // d_p = d_p + weight_decay * p
    float b = first ? gv : fmaf(mom, buf[i], gv);  // buf = momentum*buf + d_p (dampening 0)
    buf[i] = b;
    gv = fmaf(mom, b, gv);                    // nesterov: d_p = d_p + momentum*buf
    p[i] = pv - lr * gv;
  }
}
extern "C" int mt_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float wd, float mom, int first_step,
                               const float* sumsq_dev, float max_norm, mt_stream_t stream) {
  MT_REQUIRE(p && g && buf && n > 0, "sgd_nesterov: bad args");
  int blocks = mt_cdiv(n, 256 * 4); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, wd, mom, first_step,
                     sumsq_dev, max_norm);
  MT_CHECK_LAUNCH("sgd_nesterov");
  return MT_OK;
}
