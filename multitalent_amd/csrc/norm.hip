// norm.hip — InstanceNorm3d(eps, affine) + LeakyReLU forward finalisation / materialisation and the
// fused backward, channel reductions, layout transposes.  All HBM-bound: one pass per tensor,
// "channel-lane" mapping (32 consecutive lanes = 32 consecutive channels of one voxel, 128 B
// coalesced), fp32 per-thread partial sums combined in fp64.
//
// Reference: nn.InstanceNorm3d(eps=1e-5, affine=True) + nn.LeakyReLU(1e-2, inplace) inside
// ConvDropoutNormNonlin (generic_UNet.py:63-64,69-70; nnUNetTrainerV2.py:152-155), biased variance.
#include "mt_common.h"
#include <initializer_list>

// ---- finalize: partial (sum,sumsq) -> mean, rstd, scale, shift ---------------------------------
// one block per (n, c), 256 threads walking the partial blocks (the 64-thread form needed 108 dependent-latency iterations for the
// 6912 tiles of a full-resolution layer: 16 us on the critical path between every conv and its consumer); double, fixed order
__global__ __launch_bounds__(256) void inorm_finalize_kernel(const float* __restrict__ part, int nsb, int C, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float* mean, float* rstd, float* scale, float* shift) {
  __shared__ double red[4][2];
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  const float2* p = (const float2*)part + (size_t)n * nsb * C + c;
  double s1 = 0.0, s2 = 0.0;
  for (int s = threadIdx.x; s < nsb; s += 256) {
    const float2 v = p[(size_t)s * C];
    s1 += (double)v.x; s2 += (double)v.y;
  }
  s1 = mt_wave_sum_d(s1);
  s2 = mt_wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double rs = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    const size_t o = (size_t)n * C + c;
    mean[o] = (float)m;
    rstd[o] = (float)rs;
    scale[o] = (float)(g * rs);
    shift[o] = (float)(b - m * g * rs);
  }
}

extern "C" int mt_inorm_finalize(const float* part, int N, int nsb, int C, double count, const float* gamma,
                                 const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                                 mt_stream_t stream) {
  MT_REQUIRE(part && mean && rstd && scale && shift && N > 0 && C > 0 && nsb > 0 && count > 0, "inorm_finalize: bad args");
  hipLaunchKernelGGL(inorm_finalize_kernel, dim3(N * C), dim3(256), 0, (hipStream_t)stream, part, nsb, C, count, gamma, beta,
                     eps, mean, rstd, scale, shift);
  MT_CHECK_LAUNCH("inorm_finalize");
  return MT_OK;
}

// ---- generic voxel-block geometry for channel-lane kernels -------------------------------------
// grid = (nvb, N); block 256 = 8 voxel rows x 32 channel lanes; block covers VB voxels.
// Voxels per workgroup of the streaming kernels below (one definition for host and device: the partial-sum buffers are indexed by
// it).  2048 at full resolution (1 728 workgroups for 2 x 48 x 192 x 192); a fixed 2048 left the mid-resolution layers with ~200
// workgroups of 256 threads on 256 CUs — 2.9 TB/s where the full-resolution launches reach 5.4 — so smaller tensors get
// proportionally smaller blocks (>= 128 voxels, about 512 blocks per sample).
#ifndef MT_VB_BLOCKS
#define MT_VB_BLOCKS 512      // target workgroups per sample below full resolution (256 / 1024 measured no better: tools/build_norm_variant.sh)
#endif
#ifndef MT_VB_MIN
#define MT_VB_MIN 32
#endif
__host__ __device__ static inline int mt_vb(long V) {
  if (V >= 2048L * MT_VB_BLOCKS) return 2048;
  long vb = ((V + MT_VB_BLOCKS - 1) / MT_VB_BLOCKS + 31) / 32 * 32;
  // (the low-resolution tensors, V < 64 K voxels: MT_VB_MIN voxels per workgroup — at 128 a 6 x 12 x 12 tensor was 7 workgroups per
  // sample walking ten dependent load rounds each)
  return vb < MT_VB_MIN ? MT_VB_MIN : (int)vb;
}
static inline int nb_blocks(long V) { return mt_cdiv(V, (long)mt_vb(V)); }

// ---- storage types ---------------------------------------------------------------------------------
// Every streaming kernel below takes the storage type of its ACTIVATION operands (AT: y, residual, materialised output) and of its
// GRADIENT operands (GT: g, copies of g) — mt_common.h.  The dense fast paths are compiled for the combinations the engine produces
// (fp32 / fp32, fp16 activations with bf16 gradients, bf16 / bf16); anything else takes the strided kernels, which read the type at
// run time.  Arithmetic and partial sums are fp32 / fp64 either way.
#define MT_AG_SWITCH(at_, gt_, CALL_, ELSE_)                                                                                  \
  do {                                                                                                                        \
    if ((at_) == MT_F32 && (gt_) == MT_F32) { constexpr int AT = MT_F32, GT = MT_F32; CALL_; }                                \
    else if ((at_) == MT_F16 && (gt_) == MT_BF16) { constexpr int AT = MT_F16, GT = MT_BF16; CALL_; }                         \
    else if ((at_) == MT_BF16 && (gt_) == MT_BF16) { constexpr int AT = MT_BF16, GT = MT_BF16; CALL_; }                       \
    else { ELSE_; }                                                                                                           \
  } while (0)
#define MT_A_SWITCH(at_, CALL_)                                                                                               \
  do {                                                                                                                        \
    if ((at_) == MT_F32) { constexpr int AT = MT_F32; CALL_; }                                                                \
    else if ((at_) == MT_F16) { constexpr int AT = MT_F16; CALL_; }                                                           \
    else { constexpr int AT = MT_BF16; CALL_; }                                                                               \
  } while (0)
// widest vector (elements) the dense fast paths may use: divides C, keeps every sample base aligned, <= 16 bytes
static int dense_vec(int C, long per_sample, int dtype, std::initializer_list<const void*> ptrs) {
  const int es = (int)mt_esize(dtype);
  int v = 16 / es;                                   // 4 (fp32) or 8 (bf16)
  while (v > 1 && (C % v || per_sample % v)) v >>= 1;
  for (const void* q : ptrs)
    if (q != nullptr) { while (v > 1 && (((uintptr_t)q) & (size_t)(v * es - 1))) v >>= 1; }
  if (mt_is16(dtype) && v < 2) return 0;             // 16-bit vectors are at least one dword
  return v;
}

// ---- materialise a = lrelu(y*sc+sh [+ residual]) ----------------------------------------------
struct ApplyParams {
  const void* y; int ycs; const float* scale; const float* shift; float slope;
  const void* res; int rcs; const float* rscale; const float* rshift; float rslope;
  void* out; int ocs; long V; int C;
};
__global__ __launch_bounds__(256) void inorm_apply_kernel(const ApplyParams P, int at) {
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  const int vb = mt_vb(P.V);
  const long v0 = (long)blockIdx.x * vb;
  const long v1 = (v0 + vb < P.V) ? v0 + vb : P.V;
  for (int cb = 0; cb < P.C; cb += 32) {
    const int c = cb + cl;
    if (c >= P.C) continue;
    const float sc = P.scale ? P.scale[(size_t)n * P.C + c] : 1.f, sh = P.scale ? P.shift[(size_t)n * P.C + c] : 0.f;
    const float rsc = (P.res && P.rscale) ? P.rscale[(size_t)n * P.C + c] : 1.f;
    const float rsh = (P.res && P.rscale) ? P.rshift[(size_t)n * P.C + c] : 0.f;
    for (long v = v0 + vr; v < v1; v += 8) {
      const size_t e = (size_t)n * P.V + v;
      float t = fmaf(mt_ld_rt(P.y, e * P.ycs + c, at), sc, sh);
      if (P.res) t += mt_lrelu(fmaf(mt_ld_rt(P.res, e * P.rcs + c, at), rsc, rsh), P.rslope);
      mt_st_rt(P.out, e * P.ocs + c, mt_lrelu(t, P.slope), at);
    }
  }
}
static int launch_apply_fast(const ApplyParams& P, int N, int dtype, int vec, mt_stream_t stream);
extern "C" int mt_inorm_lrelu_apply(const float* y, int ycs, const float* scale, const float* shift, float slope,
                                    const float* res, int rcs, const float* rscale, const float* rshift, float rslope,
                                    float* out, int ocs, int N, long V, int C, int dtype, mt_stream_t stream) {
  MT_REQUIRE(y && out && N > 0 && V > 0 && C > 0 && mt_dtype_ok(dtype), "inorm_lrelu_apply: bad args");
  ApplyParams P{y, ycs, scale, shift, slope, res, rcs, rscale, rshift, rslope, out, ocs, V, C};
  if (ycs == C && ocs == C && (res == nullptr || rcs == C)) {
    const int vec = dense_vec(C, V * C, dtype, {y, out, res});
    if (vec > 0 && C / vec <= 256) return launch_apply_fast(P, N, dtype, vec, stream);
  }
  hipLaunchKernelGGL(inorm_apply_kernel, dim3(nb_blocks(V), N), dim3(256), 0, (hipStream_t)stream, P, dtype);
  MT_CHECK_LAUNCH("inorm_lrelu_apply");
  return MT_OK;
}

// ---- backward of out = lrelu(IN(y)) -------------------------------------------------------------
// pass 1: per (n,c) A = sum dz, B = sum dz*zhat   (dz = g * lrelu'(z)); pass 2: dy in place.
struct InBwdParams {
  void* g; int gcs; const void* y; int ycs;
  const float* mean; const float* rstd; const float* gamma; const float* beta; float slope;
  long V; int C; int nvb;
  float* part1;  // [N][nvb][C][2]
  float* m;      // [N][C][2]  (A/V, B/V)
  float* part2;  // [N][nvb][C]
  // first-pass partials as the finalize kernel reads them: [N][p1_nblk][p1_cs][2], channel c at column p1_c0 + c (own reduce
  // pass: part1, nvb, C, 0; fused into the producing convolution: its stats_part, its nsb, its Cout, bstats.c0)
  const float* p1; int p1_nblk, p1_cs, p1_c0;
  int gt, at;    // storage types of g and y
};
__global__ __launch_bounds__(256) void inorm_bwd_reduce_kernel(const InBwdParams P) {
  __shared__ float red[8][32][2];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  const int vb = mt_vb(P.V);
  const long v0 = (long)blockIdx.x * vb;
  const long v1 = (v0 + vb < P.V) ? v0 + vb : P.V;
  for (int cb = 0; cb < P.C; cb += 32) {
    const int c = cb + cl;
    float a = 0.f, b = 0.f;
    if (c < P.C) {
      const float mu = P.mean[(size_t)n * P.C + c], rs = P.rstd[(size_t)n * P.C + c];
      const float ga = P.gamma ? P.gamma[c] : 1.f, be = P.beta ? P.beta[c] : 0.f;
      for (long v = v0 + vr; v < v1; v += 8) {
        const size_t e = (size_t)n * P.V + v;
        const float zh = (mt_ld_rt(P.y, e * P.ycs + c, P.at) - mu) * rs;
        const float z = fmaf(zh, ga, be);
        float dz = mt_ld_rt(P.g, e * P.gcs + c, P.gt);
        dz = z > 0.f ? dz : dz * P.slope;
        a += dz;
        b += dz * zh;
      }
    }
    red[vr][cl][0] = a; red[vr][cl][1] = b;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int k = threadIdx.x & 1, cc = threadIdx.x >> 1;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r][cc][k];
      if (cb + cc < P.C) P.part1[(((size_t)n * P.nvb + blockIdx.x) * P.C + cb + cc) * 2 + k] = s;
    }
    __syncthreads();
  }
}
// finalize pass 1: m[n][c] = (A/V, B/V); dgamma[c] += sum_n B; dbeta[c] += sum_n A
__global__ __launch_bounds__(256) void inorm_bwd_finalize_kernel(const InBwdParams P, int N, float* dgamma, float* dbeta) {
  // one workgroup per channel, four waves over the blocks of a sample (a chain of dependent loads per thread: 54 deep with one
  // wave at full resolution); the waves' sums are combined in a fixed order
  __shared__ double red[4][2];
  const int c = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  double ta = 0.0, tb = 0.0;
  for (int n = 0; n < N; ++n) {
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < P.p1_nblk; s += 256) {
      const float* q = P.p1 + (((size_t)n * P.p1_nblk + s) * P.p1_cs + P.p1_c0 + c) * 2;
      a += (double)q[0];
      b += (double)q[1];
    }
    a = mt_wave_sum_d(a); b = mt_wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) { red[wave][0] = a; red[wave][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      b = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      P.m[((size_t)n * P.C + c) * 2] = (float)(a / (double)P.V);
      P.m[((size_t)n * P.C + c) * 2 + 1] = (float)(b / (double)P.V);
      ta += a; tb += b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] += (float)tb;
    if (dbeta) dbeta[c] += (float)ta;
  }
}
__global__ __launch_bounds__(256) void inorm_bwd_apply_kernel(const InBwdParams P) {
  __shared__ float red[8][32];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  const int vb = mt_vb(P.V);
  const long v0 = (long)blockIdx.x * vb;
  const long v1 = (v0 + vb < P.V) ? v0 + vb : P.V;
  for (int cb = 0; cb < P.C; cb += 32) {
    const int c = cb + cl;
    float sdy = 0.f;
    if (c < P.C) {
      const float mu = P.mean[(size_t)n * P.C + c], rs = P.rstd[(size_t)n * P.C + c];
      const float ga = P.gamma ? P.gamma[c] : 1.f, be = P.beta ? P.beta[c] : 0.f;
      const float m1 = P.m[((size_t)n * P.C + c) * 2], m2 = P.m[((size_t)n * P.C + c) * 2 + 1];
      const float k = ga * rs;
      for (long v = v0 + vr; v < v1; v += 8) {
        const size_t e = (size_t)n * P.V + v;
        const float zh = (mt_ld_rt(P.y, e * P.ycs + c, P.at) - mu) * rs;
        const float z = fmaf(zh, ga, be);
        float dz = mt_ld_rt(P.g, e * P.gcs + c, P.gt);
        dz = z > 0.f ? dz : dz * P.slope;
        const float dy = mt_round_rt(k * (dz - m1 - zh * m2), P.gt);       // the bias gradient sums what the next kernel reads
        mt_st_rt(P.g, e * P.gcs + c, dy, P.gt);
        sdy += dy;
      }
    }
    if (P.part2) {
      red[vr][cl] = sdy;
      __syncthreads();
      if (threadIdx.x < 32) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
        if (cb + threadIdx.x < P.C) P.part2[((size_t)n * P.nvb + blockIdx.x) * P.C + cb + threadIdx.x] = s;
      }
      __syncthreads();
    }
  }
}
// out[c] (+)= sum over rows of part[rows][C]
__global__ __launch_bounds__(64) void colsum_kernel(const float* part, long rows, int C, float* out, int accumulate) {
  const int c = blockIdx.x;
  double a = 0.0;
  for (long s = threadIdx.x; s < rows; s += 64) a += (double)part[(size_t)s * C + c];
  a = mt_wave_sum_d(a);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)a : (float)a;
}


// ---- fast path for CONTIGUOUS tensors (cs == C): the whole [V][C] slab of a sample is one linear array.  A thread owns a
// fixed group of VEC consecutive channels: with G = C/VEC groups, thread t < A = (256/G)*G handles flat vector index
// base + t + k*A, whose channel group is (t mod G) for every k (A is a multiple of G) -> per-channel constants live in
// registers, consecutive lanes read consecutive vectors (fully coalesced), 4 independent loads in flight per tensor.
#define NF_UNROLL 4

struct InBwdFast {
  void* g; const void* y;
  const float* mean; const float* rstd; const float* gamma; const float* beta; float slope;
  long nvec;      // vectors per sample = V*C/VEC
  int C, G, A, nblk;
  float* part1;   // [N][nblk][C][2]
  const float* m; // [N][C][2]
  float* part2;   // [N][nblk][C] or null
};

template <int VEC, bool APPLY, int AT, int GT>
__global__ __launch_bounds__(256) void inorm_bwd_fast_kernel(const InBwdFast P) {
  static_assert(mt_ebytes<AT>() == mt_ebytes<GT>(), "one vector index addresses both tensors");
  __shared__ float red[256 * VEC * 2];
  const int n = blockIdx.y, t = threadIdx.x;
  const bool act = t < P.A;
  const int grp = t % P.G;
  float mu[VEC], rs[VEC], ga[VEC], be[VEC], m1[VEC], m2[VEC], a0[VEC], a1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int c = grp * VEC + e;
    mu[e] = P.mean[(size_t)n * P.C + c]; rs[e] = P.rstd[(size_t)n * P.C + c];
    ga[e] = P.gamma ? P.gamma[c] : 1.f; be[e] = P.beta ? P.beta[c] : 0.f;
    if (APPLY) { m1[e] = P.m[((size_t)n * P.C + c) * 2]; m2[e] = P.m[((size_t)n * P.C + c) * 2 + 1]; }
    a0[e] = 0.f; a1[e] = 0.f;
  }
  // block range: vectors [blk*per, (blk+1)*per), per a multiple of A
  const long per = ((P.nvec + P.nblk - 1) / P.nblk + P.A - 1) / P.A * P.A;
  const long lo = (long)blockIdx.x * per;
  long hi = lo + per; if (hi > P.nvec) hi = P.nvec;
  const size_t sbytes = (size_t)n * P.nvec * VEC * mt_ebytes<GT>();
  char* gp = (char*)P.g + sbytes;
  const char* yp = (const char*)P.y + sbytes;
  if (act) {
    for (long i0 = lo + t; i0 < hi; i0 += (long)P.A * NF_UNROLL) {
      float gv[NF_UNROLL][VEC], yv[NF_UNROLL][VEC];
#pragma unroll
      for (int u = 0; u < NF_UNROLL; ++u) {
        const long i = i0 + (long)u * P.A;
        if (i < hi) { mt_ldv<VEC, GT>(gp, (size_t)i, gv[u]); mt_ldv<VEC, AT>(yp, (size_t)i, yv[u]); }
      }
#pragma unroll
      for (int u = 0; u < NF_UNROLL; ++u) {
        const long i = i0 + (long)u * P.A;
        if (i < hi) {
          float out[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float zh = (yv[u][e] - mu[e]) * rs[e];
            const float z = fmaf(zh, ga[e], be[e]);
            float dz = gv[u][e];
            dz = z > 0.f ? dz : dz * P.slope;
            if (APPLY) {
              const float dy = mt_round_st<GT>(ga[e] * rs[e] * (dz - m1[e] - zh * m2[e]));
              out[e] = dy;
              a0[e] += dy;
            } else {
              a0[e] += dz;
              a1[e] = fmaf(dz, zh, a1[e]);
            }
          }
          if (APPLY) mt_stv<VEC, GT>(gp, (size_t)i, out);
        }
      }
    }
  }
  if (APPLY && P.part2 == nullptr) return;
  // reduce over the threads that share a channel group: threads t, t+G, t+2G, ...
#pragma unroll
  for (int e = 0; e < VEC; ++e) { red[(t * VEC + e) * 2] = act ? a0[e] : 0.f; red[(t * VEC + e) * 2 + 1] = act ? a1[e] : 0.f; }
  __syncthreads();
  if (t < P.G) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float s0 = 0.f, s1 = 0.f;
      for (int k = t; k < P.A; k += P.G) { s0 += red[(k * VEC + e) * 2]; s1 += red[(k * VEC + e) * 2 + 1]; }
      const int c = t * VEC + e;
      if (APPLY) P.part2[((size_t)n * P.nblk + blockIdx.x) * P.C + c] = s0;
      else { float* o = P.part1 + (((size_t)n * P.nblk + blockIdx.x) * P.C + c) * 2; o[0] = s0; o[1] = s1; }
    }
  }
}

// ---- small tensors (round 4): the low-resolution layers (<= 16 K voxels over the batch) are three latency-bound launches of ~15 us each
// (reduce, finalize, apply) + the dbias column sum — 37 layers of the residual encoder, 1.2 ms of its 26-ms mixed step.  Here ONE
// launch: a workgroup owns a group of VEC channels of ALL samples, so the per-(n, c) sums never leave it: pass 1 over its slab ->
// block reduction (wave sums, then the waves in a fixed order, in double) -> m -> pass 2 re-reads the slab (L2) and applies; dgamma,
// dbeta (+=) and dbias (=) from the same workgroup.  Same arithmetic per element as inorm_bwd_fast_kernel.
struct InBwdSmall {
  void* g; const void* y;
  const float* mean; const float* rstd; const float* gamma; const float* beta; float slope;
  long V; int C, N;
  float* dgamma; float* dbeta; float* dbias;
};
// 256 .. 1024 threads (round 5): a sample's voxels in as few dependent load rounds as possible — at 256 threads a 6 x 12 x 12 tensor
// was four rounds per pass, two passes, per sample
template <int VEC, int AT, int GT>
__global__ __launch_bounds__(1024) void inorm_bwd_small_kernel(const InBwdSmall P) {
  static_assert(mt_ebytes<AT>() == mt_ebytes<GT>(), "one vector index addresses both tensors");
  __shared__ double redd[16][2 * VEC];
  __shared__ float msh[2 * VEC];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int NT = blockDim.x, NWV = NT >> 6;
  const int grp = blockIdx.x, G = P.C / VEC;
  float ga[VEC], be[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { const int c = grp * VEC + e; ga[e] = P.gamma ? P.gamma[c] : 1.f; be[e] = P.beta ? P.beta[c] : 0.f; }
  double tA = 0.0, tB = 0.0, tD = 0.0;         // thread e < VEC: sums over the samples of channel grp * VEC + e
  for (int n = 0; n < P.N; ++n) {
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const int c = grp * VEC + e; mu[e] = P.mean[(size_t)n * P.C + c]; rs[e] = P.rstd[(size_t)n * P.C + c]; }
    const size_t sbytes = (size_t)n * P.V * P.C * mt_ebytes<GT>();
    char* gp = (char*)P.g + sbytes;
    const char* yp = (const char*)P.y + sbytes;
    // ---- pass 1
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    for (long v = t; v < P.V; v += NT) {
      float gv[VEC], yv[VEC];
      mt_ldv<VEC, GT>(gp, (size_t)(v * G + grp), gv);
      mt_ldv<VEC, AT>(yp, (size_t)(v * G + grp), yv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float zh = (yv[e] - mu[e]) * rs[e];
        const float z = fmaf(zh, ga[e], be[e]);
        float dz = gv[e];
        dz = z > 0.f ? dz : dz * P.slope;
        a0[e] += dz;
        a1[e] = fmaf(dz, zh, a1[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float s0 = mt_wave_sum(a0[e]), s1 = mt_wave_sum(a1[e]);
      if (lane == 0) { redd[wave][2 * e] = (double)s0; redd[wave][2 * e + 1] = (double)s1; }
    }
    __syncthreads();
    if (t < 2 * VEC) {
      double s = 0.0;
      for (int w = 0; w < NWV; ++w) s += redd[w][t];           // fixed order
      msh[t] = (float)(s / (double)P.V);
      if (t & 1) tB += s; else tA += s;         // thread 2e: A of channel e, thread 2e + 1: B
    }
    __syncthreads();
    float m1[VEC], m2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { m1[e] = msh[2 * e]; m2[e] = msh[2 * e + 1]; }
    // ---- pass 2
    float d0[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) d0[e] = 0.f;
    for (long v = t; v < P.V; v += NT) {
      float gv[VEC], yv[VEC], out[VEC];
      mt_ldv<VEC, GT>(gp, (size_t)(v * G + grp), gv);
      mt_ldv<VEC, AT>(yp, (size_t)(v * G + grp), yv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float zh = (yv[e] - mu[e]) * rs[e];
        const float z = fmaf(zh, ga[e], be[e]);
        float dz = gv[e];
        dz = z > 0.f ? dz : dz * P.slope;
        const float dy = mt_round_st<GT>(ga[e] * rs[e] * (dz - m1[e] - zh * m2[e]));
        out[e] = dy;
        d0[e] += dy;
      }
      mt_stv<VEC, GT>(gp, (size_t)(v * G + grp), out);
    }
    if (P.dbias != nullptr) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float s0 = mt_wave_sum(d0[e]);
        if (lane == 0) redd[wave][e] = (double)s0;
      }
      __syncthreads();
      if (t < VEC) { double s = 0.0; for (int w = 0; w < NWV; ++w) s += redd[w][t]; tD += s; }
    }
    __syncthreads();                             // redd / msh are rewritten for the next sample
  }
  if (t < 2 * VEC) {
    const int c = grp * VEC + (t >> 1);
    if (t & 1) { if (P.dgamma) P.dgamma[c] += (float)tB; }
    else if (P.dbeta) P.dbeta[c] += (float)tA;
  }
  if (t < VEC && P.dbias != nullptr) P.dbias[grp * VEC + t] = (float)tD;
}
#define INORM_SMALL_MAX 2048         // voxels over the batch.  (16384 through round 4: at 2 x 12x24x24 x 256 the one-launch kernel — C / 8 workgroups
                                     // streaming the whole tensor — takes 115 us where the three launches take 31, tools/bench_norm_small.py; at 2 x 6x12x12 they tie,
                                     // at 2 x 3x6x6 it is 11 against 20 us)
extern "C" size_t mt_inorm_bwd_workspace(int N, long V, int C) {
  const size_t nvb = (size_t)nb_blocks(V);
  return ((size_t)N * nvb * C * 3 + (size_t)N * C * 2) * sizeof(float);
}
template <int VEC, int AT, int GT>
static void launch_inorm_bwd_fast(const InBwdFast& F, dim3 grid, bool reduce, hipStream_t st) {
  if (reduce) hipLaunchKernelGGL((inorm_bwd_fast_kernel<VEC, false, AT, GT>), grid, dim3(256), 0, st, F);
  else hipLaunchKernelGGL((inorm_bwd_fast_kernel<VEC, true, AT, GT>), grid, dim3(256), 0, st, F);
}
static bool ag_fast(int at, int gt) { return (at == MT_F32 && gt == MT_F32) || (at == MT_F16 && gt == MT_BF16) || (at == MT_BF16 && gt == MT_BF16); }
static void inorm_bwd_fast_dispatch(const InBwdFast& F, dim3 grid, int vec, int at, int gt, bool reduce, hipStream_t st) {
  if (at == MT_F32) {
    if (vec == 4) launch_inorm_bwd_fast<4, MT_F32, MT_F32>(F, grid, reduce, st);
    else if (vec == 2) launch_inorm_bwd_fast<2, MT_F32, MT_F32>(F, grid, reduce, st);
    else launch_inorm_bwd_fast<1, MT_F32, MT_F32>(F, grid, reduce, st);
  } else if (at == MT_F16) {
    if (vec == 8) launch_inorm_bwd_fast<8, MT_F16, MT_BF16>(F, grid, reduce, st);
    else if (vec == 4) launch_inorm_bwd_fast<4, MT_F16, MT_BF16>(F, grid, reduce, st);
    else launch_inorm_bwd_fast<2, MT_F16, MT_BF16>(F, grid, reduce, st);
  } else {
    if (vec == 8) launch_inorm_bwd_fast<8, MT_BF16, MT_BF16>(F, grid, reduce, st);
    else if (vec == 4) launch_inorm_bwd_fast<4, MT_BF16, MT_BF16>(F, grid, reduce, st);
    else launch_inorm_bwd_fast<2, MT_BF16, MT_BF16>(F, grid, reduce, st);
  }
}
extern "C" int mt_inorm_lrelu_bwd(float* g, int gcs, const float* y, int ycs, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float slope, int N, long V, int C,
                                  float* dgamma, float* dbeta, float* dbias, const float* part, int part_nblk, int part_cs, int part_c0,
                                  void* ws, size_t ws_bytes, int gdtype, int ydtype, mt_stream_t stream) {
  MT_REQUIRE(g && y && mean && rstd && N > 0 && V > 0 && C > 0 && mt_dtype_ok(gdtype) && mt_dtype_ok(ydtype), "inorm_lrelu_bwd: bad args");
  MT_REQUIRE(part == nullptr || (part_nblk > 0 && part_cs >= part_c0 + C && part_c0 >= 0), "inorm_lrelu_bwd: bad external partials");
  if (ws == nullptr || ws_bytes < mt_inorm_bwd_workspace(N, V, C)) { mt_set_error("inorm_lrelu_bwd: workspace too small"); return MT_EWORKSPACE; }
  InBwdParams P;
  P.g = g; P.gcs = gcs; P.y = y; P.ycs = ycs; P.mean = mean; P.rstd = rstd; P.gamma = gamma; P.beta = beta; P.slope = slope;
  P.V = V; P.C = C; P.nvb = nb_blocks(V);
  float* w = (float*)ws;
  P.part1 = w; w += (size_t)N * P.nvb * C * 2;
  P.m = w; w += (size_t)N * C * 2;
  P.part2 = dbias ? w : nullptr;
  P.p1 = part ? part : P.part1; P.p1_nblk = part ? part_nblk : P.nvb; P.p1_cs = part ? part_cs : C; P.p1_c0 = part ? part_c0 : 0;
  P.gt = gdtype; P.at = ydtype;
  hipStream_t st = (hipStream_t)stream;
  // contiguous tensors take the vectorised lane-constant-channel path
  const int vec = (gcs == C && ycs == C && ag_fast(ydtype, gdtype)) ? dense_vec(C, V * C, gdtype, {g, y}) : 0;
  const bool contig = vec > 0 && (C / vec <= 256);
  constexpr int small_on = 1;
  const int svec = gdtype == MT_F32 ? 4 : 8;
  if (small_on && contig && part == nullptr && vec == svec && (long)N * V <= INORM_SMALL_MAX) {
    InBwdSmall S;
    S.g = g; S.y = y; S.mean = mean; S.rstd = rstd; S.gamma = gamma; S.beta = beta; S.slope = slope; S.V = V; S.C = C; S.N = N;
    S.dgamma = dgamma; S.dbeta = dbeta; S.dbias = dbias;
    const dim3 grid(C / svec);
    const dim3 blk(V > 768 ? 1024 : V > 512 ? 768 : V > 256 ? 512 : 256);
    if (gdtype == MT_F32) hipLaunchKernelGGL((inorm_bwd_small_kernel<4, MT_F32, MT_F32>), grid, blk, 0, st, S);
    else if (ydtype == MT_F16) hipLaunchKernelGGL((inorm_bwd_small_kernel<8, MT_F16, MT_BF16>), grid, blk, 0, st, S);
    else hipLaunchKernelGGL((inorm_bwd_small_kernel<8, MT_BF16, MT_BF16>), grid, blk, 0, st, S);
    MT_CHECK_LAUNCH("inorm_lrelu_bwd(small)");
    return MT_OK;
  }
  if (contig) {
    InBwdFast F;
    F.g = g; F.y = y; F.mean = mean; F.rstd = rstd; F.gamma = gamma; F.beta = beta; F.slope = slope;
    F.nvec = (long)V * C / vec; F.C = C; F.G = C / vec; F.A = (256 / F.G) * F.G; F.nblk = P.nvb;
    F.part1 = P.part1; F.m = P.m; F.part2 = P.part2;
    dim3 grid(P.nvb, N);
    if (part == nullptr) inorm_bwd_fast_dispatch(F, grid, vec, ydtype, gdtype, true, st);   // (the first pass was not fused into the kernel that produced g)
    hipLaunchKernelGGL(inorm_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, P, N, dgamma, dbeta);
    inorm_bwd_fast_dispatch(F, grid, vec, ydtype, gdtype, false, st);
  } else {
    if (part == nullptr) hipLaunchKernelGGL(inorm_bwd_reduce_kernel, dim3(P.nvb, N), dim3(256), 0, st, P);
    hipLaunchKernelGGL(inorm_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, P, N, dgamma, dbeta);
    hipLaunchKernelGGL(inorm_bwd_apply_kernel, dim3(P.nvb, N), dim3(256), 0, st, P);
  }
  if (dbias) hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(64), 0, st, (const float*)P.part2, (long)N * P.nvb, C, dbias, 0);
  MT_CHECK_LAUNCH("inorm_lrelu_bwd");
  return MT_OK;
}

// ---- g *= lrelu'(t), t = y*sc+sh (+ second term) ; optional copy --------------------------------
struct LBwdParams {
  void* g; int gcs; const void* y; int ycs; const float* scale; const float* shift; float slope;
  const void* y2; int y2cs; const float* scale2; const float* shift2; float slope2;
  void* gcopy; int gcopycs; long V; int C;
  int gt, at;   // storage types of (g, gcopy) and of (y, y2)
};
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const LBwdParams P) {
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  const int vb = mt_vb(P.V);
  const long v0 = (long)blockIdx.x * vb;
  const long v1 = (v0 + vb < P.V) ? v0 + vb : P.V;
  for (int cb = 0; cb < P.C; cb += 32) {
    const int c = cb + cl;
    if (c >= P.C) continue;
    const float sc = P.scale ? P.scale[(size_t)n * P.C + c] : 1.f, sh = P.scale ? P.shift[(size_t)n * P.C + c] : 0.f;
    const float sc2 = (P.y2 && P.scale2) ? P.scale2[(size_t)n * P.C + c] : 1.f;
    const float sh2 = (P.y2 && P.scale2) ? P.shift2[(size_t)n * P.C + c] : 0.f;
    for (long v = v0 + vr; v < v1; v += 8) {
      const size_t e = (size_t)n * P.V + v;
      float t = fmaf(mt_ld_rt(P.y, e * P.ycs + c, P.at), sc, sh);
      if (P.y2) t += mt_lrelu(fmaf(mt_ld_rt(P.y2, e * P.y2cs + c, P.at), sc2, sh2), P.slope2);
      float gv = mt_ld_rt(P.g, e * P.gcs + c, P.gt);
      gv = t > 0.f ? gv : gv * P.slope;
      mt_st_rt(P.g, e * P.gcs + c, gv, P.gt);
      if (P.gcopy) mt_st_rt(P.gcopy, e * P.gcopycs + c, gv, P.gt);
    }
  }
}

// ---- dense fast paths of the two element-wise kernels above (every operand contiguous, cs == C, C % VEC == 0, C / VEC <= 256): the
// mapping of inorm_bwd_fast_kernel — thread t keeps the VEC channels of group t % G for all its voxels (A = (256 / G) G active
// threads, a block's range a multiple of A), so the per-(n, c) scale / shift are registers loaded once and an iteration is NF_UNROLL
// independent 16 / 8 / 4-byte accesses per stream.  (Through round 4 a thread walked the flat element index: one 64-bit modulo and
// up to 4 VEC table loads through the vector memory path per iteration — 2.8 TB/s at 30 channels where inorm_bwd_fast_kernel moves
// 5.4.)  HBM-bound: apply = 2-3 streams, lrelu_bwd = 3-5 streams.
struct DenseGeo { long nvec; int G, A, nblk; };
template <int VEC, int AT>
__global__ __launch_bounds__(256) void inorm_apply_fast_kernel(const ApplyParams P, const DenseGeo D) {
  const int n = blockIdx.y, t = threadIdx.x;
  if (t >= D.A) return;
  const int grp = t % D.G;
  const bool has_res = P.res != nullptr;
  float sc[VEC], sh[VEC], rsc[VEC], rsh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const size_t k = (size_t)n * P.C + grp * VEC + e;
    sc[e] = P.scale ? P.scale[k] : 1.f; sh[e] = P.scale ? P.shift[k] : 0.f;
    rsc[e] = (has_res && P.rscale) ? P.rscale[k] : 1.f; rsh[e] = (has_res && P.rscale) ? P.rshift[k] : 0.f;
  }
  const long per = ((D.nvec + D.nblk - 1) / D.nblk + D.A - 1) / D.A * D.A;
  const long lo = (long)blockIdx.x * per;
  long hi = lo + per; if (hi > D.nvec) hi = D.nvec;
  const size_t sb = (size_t)n * D.nvec * VEC * mt_ebytes<AT>();        // bytes
  const char* yp = (const char*)P.y + sb;
  const char* rp = has_res ? (const char*)P.res + sb : nullptr;
  char* op = (char*)P.out + sb;
  for (long i0 = lo + t; i0 < hi; i0 += (long)D.A * NF_UNROLL) {
    float y[NF_UNROLL][VEC], r[NF_UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < NF_UNROLL; ++u) {
      const long i = i0 + (long)u * D.A;
      if (i < hi) { mt_ldv<VEC, AT>(yp, (size_t)i, y[u]); if (has_res) mt_ldv<VEC, AT>(rp, (size_t)i, r[u]); }
    }
#pragma unroll
    for (int u = 0; u < NF_UNROLL; ++u) {
      const long i = i0 + (long)u * D.A;
      if (i < hi) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float tt = fmaf(y[u][e], sc[e], sh[e]);
          if (has_res) tt += mt_lrelu(fmaf(r[u][e], rsc[e], rsh[e]), P.rslope);
          o[e] = mt_lrelu(tt, P.slope);
        }
        mt_stv<VEC, AT>(op, (size_t)i, o);
      }
    }
  }
}

template <int VEC, int AT, int GT>
__global__ __launch_bounds__(256) void lrelu_bwd_fast_kernel(const LBwdParams P, const DenseGeo D) {
  static_assert(mt_ebytes<AT>() == mt_ebytes<GT>(), "one vector index addresses all tensors");
  const int n = blockIdx.y, t = threadIdx.x;
  if (t >= D.A) return;
  const int grp = t % D.G;
  const bool has2 = P.y2 != nullptr;
  float sc[VEC], sh[VEC], sc2[VEC], sh2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const size_t k = (size_t)n * P.C + grp * VEC + e;
    sc[e] = P.scale ? P.scale[k] : 1.f; sh[e] = P.scale ? P.shift[k] : 0.f;
    sc2[e] = (has2 && P.scale2) ? P.scale2[k] : 1.f; sh2[e] = (has2 && P.scale2) ? P.shift2[k] : 0.f;
  }
  const long per = ((D.nvec + D.nblk - 1) / D.nblk + D.A - 1) / D.A * D.A;
  const long lo = (long)blockIdx.x * per;
  long hi = lo + per; if (hi > D.nvec) hi = D.nvec;
  const size_t sb = (size_t)n * D.nvec * VEC * mt_ebytes<GT>();
  const char* yp = (const char*)P.y + sb;
  const char* y2p = has2 ? (const char*)P.y2 + sb : nullptr;
  char* gp = (char*)P.g + sb;
  char* cp = P.gcopy ? (char*)P.gcopy + sb : nullptr;
  for (long i0 = lo + t; i0 < hi; i0 += (long)D.A * NF_UNROLL) {
    float y[NF_UNROLL][VEC], y2[NF_UNROLL][VEC], g[NF_UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < NF_UNROLL; ++u) {
      const long i = i0 + (long)u * D.A;
      if (i < hi) { mt_ldv<VEC, AT>(yp, (size_t)i, y[u]); mt_ldv<VEC, GT>(gp, (size_t)i, g[u]); if (has2) mt_ldv<VEC, AT>(y2p, (size_t)i, y2[u]); }
    }
#pragma unroll
    for (int u = 0; u < NF_UNROLL; ++u) {
      const long i = i0 + (long)u * D.A;
      if (i < hi) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float tt = fmaf(y[u][e], sc[e], sh[e]);
          if (has2) tt += mt_lrelu(fmaf(y2[u][e], sc2[e], sh2[e]), P.slope2);
          g[u][e] = tt > 0.f ? g[u][e] : g[u][e] * P.slope;
        }
        mt_stv<VEC, GT>(gp, (size_t)i, g[u]);
        if (cp) mt_stv<VEC, GT>(cp, (size_t)i, g[u]);
      }
    }
  }
}

static DenseGeo dense_geo(long V, int C, int vec) {
  DenseGeo D;
  D.nvec = V * C / vec; D.G = C / vec; D.A = (256 / D.G) * D.G; D.nblk = nb_blocks(V);
  return D;
}
static int launch_apply_fast(const ApplyParams& P, int N, int dtype, int vec, mt_stream_t stream) {
  const DenseGeo per = dense_geo(P.V, P.C, vec);
  const dim3 grid(per.nblk, N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MT_F32) {
    if (vec == 4) hipLaunchKernelGGL((inorm_apply_fast_kernel<4, MT_F32>), grid, dim3(256), 0, st, P, per);
    else if (vec == 2) hipLaunchKernelGGL((inorm_apply_fast_kernel<2, MT_F32>), grid, dim3(256), 0, st, P, per);
    else hipLaunchKernelGGL((inorm_apply_fast_kernel<1, MT_F32>), grid, dim3(256), 0, st, P, per);
  } else if (dtype == MT_F16) {
    if (vec == 8) hipLaunchKernelGGL((inorm_apply_fast_kernel<8, MT_F16>), grid, dim3(256), 0, st, P, per);
    else if (vec == 4) hipLaunchKernelGGL((inorm_apply_fast_kernel<4, MT_F16>), grid, dim3(256), 0, st, P, per);
    else hipLaunchKernelGGL((inorm_apply_fast_kernel<2, MT_F16>), grid, dim3(256), 0, st, P, per);
  } else {
    if (vec == 8) hipLaunchKernelGGL((inorm_apply_fast_kernel<8, MT_BF16>), grid, dim3(256), 0, st, P, per);
    else if (vec == 4) hipLaunchKernelGGL((inorm_apply_fast_kernel<4, MT_BF16>), grid, dim3(256), 0, st, P, per);
    else hipLaunchKernelGGL((inorm_apply_fast_kernel<2, MT_BF16>), grid, dim3(256), 0, st, P, per);
  }
  MT_CHECK_LAUNCH("inorm_apply_fast");
  return MT_OK;
}
template <int AT, int GT>
static void launch_lrelu_bwd_fast(const LBwdParams& P, dim3 grid, int vec, const DenseGeo per, hipStream_t st) {
  if constexpr (AT == MT_F32) {
    if (vec == 4) hipLaunchKernelGGL((lrelu_bwd_fast_kernel<4, AT, GT>), grid, dim3(256), 0, st, P, per);
    else if (vec == 2) hipLaunchKernelGGL((lrelu_bwd_fast_kernel<2, AT, GT>), grid, dim3(256), 0, st, P, per);
    else hipLaunchKernelGGL((lrelu_bwd_fast_kernel<1, AT, GT>), grid, dim3(256), 0, st, P, per);
  } else {
    if (vec == 8) hipLaunchKernelGGL((lrelu_bwd_fast_kernel<8, AT, GT>), grid, dim3(256), 0, st, P, per);
    else if (vec == 4) hipLaunchKernelGGL((lrelu_bwd_fast_kernel<4, AT, GT>), grid, dim3(256), 0, st, P, per);
    else hipLaunchKernelGGL((lrelu_bwd_fast_kernel<2, AT, GT>), grid, dim3(256), 0, st, P, per);
  }
}

extern "C" int mt_lrelu_bwd(float* g, int gcs, const float* y, int ycs, const float* scale, const float* shift, float slope,
                            const float* y2, int y2cs, const float* scale2, const float* shift2, float slope2,
                            float* gcopy, int gcopycs, int N, long V, int C, int gdtype, int ydtype, mt_stream_t stream) {
  MT_REQUIRE(g && y && N > 0 && V > 0 && C > 0 && mt_dtype_ok(gdtype) && mt_dtype_ok(ydtype), "lrelu_bwd: bad args");
  LBwdParams P{g, gcs, y, ycs, scale, shift, slope, y2, y2cs, scale2, shift2, slope2, gcopy, gcopycs, V, C, gdtype, ydtype};
  if (gcs == C && ycs == C && (y2 == nullptr || y2cs == C) && (gcopy == nullptr || gcopycs == C) && V * C < (1L << 40) && ag_fast(ydtype, gdtype)) {
    const int vec = dense_vec(C, V * C, gdtype, {g, y, y2, gcopy});
    if (vec > 0 && C / vec <= 256) {
      const DenseGeo geo = dense_geo(V, C, vec);
      MT_AG_SWITCH(ydtype, gdtype, (launch_lrelu_bwd_fast<AT, GT>(P, dim3(geo.nblk, N), vec, geo, (hipStream_t)stream)), (void)0);
      MT_CHECK_LAUNCH("lrelu_bwd_fast");
      return MT_OK;
    }
  }
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(nb_blocks(V), N), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("lrelu_bwd");
  return MT_OK;
}

// ---- the same with the first pass of the NEXT InstanceNorm backward fused in (residual blocks) ------------------------------------
// out = lrelu(IN(y) + residual): the masked gradient g' = g * lrelu'(t) is the gradient of IN(y) (norm2 of the block, no nonlinearity
// of its own), so the sums its norm backward needs, A = sum g' and B = sum g' * zhat with zhat = (y - mean) * rstd, can be taken
// while g' is being produced — mt_inorm_lrelu_bwd(part = ...) then skips its own reduction over (g', y).  Dense tensors only (channel
// stride == C); thread t keeps VEC constant channels (the mapping of inorm_bwd_fast_kernel), partials [N][nblk][C][2].
struct LBwdStats {
  void* g; const void* y; const void* y2; void* gcopy;
  const float* scale; const float* shift; float slope;
  const float* scale2; const float* shift2; float slope2;
  const float* mean; const float* rstd;
  long nvec; int C, G, A, nblk;
  float* part;
};
template <int VEC, int AT, int GT>
__global__ __launch_bounds__(256) void lrelu_bwd_stats_kernel(const LBwdStats P) {
  static_assert(mt_ebytes<AT>() == mt_ebytes<GT>(), "one vector index addresses all tensors");
  __shared__ float red[256 * VEC * 2];
  const int n = blockIdx.y, t = threadIdx.x;
  const bool act = t < P.A;
  const int grp = t % P.G;
  const bool has2 = P.y2 != nullptr;
  float sc[VEC], sh[VEC], sc2[VEC], sh2[VEC], mu[VEC], rs[VEC], a0[VEC], a1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const size_t k = (size_t)n * P.C + grp * VEC + e;
    sc[e] = P.scale ? P.scale[k] : 1.f; sh[e] = P.scale ? P.shift[k] : 0.f;
    sc2[e] = (has2 && P.scale2) ? P.scale2[k] : 1.f; sh2[e] = (has2 && P.scale2) ? P.shift2[k] : 0.f;
    mu[e] = P.mean[k]; rs[e] = P.rstd[k];
    a0[e] = 0.f; a1[e] = 0.f;
  }
  const long per = ((P.nvec + P.nblk - 1) / P.nblk + P.A - 1) / P.A * P.A;
  const long lo = (long)blockIdx.x * per;
  long hi = lo + per; if (hi > P.nvec) hi = P.nvec;
  const size_t sb = (size_t)n * P.nvec * VEC * mt_ebytes<GT>();
  char* gp = (char*)P.g + sb;
  char* cp = P.gcopy ? (char*)P.gcopy + sb : nullptr;
  const char* yp = (const char*)P.y + sb;
  const char* y2p = has2 ? (const char*)P.y2 + sb : nullptr;
  if (act) {
    for (long i0 = lo + t; i0 < hi; i0 += (long)P.A * NF_UNROLL) {
      float gv[NF_UNROLL][VEC], yv[NF_UNROLL][VEC], y2v[NF_UNROLL][VEC];
#pragma unroll
      for (int u = 0; u < NF_UNROLL; ++u) {
        const long i = i0 + (long)u * P.A;
        if (i < hi) { mt_ldv<VEC, GT>(gp, (size_t)i, gv[u]); mt_ldv<VEC, AT>(yp, (size_t)i, yv[u]); if (has2) mt_ldv<VEC, AT>(y2p, (size_t)i, y2v[u]); }
      }
#pragma unroll
      for (int u = 0; u < NF_UNROLL; ++u) {
        const long i = i0 + (long)u * P.A;
        if (i < hi) {
          float out[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float yy = yv[u][e];
            float tt = fmaf(yy, sc[e], sh[e]);
            if (has2) tt += mt_lrelu(fmaf(y2v[u][e], sc2[e], sh2[e]), P.slope2);
            float gg = gv[u][e];
            gg = tt > 0.f ? gg : gg * P.slope;
            gg = mt_round_st<GT>(gg);
            out[e] = gg;
            const float zh = (yy - mu[e]) * rs[e];
            a0[e] += gg;
            a1[e] = fmaf(gg, zh, a1[e]);
          }
          mt_stv<VEC, GT>(gp, (size_t)i, out);
          if (cp) mt_stv<VEC, GT>(cp, (size_t)i, out);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) { red[(t * VEC + e) * 2] = act ? a0[e] : 0.f; red[(t * VEC + e) * 2 + 1] = act ? a1[e] : 0.f; }
  __syncthreads();
  if (t < P.G) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float s0 = 0.f, s1 = 0.f;
      for (int k = t; k < P.A; k += P.G) { s0 += red[(k * VEC + e) * 2]; s1 += red[(k * VEC + e) * 2 + 1]; }
      float* o = P.part + (((size_t)n * P.nblk + blockIdx.x) * P.C + t * VEC + e) * 2;
      o[0] = s0; o[1] = s1;
    }
  }
}
// blocks per sample of the partials (0: the tensors do not take the fused path — call mt_lrelu_bwd and let the norm backward reduce)
extern "C" int mt_lrelu_bwd_stats_blocks(long V, int C) {
  if (V <= 0 || C <= 0 || (C % 4) != 0 || C / 4 > 256 || (((long)V * C * 4) % 16) != 0) return 0;
  return nb_blocks(V);
}
extern "C" int mt_lrelu_bwd_stats(float* g, const float* y, const float* scale, const float* shift, float slope,
                                  const float* y2, const float* scale2, const float* shift2, float slope2, float* gcopy,
                                  const float* mean, const float* rstd, float* part, int N, long V, int C, int gdtype, int ydtype, mt_stream_t stream) {
  MT_REQUIRE(g && y && mean && rstd && part && N > 0 && ag_fast(ydtype, gdtype), "lrelu_bwd_stats: storage types (g %d, y %d) not taken (fp32/fp32, bf16/fp16, bf16/bf16)", gdtype, ydtype);
  const int nblk = mt_lrelu_bwd_stats_blocks(V, C);
  MT_REQUIRE(nblk > 0, "lrelu_bwd_stats: unsupported shape (V=%ld, C=%d): ask mt_lrelu_bwd_stats_blocks", V, C);
  MT_REQUIRE(((((uintptr_t)g) | ((uintptr_t)y) | ((uintptr_t)y2) | ((uintptr_t)gcopy)) & 15) == 0, "lrelu_bwd_stats: tensors must be 16-byte aligned");
  LBwdStats P;
  P.g = g; P.y = y; P.y2 = y2; P.gcopy = gcopy; P.scale = scale; P.shift = shift; P.slope = slope;
  P.scale2 = scale2; P.shift2 = shift2; P.slope2 = slope2; P.mean = mean; P.rstd = rstd;
  P.nvec = (long)V * C / 4; P.C = C; P.G = C / 4; P.A = (256 / P.G) * P.G; P.nblk = nblk; P.part = part;
  MT_AG_SWITCH(ydtype, gdtype, hipLaunchKernelGGL((lrelu_bwd_stats_kernel<4, AT, GT>), dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, P), (void)0);
  MT_CHECK_LAUNCH("lrelu_bwd_stats");
  return MT_OK;
}

// ---- per-channel sum -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const void* x, int xcs, long V, int C, int nvb, float* part, int st) {
  __shared__ float red[8][32];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  const int vb = mt_vb(V);
  const long v0 = (long)blockIdx.x * vb;
  const long v1 = (v0 + vb < V) ? v0 + vb : V;
  for (int cb = 0; cb < C; cb += 32) {
    const int c = cb + cl;
    float a = 0.f;
    if (c < C)
      for (long v = v0 + vr; v < v1; v += 8) a += mt_ld_rt(x, ((size_t)n * V + v) * xcs + c, st);
    red[vr][cl] = a;
    __syncthreads();
    if (threadIdx.x < 32) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
      if (cb + threadIdx.x < C) part[((size_t)n * nvb + blockIdx.x) * C + cb + threadIdx.x] = s;
    }
    __syncthreads();
  }
}
extern "C" size_t mt_channel_sum_workspace(int N, long V, int C) { return (size_t)N * nb_blocks(V) * C * sizeof(float); }
extern "C" int mt_channel_sum(const float* x, int xcs, int N, long V, int C, float* out, int accumulate, void* ws,
                              size_t ws_bytes, int dtype, mt_stream_t stream) {
  MT_REQUIRE(x && out && N > 0 && V > 0 && C > 0 && mt_dtype_ok(dtype), "channel_sum: bad args");
  const int nvb = nb_blocks(V);
  if (ws == nullptr || ws_bytes < (size_t)N * nvb * C * sizeof(float)) { mt_set_error("channel_sum: workspace too small"); return MT_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(nvb, N), dim3(256), 0, st, (const void*)x, xcs, V, C, nvb, (float*)ws, dtype);
  hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(64), 0, st, (const float*)ws, (long)N * nvb, C, out, accumulate);
  MT_CHECK_LAUNCH("channel_sum");
  return MT_OK;
}

// ---- storage-type conversion ---------------------------------------------------------------------------
// dst[r][c] (+)= src[r][c] for rows r < rows (voxels of all samples) and c < C, each side with its own storage type and channel
// stride.  The boundary op of the mixed-precision mode: a kernel that does not take a tensor's storage type reads / writes an
// fp32 (or bf16) copy instead (see the *_io_supported queries); accumulate adds in fp32 and rounds once.
struct CastParams { const void* src; void* dst; long rows; int C, scs, dcs, accumulate; };
template <int SB, int DB, int VEC>
__global__ __launch_bounds__(256) void cast_dense_kernel(const CastParams P, long nvec) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float v[VEC];
    mt_ldv<VEC, SB>(P.src, (size_t)i, v);
    if (P.accumulate) {
      float o[VEC];
      mt_ldv<VEC, DB>(P.dst, (size_t)i, o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += o[e];
    }
    mt_stv<VEC, DB>(P.dst, (size_t)i, v);
  }
}
__global__ __launch_bounds__(256) void cast_strided_kernel(const CastParams P, int sb, int db) {
  const int cl = threadIdx.x & 31, vr = threadIdx.x >> 5;
  for (long r = (long)blockIdx.x * 8 + vr; r < P.rows; r += (long)gridDim.x * 8)
    for (int c = cl; c < P.C; c += 32) {
      float v = mt_ld_rt(P.src, (size_t)r * P.scs + c, sb);
      if (P.accumulate) v += mt_ld_rt(P.dst, (size_t)r * P.dcs + c, db);
      mt_st_rt(P.dst, (size_t)r * P.dcs + c, v, db);
    }
}
extern "C" int mt_cast(const void* src, int scs, int sdtype, void* dst, int dcs, int ddtype, long rows, int C, int accumulate,
                       mt_stream_t stream) {
  MT_REQUIRE(src && dst && rows > 0 && C > 0 && scs >= C && dcs >= C && mt_dtype_ok(sdtype) && mt_dtype_ok(ddtype), "cast: bad args");
  CastParams P{src, dst, rows, C, scs, dcs, accumulate};
  hipStream_t st = (hipStream_t)stream;
  const long n = rows * C;
  const bool dense = scs == C && dcs == C && (n % 4) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  if (dense) {
    const long nvec = n / 4;
    int blocks = (int)((nvec + 255) / 256); if (blocks > 16384) blocks = 16384;
    const dim3 grid(blocks);
#define MT_CAST_CASE(S_, D_) if (sdtype == S_ && ddtype == D_) hipLaunchKernelGGL((cast_dense_kernel<S_, D_, 4>), grid, dim3(256), 0, st, P, nvec);
    MT_CAST_CASE(MT_F32, MT_F32) else MT_CAST_CASE(MT_F32, MT_BF16) else MT_CAST_CASE(MT_F32, MT_F16)
    else MT_CAST_CASE(MT_BF16, MT_F32) else MT_CAST_CASE(MT_BF16, MT_BF16) else MT_CAST_CASE(MT_BF16, MT_F16)
    else MT_CAST_CASE(MT_F16, MT_F32) else MT_CAST_CASE(MT_F16, MT_BF16) else MT_CAST_CASE(MT_F16, MT_F16)
#undef MT_CAST_CASE
  } else {
    int blocks = (int)((rows + 7) / 8); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(cast_strided_kernel, dim3(blocks), dim3(256), 0, st, P, sdtype, ddtype);
  }
  MT_CHECK_LAUNCH("cast");
  return MT_OK;
}

// ---- layout transposes at the module boundary (NCDHW <-> NDHWC) ------------------------------------
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const float* in, float* out, int C, long V, int ocs) {
  __shared__ float t[32][33];
  const int n = blockIdx.z;
  const long v0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r; const long v = v0 + tx;
    t[r][tx] = (c < C && v < V) ? in[((size_t)n * C + c) * V + v] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long v = v0 + r; const int c = c0 + tx;
    if (c < C && v < V) out[((size_t)n * V + v) * ocs + c] = t[tx][r];
  }
}
__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const float* in, int ics, float* out, int C, long V) {
  __shared__ float t[32][33];
  const int n = blockIdx.z;
  const long v0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const long v = v0 + r; const int c = c0 + tx;
    t[r][tx] = (c < C && v < V) ? in[((size_t)n * V + v) * ics + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r; const long v = v0 + tx;
    if (c < C && v < V) out[((size_t)n * C + c) * V + v] = t[tx][r];
  }
}
extern "C" int mt_ncdhw_to_ndhwc(const float* in, float* out, int N, int C, long V, int ocs, mt_stream_t stream) {
  MT_REQUIRE(in && out && N > 0 && C > 0 && V > 0, "ncdhw_to_ndhwc: bad args");
  hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, dim3(mt_cdiv(V, 32), mt_cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream, in, out, C, V, ocs);
  MT_CHECK_LAUNCH("ncdhw_to_ndhwc");
  return MT_OK;
}
extern "C" int mt_ndhwc_to_ncdhw(const float* in, int ics, float* out, int N, int C, long V, mt_stream_t stream) {
  MT_REQUIRE(in && out && N > 0 && C > 0 && V > 0, "ndhwc_to_ncdhw: bad args");
  hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, dim3(mt_cdiv(V, 32), mt_cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream, in, ics, out, C, V);
  MT_CHECK_LAUNCH("ndhwc_to_ncdhw");
  return MT_OK;
}
