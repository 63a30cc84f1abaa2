// loss.hip — fused segmentation losses, forward statistics and closed-form backward.
//
// MultiTalent loss (MultiTalent_Trainer_DDP.py:544-623 == MultiTalent_meets_resenc.py:713-798):
//   per sample b and region r valid for b's source dataset:  y = target in labels(r);
//   ce += mean_v BCEWithLogits(x[b,idx_r], y);  tp/fp/fn[b,idx_r] = sum sigmoid*y, sigmoid*(1-y), (1-sigmoid)*y
// The reference launches ~10 kernels per (b, region, level) and its autograd materialises a full
// zero tensor per region; here ONE pass reads the logits once (wave = 64 channel lanes of one voxel,
// label-set membership from a 64-bit LUT per channel) and ONE pass writes dlogits once.
//
// Softmax Dice+CE (dice_loss.py:100-195,488-545; crossentropy.py:4-11; nnUNetTrainerV2_DDP.py:249-282):
// one thread per voxel, classes in registers.
#include "mt_common.h"

#define LS_VB 1024  // voxels per block

// stats partial layout: part[b][blk][C][4]
__global__ __launch_bounds__(256) void mt_loss_fwd_kernel(const float* __restrict__ logits, int cs,
                                                          const float* __restrict__ target, long V, int C,
                                                          const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                          int nblk, float* __restrict__ part) {
  __shared__ float red[4][64][4];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long v0 = (long)blockIdx.x * LS_VB;
  const long v1 = (v0 + LS_VB < V) ? v0 + LS_VB : V;
  const uint64_t vmask = valid[b];
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + lane;
    const bool act = (c < C) && ((vmask >> c) & 1ull);
    const uint64_t l = act ? lut[c] : 0ull;
    float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    for (long v = v0 + wave; v < v1; v += 4) {
      const size_t e = (size_t)b * V + v;
      const int lab = (int)target[e];
      if (act) {
        const float x = logits[e * cs + c];
        const float y = (lab >= 0 && lab < 64 && ((l >> lab) & 1ull)) ? 1.f : 0.f;
        const float ax = fabsf(x);
        const float ex = __expf(-ax);
        // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|))
        bce += fmaxf(x, 0.f) - x * y + log1pf(ex);
        const float s = (x >= 0.f) ? 1.f / (1.f + ex) : ex / (1.f + ex);
        tp += s * y;
        fp += s * (1.f - y);
        fn += (1.f - s) * y;
      }
    }
    red[wave][lane][0] = bce; red[wave][lane][1] = tp; red[wave][lane][2] = fp; red[wave][lane][3] = fn;
    __syncthreads();
    {
      const int k = threadIdx.x & 3, cc = threadIdx.x >> 2;  // 64 channels x 4 stats
      const float s = red[0][cc][k] + red[1][cc][k] + red[2][cc][k] + red[3][cc][k];
      if (cb + cc < C) part[(((size_t)b * nblk + blockIdx.x) * C + cb + cc) * 4 + k] = s;
    }
    __syncthreads();
  }
}

// stats[b][c][k] = sum_blk part[b][blk][c][k]
__global__ __launch_bounds__(64) void loss_stats_finalize_kernel(const float* part, int nblk, int C, float* stats) {
  const int b = blockIdx.y, ck = blockIdx.x;  // ck = c*4 + k
  double a = 0.0;
  for (int s = threadIdx.x; s < nblk; s += 64) a += (double)part[((size_t)b * nblk + s) * C * 4 + ck];
  a = mt_wave_sum_d(a);
  if (threadIdx.x == 0) stats[(size_t)b * C * 4 + ck] = (float)a;
}

// ---- contiguous logits (cs == C): flat, fully coalesced form ------------------------------------------------------------------
// The [V][C] slab of a sample is one linear array.  A = (256 / C) * C threads are active; thread t handles the flat elements
// base + t + k*A, whose channel is t % C for every k (A and the block base are multiples of C): per-channel constants (validity,
// label set) and the four running sums live in registers, consecutive lanes read consecutive floats, and all lanes work (the
// wave-per-voxel kernel above keeps 47 of 64 lanes busy and touches 188-byte rows).  Threads of invalid channels skip their loads.
#define LF_KMAX 240          // elements per thread per block (block covers LF_KMAX * (A / C) voxels, at most LF_VMAX)
#define LF_VMAX 4096
__host__ __device__ static inline long lf_vpb(int nsub) { const long v = (long)LF_KMAX * nsub; return v < LF_VMAX ? v : LF_VMAX; }
__device__ __forceinline__ void lf_fwd_flat_body(const float* __restrict__ logits, const float* __restrict__ target, long V, int C,
                                                 const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                 int nblk, int A, float* __restrict__ part, float (*red)[4], signed char* slab) {
  // red[256][4]; slab[LF_VMAX]: the block's labels (-1 = none of the 64 label values)
  const int b = blockIdx.y, t = threadIdx.x;
  const int c = t % C, sub = t / C, nsub = A / C;                  // `sub`-th voxel of every group of nsub voxels
  const bool act = (t < A) && ((valid[b] >> c) & 1ull);
  const uint64_t l = act ? lut[c] : 0ull;
  const long vpb = lf_vpb(nsub);                                   // voxels per block
  const long v0 = (long)blockIdx.x * vpb;
  const long v1 = (v0 + vpb < V) ? v0 + vpb : V;
  const float* x = logits + ((size_t)b * V) * C + c;
  const float* tg = target + (size_t)b * V;
  for (int i = t; i < (int)(v1 - v0); i += 256) {                  // every label is read once per block, not once per channel
    const int lab = (int)tg[v0 + i];
    slab[i] = (signed char)((lab >= 0 && lab < 64) ? lab : -1);
  }
  __syncthreads();
  float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
  if (act) {
    for (long vb = v0 + sub; vb < v1; vb += 4 * nsub) {
      float xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long v = vb + (long)u * nsub;
        xv[u] = (v < v1) ? x[(size_t)v * C] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long v = vb + (long)u * nsub;
        if (v < v1) {
          const float xx = xv[u];
          const int lab = slab[v - v0];
          const float y = (lab >= 0 && ((l >> lab) & 1ull)) ? 1.f : 0.f;
          const float ex = __expf(-fabsf(xx));                       // in (0, 1]
          const float r = __builtin_amdgcn_rcpf(1.f + ex);
          // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|)); log(1 + ex) with 1 + ex rounded: absolute error <= 6e-8
          bce += fmaxf(xx, 0.f) - xx * y + __logf(1.f + ex);
          const float sg = (xx >= 0.f) ? r : ex * r;
          tp += sg * y;
          fp += sg * (1.f - y);
          fn += (1.f - sg) * y;
        }
      }
    }
  }
  red[t][0] = bce; red[t][1] = tp; red[t][2] = fp; red[t][3] = fn;
  __syncthreads();
  if (t < C * 4) {                                                  // fixed-order sum over the nsub threads of a channel
    const int cc = t >> 2, k = t & 3;
    float sum = 0.f;
    for (int q = 0; q < nsub; ++q) sum += red[q * C + cc][k];
    part[(((size_t)b * nblk + blockIdx.x) * C + cc) * 4 + k] = sum;
  }
}
__global__ __launch_bounds__(256) void mt_loss_fwd_flat_kernel(const float* __restrict__ logits, const float* __restrict__ target, long V, int C,
                                                               const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                               int nblk, int A, float* __restrict__ part) {
  __shared__ float red[256][4];
  __shared__ signed char slab[LF_VMAX];
  lf_fwd_flat_body(logits, target, V, C, valid, lut, nblk, A, part, red, slab);
}

__global__ __launch_bounds__(256) void mt_loss_bwd_flat_kernel(const float* __restrict__ logits, const float* __restrict__ target, long V, int C,
                                                               const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                               const float* __restrict__ gstats, int A, float* __restrict__ dlogits) {
  __shared__ signed char slab[LF_VMAX];
  const int b = blockIdx.y, t = threadIdx.x;
  const int c = t % C, sub = t / C, nsub = A / C;
  const bool act = (t < A) && ((valid[b] >> c) & 1ull);
  const uint64_t l = act ? lut[c] : 0ull;
  {
    const long vpb_ = lf_vpb(nsub), a0 = (long)blockIdx.x * vpb_, a1 = (a0 + vpb_ < V) ? a0 + vpb_ : V;
    for (int i = t; i < (int)(a1 - a0); i += 256) {
      const int lab = (int)target[(size_t)b * V + a0 + i];
      slab[i] = (signed char)((lab >= 0 && lab < 64) ? lab : -1);
    }
    __syncthreads();
  }
  if (t >= A) return;
  const float* gs = gstats + ((size_t)b * C + c) * 4;
  const float bce_coef = act ? gs[0] : 0.f, a_tp = act ? gs[1] : 0.f, a_fp = act ? gs[2] : 0.f, a_fn = act ? gs[3] : 0.f;
  const long vpb = lf_vpb(nsub);
  const long v0 = (long)blockIdx.x * vpb;
  const long v1 = (v0 + vpb < V) ? v0 + vpb : V;
  const float* x = logits + ((size_t)b * V) * C + c;
  float* dx = dlogits + ((size_t)b * V) * C + c;
  for (long vb = v0 + sub; vb < v1; vb += 4 * nsub) {
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long v = vb + (long)u * nsub;
      xv[u] = (act && v < v1) ? x[(size_t)v * C] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long v = vb + (long)u * nsub;
      if (v < v1) {
        float d = 0.f;
        if (act) {
          const float xx = xv[u];
          const int lab = slab[v - v0];
          const float y = (lab >= 0 && ((l >> lab) & 1ull)) ? 1.f : 0.f;
          const float ex = __expf(-fabsf(xx));
          const float r = __builtin_amdgcn_rcpf(1.f + ex);
          const float sg = (xx >= 0.f) ? r : ex * r;
          d = bce_coef * (sg - y) + sg * (1.f - sg) * (y * (a_tp - a_fn) + (1.f - y) * a_fp);
        }
        dx[(size_t)v * C] = d;
      }
    }
  }
}
// ---- few valid regions (round 4).  A MultiTalent sample carries the regions of ITS dataset only (2-3 of 47 for most of the thirteen):
// in the flat kernels above 15 of 256 threads then do all the work (forward) resp. every thread walks its channel's elements to store
// zeros one dword at a time (backward): 1.2 / 2.5 TB/s by the counters (profiles/r04_pmc_per_kernel.json), 9 % of a Task100 mixed step.
// Forward: the block's threads are spread over (voxel, ACTIVE channel) pairs — thread t owns active channel alist[t % nact] of every
// (256 / nact)-th voxel, per-channel constants in registers as before, partials in the same [block][C][4] layout (zeros for the other
// channels), fixed-order sums.
#define LF_SPARSE_MAX 23      // at most this many valid regions: every thread of the block can be given work
__global__ __launch_bounds__(256) void mt_loss_fwd_sparse_kernel(const float* __restrict__ logits, const float* __restrict__ target, long V, int C,
                                                                 const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                                 int nblk, long vpb, float* __restrict__ part) {
  __shared__ float red[256][4];
  __shared__ signed char slab[LF_VMAX];
  __shared__ int alist[64];
  const int b = blockIdx.y, t = threadIdx.x;
  const uint64_t vm = valid[b] & (C < 64 ? ((1ull << C) - 1ull) : ~0ull);
  const int nact = __popcll(vm);                                   // block-uniform
  if (nact < 1 || nact > LF_SPARSE_MAX) {                          // a sample with many valid regions: the flat form keeps every thread busy
    lf_fwd_flat_body(logits, target, V, C, valid, lut, nblk, (256 / C) * C, part, red, slab);
    return;
  }
  if (t < C && ((vm >> t) & 1ull)) alist[__popcll(vm & ((1ull << t) - 1ull))] = t;
  const long v0 = (long)blockIdx.x * vpb;
  const long v1 = (v0 + vpb < V) ? v0 + vpb : V;
  const float* tg = target + (size_t)b * V;
  for (int i = t; i < (int)(v1 - v0); i += 256) {
    const int lab = (int)tg[v0 + i];
    slab[i] = (signed char)((lab >= 0 && lab < 64) ? lab : -1);
  }
  __syncthreads();
  const int nsub = 256 / nact, A = nsub * nact;
  const int k = t % nact, sub = t / nact;
  const bool act = t < A;
  const int c = alist[act ? k : 0];
  const uint64_t l = lut[c];
  const float* x = logits + ((size_t)b * V) * C + c;
  float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
  if (act) {
    for (long vb = v0 + sub; vb < v1; vb += 4 * nsub) {
      float xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long v = vb + (long)u * nsub;
        xv[u] = (v < v1) ? x[(size_t)v * C] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long v = vb + (long)u * nsub;
        if (v < v1) {
          const float xx = xv[u];
          const int lab = slab[v - v0];
          const float y = (lab >= 0 && ((l >> lab) & 1ull)) ? 1.f : 0.f;
          const float ex = __expf(-fabsf(xx));
          const float r = __builtin_amdgcn_rcpf(1.f + ex);
          bce += fmaxf(xx, 0.f) - xx * y + __logf(1.f + ex);
          const float sg = (xx >= 0.f) ? r : ex * r;
          tp += sg * y;
          fp += sg * (1.f - y);
          fn += (1.f - sg) * y;
        }
      }
    }
  }
  red[t][0] = bce; red[t][1] = tp; red[t][2] = fp; red[t][3] = fn;
  __syncthreads();
  float* pb = part + ((size_t)b * nblk + blockIdx.x) * C * 4;
  if (t < C * 4 && !((vm >> (t >> 2)) & 1ull)) pb[t] = 0.f;         // regions this sample does not carry
  if (t < nact * 4) {                                               // fixed-order sum over the nsub threads of an active channel
    const int kk = t >> 2, st = t & 3;
    float sum = 0.f;
    for (int q = 0; q < nsub; ++q) sum += red[q * nact + kk][st];
    pb[alist[kk] * 4 + st] = sum;
  }
}
// Backward: the gradient is dense (zeros for the regions a sample does not carry), so every thread owns FOUR CONSECUTIVE floats of the
// flat [V][C] slab and stores them as one 16-byte vector; the logit is read (a scalar load) only for the few elements whose channel is
// valid.  (V * C and the block's first element are multiples of four.)
__global__ __launch_bounds__(256) void mt_loss_bwd_wide_kernel(const float* __restrict__ logits, const float* __restrict__ target, long V, int C,
                                                               const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                               const float* __restrict__ gstats, long vpb, float* __restrict__ dlogits) {
  __shared__ signed char slab[LF_VMAX];
  __shared__ float gsl[64][4];
  __shared__ uint64_t lutl[64];
  const int b = blockIdx.y, t = threadIdx.x;
  const uint64_t vm = valid[b] & (C < 64 ? ((1ull << C) - 1ull) : ~0ull);
  const long v0 = (long)blockIdx.x * vpb;
  const long v1 = (v0 + vpb < V) ? v0 + vpb : V;
  for (int i = t; i < (int)(v1 - v0); i += 256) {
    const int lab = (int)target[(size_t)b * V + v0 + i];
    slab[i] = (signed char)((lab >= 0 && lab < 64) ? lab : -1);
  }
  if (t < C) {
    lutl[t] = lut[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) gsl[t][k] = gstats[((size_t)b * C + t) * 4 + k];
  }
  __syncthreads();
  const size_t base = ((size_t)b * V + v0) * C;                    // first element of the block
  const int nel = (int)((v1 - v0) * C);                            // multiple of 4
  int fr = 4 * t;                                                  // element offset inside the block
  int vr = fr / C, c = fr - vr * C;                                // its voxel (relative) and channel
  const int dv = 1024 / C, dc = 1024 - dv * C;                     // advance of 256 threads x 4 elements
  const float* xb = logits + base;
  float* db = dlogits + base;
  for (; fr < nel; fr += 1024) {
    float d[4];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int ce = (c + e >= C) ? c + e - C : c + e; any = any || ((vm >> ce) & 1ull); }
    float4 xq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) xq = *(const float4*)(xb + fr);
    const float xa[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int ce = c + e, ve = vr;
      if (ce >= C) { ce -= C; ve += 1; }
      d[e] = 0.f;
      if ((vm >> ce) & 1ull) {
        const float xx = xa[e];
        const int lab = slab[ve];
        const float y = (lab >= 0 && ((lutl[ce] >> lab) & 1ull)) ? 1.f : 0.f;
        const float ex = __expf(-fabsf(xx));
        const float r = __builtin_amdgcn_rcpf(1.f + ex);
        const float sg = (xx >= 0.f) ? r : ex * r;
        d[e] = gsl[ce][0] * (sg - y) + sg * (1.f - sg) * (y * (gsl[ce][1] - gsl[ce][3]) + (1.f - y) * gsl[ce][2]);
      }
    }
    *(float4*)(db + fr) = make_float4(d[0], d[1], d[2], d[3]);
    vr += dv; c += dc;
    if (c >= C) { c -= C; vr += 1; }
  }
}
static inline int lf_A(int C) { return (256 / C) * C; }
static inline int lf_blocks(long V, int C) { return mt_cdiv(V, lf_vpb(lf_A(C) / C)); }

extern "C" size_t mt_loss_workspace(int B, long V, int C) {
  size_t nb = (size_t)mt_cdiv(V, LS_VB);
  if (C > 0 && C <= 64 && (size_t)lf_blocks(V, C) > nb) nb = (size_t)lf_blocks(V, C);
  return (size_t)B * nb * C * 4 * sizeof(float);
}

extern "C" int mt_multitalent_loss_fwd(const float* logits, int cs, const float* target, int B, long V, int C,
                                       const uint64_t* valid, const uint64_t* lut, float* stats, void* ws,
                                       size_t ws_bytes, mt_stream_t stream) {
  MT_REQUIRE(logits && target && valid && lut && stats && B > 0 && V > 0 && C > 0 && C <= 64, "multitalent_loss_fwd: bad args (C must be <= 64)");
  if (ws == nullptr || ws_bytes < mt_loss_workspace(B, V, C)) { mt_set_error("multitalent_loss_fwd: workspace too small"); return MT_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  int nblk;
  if (cs == C) {            // contiguous logits (what the engine produces): flat coalesced kernel
    nblk = lf_blocks(V, C);
    constexpr int sparse = 1;
    if (sparse) hipLaunchKernelGGL(mt_loss_fwd_sparse_kernel, dim3(nblk, B), dim3(256), 0, st, logits, target, V, C, valid, lut, nblk, lf_vpb(lf_A(C) / C), (float*)ws);
    else hipLaunchKernelGGL(mt_loss_fwd_flat_kernel, dim3(nblk, B), dim3(256), 0, st, logits, target, V, C, valid, lut, nblk, lf_A(C), (float*)ws);
  } else {
    nblk = mt_cdiv(V, LS_VB);
    hipLaunchKernelGGL(mt_loss_fwd_kernel, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, valid, lut, nblk, (float*)ws);
  }
  hipLaunchKernelGGL(loss_stats_finalize_kernel, dim3(C * 4, B), dim3(64), 0, st, (const float*)ws, nblk, C, stats);
  MT_CHECK_LAUNCH("multitalent_loss_fwd");
  return MT_OK;
}

__global__ __launch_bounds__(256) void mt_loss_bwd_kernel(const float* __restrict__ logits, int cs,
                                                          const float* __restrict__ target, long V, int C,
                                                          const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                          const float* __restrict__ gstats,
                                                          float* __restrict__ dlogits, int dcs) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long v0 = (long)blockIdx.x * LS_VB;
  const long v1 = (v0 + LS_VB < V) ? v0 + LS_VB : V;
  const uint64_t vmask = valid[b];
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + lane;
    if (c >= C) continue;
    const bool act = (vmask >> c) & 1ull;
    const uint64_t l = act ? lut[c] : 0ull;
    const float* gs = gstats + ((size_t)b * C + c) * 4;
    const float bce_coef = act ? gs[0] : 0.f, a_tp = act ? gs[1] : 0.f, a_fp = act ? gs[2] : 0.f, a_fn = act ? gs[3] : 0.f;
    for (long v = v0 + wave; v < v1; v += 4) {
      const size_t e = (size_t)b * V + v;
      float d = 0.f;
      if (act) {
        const int lab = (int)target[e];
        const float x = logits[e * cs + c];
        const float y = (lab >= 0 && lab < 64 && ((l >> lab) & 1ull)) ? 1.f : 0.f;
        const float ex = __expf(-fabsf(x));
        const float s = (x >= 0.f) ? 1.f / (1.f + ex) : ex / (1.f + ex);
        // d tp/dx = s(1-s) y ; d fp/dx = s(1-s)(1-y) ; d fn/dx = -s(1-s) y
        const float ds = s * (1.f - s);
        d = bce_coef * (s - y) + ds * (y * (a_tp - a_fn) + (1.f - y) * a_fp);
      }
      dlogits[e * dcs + c] = d;
    }
  }
}

extern "C" int mt_multitalent_loss_bwd(const float* logits, int cs, const float* target, int B, long V, int C,
                                       const uint64_t* valid, const uint64_t* lut, const float* gstats,
                                       float* dlogits, int dcs, mt_stream_t stream) {
  MT_REQUIRE(logits && target && valid && lut && gstats && dlogits && B > 0 && V > 0 && C > 0 && C <= 64, "multitalent_loss_bwd: bad args");
  constexpr int wide = 1;
  const long vpb = lf_vpb(lf_A(C) / C);
  // C >= 2: the quad's channel index is unwrapped by ONE conditional subtraction (c + 3 < 2C), wrong for a single channel
  if (cs == C && dcs == C && wide && C >= 2 && ((V * C) & 3) == 0 && ((vpb * C) & 3) == 0 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0)
    hipLaunchKernelGGL(mt_loss_bwd_wide_kernel, dim3(lf_blocks(V, C), B), dim3(256), 0, (hipStream_t)stream, logits, target, V, C, valid, lut,
                       gstats, vpb, dlogits);
  else if (cs == C && dcs == C)
    hipLaunchKernelGGL(mt_loss_bwd_flat_kernel, dim3(lf_blocks(V, C), B), dim3(256), 0, (hipStream_t)stream, logits, target, V, C, valid, lut,
                       gstats, lf_A(C), dlogits);
  else
    hipLaunchKernelGGL(mt_loss_bwd_kernel, dim3(mt_cdiv(V, LS_VB), B), dim3(256), 0, (hipStream_t)stream, logits, cs, target, V, C,
                       valid, lut, gstats, dlogits, dcs);
  MT_CHECK_LAUNCH("multitalent_loss_bwd");
  return MT_OK;
}

// ---- online evaluation (MultiTalent_Trainer_DDP.py:372-410): hard predictions sigmoid(x) > 0.5, exact integer counts --------
__global__ __launch_bounds__(256) void mt_hard_stats_kernel(const float* __restrict__ logits, int cs,
                                                            const float* __restrict__ target, long V, int C,
                                                            const uint64_t* __restrict__ valid, const uint64_t* __restrict__ lut,
                                                            unsigned long long* __restrict__ counts) {
  __shared__ unsigned int red[4][64][3];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long v0 = (long)blockIdx.x * LS_VB;
  const long v1 = (v0 + LS_VB < V) ? v0 + LS_VB : V;
  const uint64_t vmask = valid[b];
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + lane;
    const bool act = (c < C) && ((vmask >> c) & 1ull);
    const uint64_t l = act ? lut[c] : 0ull;
    unsigned int tp = 0, fp = 0, fn = 0;
    for (long v = v0 + wave; v < v1; v += 4) {
      const size_t e = (size_t)b * V + v;
      const int lab = (int)target[e];
      if (act) {
        const float x = logits[e * cs + c];
        const bool y = (lab >= 0 && lab < 64 && ((l >> lab) & 1ull));
        const float ex = __expf(-fabsf(x));
        const float sg = (x >= 0.f) ? 1.f / (1.f + ex) : ex / (1.f + ex);
        const bool p = sg > 0.5f;
        tp += (p && y); fp += (p && !y); fn += (!p && y);
      }
    }
    red[wave][lane][0] = tp; red[wave][lane][1] = fp; red[wave][lane][2] = fn;
    __syncthreads();
    if (threadIdx.x < 192) {
      const int k = threadIdx.x % 3, cc = threadIdx.x / 3;
      const unsigned int sum = red[0][cc][k] + red[1][cc][k] + red[2][cc][k] + red[3][cc][k];
      if (cb + cc < C && sum) atomicAdd(&counts[((size_t)b * C + cb + cc) * 3 + k], (unsigned long long)sum);   // integer: order-free
    }
    __syncthreads();
  }
}

__global__ void mt_counts_to_float_kernel(const unsigned long long* counts, float* stats, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stats[i] = (float)counts[i];
}

extern "C" size_t mt_hard_stats_workspace(int B, int C) { return (size_t)B * C * 3 * sizeof(unsigned long long); }

extern "C" int mt_multitalent_hard_stats(const float* logits, int cs, const float* target, int B, long V, int C,
                                         const uint64_t* valid, const uint64_t* lut, float* stats, void* ws, size_t ws_bytes,
                                         mt_stream_t stream) {
  MT_REQUIRE(logits && target && valid && lut && stats && B > 0 && V > 0 && C > 0 && C <= 64, "multitalent_hard_stats: bad args (C must be <= 64)");
  if (ws == nullptr || ws_bytes < mt_hard_stats_workspace(B, C)) { mt_set_error("multitalent_hard_stats: workspace too small"); return MT_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, mt_hard_stats_workspace(B, C), st) != hipSuccess) { mt_set_error("multitalent_hard_stats: memset failed"); return MT_EHIP; }
  hipLaunchKernelGGL(mt_hard_stats_kernel, dim3(mt_cdiv(V, LS_VB), B), dim3(256), 0, st, logits, cs, target, V, C, valid, lut,
                     (unsigned long long*)ws);
  hipLaunchKernelGGL(mt_counts_to_float_kernel, dim3(mt_cdiv(B * C * 3, 256)), dim3(256), 0, st, (const unsigned long long*)ws, stats, B * C * 3);
  MT_CHECK_LAUNCH("multitalent_hard_stats");
  return MT_OK;
}

// ---- softmax Dice + CE -------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void softmax_loss_fwd_kernel(const float* __restrict__ logits, int cs,
                                                               const float* __restrict__ target, long V, int C, int nblk,
                                                               float* __restrict__ part) {
  __shared__ float red[4][MAXC * 3 + 1];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long v0 = (long)blockIdx.x * LS_VB;
  const long v1 = (v0 + LS_VB < V) ? v0 + LS_VB : V;
  float tp[MAXC], fp[MAXC], fn[MAXC], ce = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { tp[c] = 0.f; fp[c] = 0.f; fn[c] = 0.f; }
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const size_t e = (size_t)b * V + v;
    const int lab = (int)target[e];
    float x[MAXC], mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { x[c] = (c < C) ? logits[e * cs + c] : -3.0e38f; mx = fmaxf(mx, x[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { x[c] = (c < C) ? __expf(x[c] - mx) : 0.f; se += x[c]; }
    const float inv = 1.f / se;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < C) {
        const float p = x[c] * inv;
        const float y = (lab == c) ? 1.f : 0.f;
        tp[c] += p * y; fp[c] += p * (1.f - y); fn[c] += (1.f - p) * y;
        if (lab == c) ce -= __logf(fmaxf(p, 1e-38f));
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { tp[c] = mt_wave_sum(tp[c]); fp[c] = mt_wave_sum(fp[c]); fn[c] = mt_wave_sum(fn[c]); }
  ce = mt_wave_sum(ce);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { red[wave][c * 3] = tp[c]; red[wave][c * 3 + 1] = fp[c]; red[wave][c * 3 + 2] = fn[c]; }
    red[wave][MAXC * 3] = ce;
  }
  __syncthreads();
  if (threadIdx.x < C * 4) {
    const int c = threadIdx.x >> 2, k = threadIdx.x & 3;
    float s = 0.f;
    if (k == 0) { if (c == 0) s = red[0][MAXC * 3] + red[1][MAXC * 3] + red[2][MAXC * 3] + red[3][MAXC * 3]; }
    else s = red[0][c * 3 + k - 1] + red[1][c * 3 + k - 1] + red[2][c * 3 + k - 1] + red[3][c * 3 + k - 1];
    part[(((size_t)b * nblk + blockIdx.x) * C + c) * 4 + k] = s;
  }
}

extern "C" int mt_softmax_dice_ce_fwd(const float* logits, int cs, const float* target, int B, long V, int C,
                                      float* stats, void* ws, size_t ws_bytes, mt_stream_t stream) {
  MT_REQUIRE(logits && target && stats && B > 0 && V > 0 && C > 1 && C <= 16, "softmax_dice_ce_fwd: bad args (2 <= C <= 16)");
  if (ws == nullptr || ws_bytes < mt_loss_workspace(B, V, C)) { mt_set_error("softmax_dice_ce_fwd: workspace too small"); return MT_EWORKSPACE; }
  const int nblk = mt_cdiv(V, LS_VB);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 4) hipLaunchKernelGGL(softmax_loss_fwd_kernel<4>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, nblk, (float*)ws);
  else if (C <= 8) hipLaunchKernelGGL(softmax_loss_fwd_kernel<8>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, nblk, (float*)ws);
  else hipLaunchKernelGGL(softmax_loss_fwd_kernel<16>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, nblk, (float*)ws);
  hipLaunchKernelGGL(loss_stats_finalize_kernel, dim3(C * 4, B), dim3(64), 0, st, (const float*)ws, nblk, C, stats);
  MT_CHECK_LAUNCH("softmax_dice_ce_fwd");
  return MT_OK;
}

template <int MAXC>
__global__ __launch_bounds__(256) void softmax_loss_bwd_kernel(const float* __restrict__ logits, int cs,
                                                               const float* __restrict__ target, long V, int C,
                                                               const float* __restrict__ gstats, float* __restrict__ dlogits, int dcs) {
  const int b = blockIdx.y;
  float a_tp[MAXC], a_fp[MAXC], a_fn[MAXC];
  const float ce_coef = gstats[(size_t)b * C * 4];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    a_tp[c] = (c < C) ? gstats[((size_t)b * C + c) * 4 + 1] : 0.f;
    a_fp[c] = (c < C) ? gstats[((size_t)b * C + c) * 4 + 2] : 0.f;
    a_fn[c] = (c < C) ? gstats[((size_t)b * C + c) * 4 + 3] : 0.f;
  }
  const long v0 = (long)blockIdx.x * LS_VB;
  const long v1 = (v0 + LS_VB < V) ? v0 + LS_VB : V;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const size_t e = (size_t)b * V + v;
    const int lab = (int)target[e];
    float x[MAXC], mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { x[c] = (c < C) ? logits[e * cs + c] : -3.0e38f; mx = fmaxf(mx, x[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { x[c] = (c < C) ? __expf(x[c] - mx) : 0.f; se += x[c]; }
    const float inv = 1.f / se;
    float G[MAXC], dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      x[c] *= inv;
      const float y = (lab == c) ? 1.f : 0.f;
      // dL/dp_c through tp, fp, fn:  tp = sum p y, fp = sum p (1-y), fn = sum (1-p) y
      G[c] = y * (a_tp[c] - a_fn[c]) + (1.f - y) * a_fp[c];
      dot += x[c] * G[c];
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < C) {
        const float y = (lab == c) ? 1.f : 0.f;
        dlogits[e * dcs + c] = ce_coef * (x[c] - y) + x[c] * (G[c] - dot);
      }
    }
  }
}

extern "C" int mt_softmax_dice_ce_bwd(const float* logits, int cs, const float* target, int B, long V, int C,
                                      const float* gstats, float* dlogits, int dcs, mt_stream_t stream) {
  MT_REQUIRE(logits && target && gstats && dlogits && B > 0 && V > 0 && C > 1 && C <= 16, "softmax_dice_ce_bwd: bad args");
  const int nblk = mt_cdiv(V, LS_VB);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 4) hipLaunchKernelGGL(softmax_loss_bwd_kernel<4>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, gstats, dlogits, dcs);
  else if (C <= 8) hipLaunchKernelGGL(softmax_loss_bwd_kernel<8>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, gstats, dlogits, dcs);
  else hipLaunchKernelGGL(softmax_loss_bwd_kernel<16>, dim3(nblk, B), dim3(256), 0, st, logits, cs, target, V, C, gstats, dlogits, dcs);
  MT_CHECK_LAUNCH("softmax_dice_ce_bwd");
  return MT_OK;
}

// ---- the few-hundred-float loss combination on top of the statistics, forward AND backward in one launch ---------------------
// loss = sum_l ( ce_coef[l] * sum_{(b,c) in CE set} stats[l,b,c,0]  -  dice_coef[l] * sum_{entries} r ),
// r = (2 tp + smooth_num) / max(2 tp + fp + fn + smooth_den + den_eps, clamp_min), over the channels c >= c0; with
// MT_LOSS_DICE_OVER_BATCH the statistics are summed over b before the ratio (batch_dice of dice_loss.py:150-160).  Covers
// MultiTalent_Trainer_DDP.py:598-623 (clamp 1e-7, no smooth, every (b, c) entry), dice_loss.py:180-183 + deep_supervision.py:37-42
// and nnUNetTrainerV2_DDP.py:262-282.  `dice` = the statistics the ratios are formed from ([L][B][C][dice_stride], tp/fp/fn first):
// the local ones (stats + 1, stride 4) or their sum over ranks (stride 3); gstats = dLoss/d(local stats), what the per-level
// backward kernels take; gscale multiplies the Dice part of it (the world size when `dice` is the sum over ranks: the reference's
// all-gather sums the ranks' identical gradients in its backward, distributed.py:63-73).  One workgroup: L * B * C <= a few thousand entries.
__global__ __launch_bounds__(256) void loss_combine_kernel(const float* __restrict__ stats, const float* __restrict__ dice, int ds,
                                                           int L, int B, int C, const float* __restrict__ ce_coef,
                                                           const float* __restrict__ dice_coef, int flags, int c0,
                                                           float sn, float sd, float eps, float clamp_min, float gscale,
                                                           float* __restrict__ out3, float* __restrict__ gstats) {
  __shared__ double red[2][256];
  const int tid = threadIdx.x;
  const bool all_c = flags & 1, over_b = flags & 2;
  double ce = 0.0, dc = 0.0;
  const int nE = L * B * C;
  for (int e = tid; e < nE; e += 256) {
    const int l = e / (B * C), c = e % C;
    const bool in = all_c || c == 0;
    const float k = in ? ce_coef[l] : 0.f;
    gstats[(long)e * 4] = k;
    if (in) ce += (double)k * (double)stats[(long)e * 4];
    if (c < c0) { gstats[(long)e * 4 + 1] = 0.f; gstats[(long)e * 4 + 2] = 0.f; gstats[(long)e * 4 + 3] = 0.f; }
  }
  const int nb = over_b ? B : 1, Bo = over_b ? 1 : B, Cd = C - c0;
  const int nI = L * Bo * Cd;
  for (int i = tid; i < nI; i += 256) {
    const int l = i / (Bo * Cd), bo = (i / Cd) % Bo, c = c0 + i % Cd;
    float T = 0.f, F = 0.f, N = 0.f;
    for (int b = 0; b < nb; ++b) {
      const float* p = dice + ((long)(l * B + bo + b) * C + c) * ds;
      T += p[0]; F += p[1]; N += p[2];
    }
    const float num = 2.f * T + sn;
    const float raw = 2.f * T + F + N + sd + eps;
    const bool pass = raw >= clamp_min;
    const float den = pass ? raw : clamp_min;
    const float inv = 1.f / den;
    const float k = dice_coef[l];
    dc += (double)k * (double)(num * inv);
    const float gden = pass ? -num * inv * inv : 0.f;
    const float gT = -k * gscale * (2.f * inv + 2.f * gden), gF = -k * gscale * gden;
    for (int b = 0; b < nb; ++b) {
      float* g = gstats + ((long)(l * B + bo + b) * C + c) * 4;
      g[1] = gT; g[2] = gF; g[3] = gF;
    }
  }
  red[0][tid] = ce; red[1][tid] = dc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { out3[0] = (float)(red[0][0] - red[1][0]); out3[1] = (float)red[0][0]; out3[2] = (float)red[1][0]; }
}

extern "C" int mt_loss_combine(const float* stats, const float* dice, int dice_stride, int L, int B, int C, const float* ce_coef,
                               const float* dice_coef, int flags, int c0, float smooth_num, float smooth_den, float den_eps,
                               float clamp_min, float dice_grad_scale, float* out3, float* gstats, mt_stream_t stream) {
  MT_REQUIRE(stats && dice && ce_coef && dice_coef && out3 && gstats && L > 0 && B > 0 && C > 0 && c0 >= 0 && c0 < C && dice_stride >= 3,
             "loss_combine: bad args");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, dice, dice_stride, L, B, C, ce_coef, dice_coef,
                     flags, c0, smooth_num, smooth_den, den_eps, clamp_min, dice_grad_scale, out3, gstats);
  MT_CHECK_LAUNCH("loss_combine");
  return MT_OK;
}
