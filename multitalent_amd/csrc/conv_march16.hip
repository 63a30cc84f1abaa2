// conv_march16.hip — 3x3x3 / 1x3x3 stride-1 convolution over 16-bit tensors (forward over fp16 activations, backward-data over bf16 gradients:
// nn.Conv3d of generic_UNet.py:57,67 and its backward) for layers with at most 64 input channels, round 5.
//
// conv_bf16_kernel, which served these launches, reads one 1-KiB A fragment from LDS and one 1-KiB weight fragment from L1 for EVERY
// v_mfma_f32_32x32x16: its matrix phase sits exactly at the LDS-read bound (128 B/clk for four SIMDs), and staging and epilogue add to it
// (DESIGN 3.3).  This kernel removes both streams from the inner loop:
//   * WEIGHTS IN REGISTERS.  K = 32 of v_mfma_f32_16x16x32 is a PAIR of 16-channel chunks, a wave owns 16 output channels: its 27 weight
//     fragments (108 registers) are loaded once per workgroup and never again;
//   * A REGISTER WINDOW OF THREE OUTPUT PLANES.  A workgroup owns a 4 x 32 (h, w) column and 32 output channels and marches along D: step p
//     brings input plane p, which contributes to the output planes p+1, p, p-1 (kd = 0, 1, 2).  An input operand (halo row u, shift kw: one
//     ds_read_b128) therefore feeds 3 planes x (1 or 2 rows) MFMAs instead of one: LDS reads at a third of the matrix time;
//   * NO REGISTER ROUND TRIP FOR THE INPUT.  The raw planes go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) several steps ahead,
//     into a ring of plane images; one step before its use an image is activated IN PLACE (fp32 scale / shift / LeakyReLU, one rounding);
//   * eight waves, two per SIMD (inside one wave nothing overlaps a 16x16x32 MFMA: tools/ubench/mfma16x16x32_stream.hip): wave
//     (third, rp, coh) owns output rows 2rp, 2rp+1 and the cout half coh; NP = 1 (<= 32 input channels): third = w half (16 voxels);
//     NP = 2 (<= 64): third = chunk pair, both w halves, the two pairs' partial sums meet through LDS one step later;
//   * a finished output plane leaves the accumulators at the end of its last step and is stored (bias, channel-pair exchange, rounding,
//     InstanceNorm statistics of the stored values) during the NEXT step, task by task behind the MFMAs;
//   * the work list of a workgroup is a contiguous range of (column, plane) pairs, laid out per XCD (as conv_bwdw_tr16_kernel).
// One destination, no accumulation (the launches with two destinations or accumulation stay on conv_bf16_kernel).
#include "bwdw_common.h"
#include <atomic>
#include <type_traits>

#define CM_THREADS 512
#define CM_CHB 7168                    // bytes of one chunk's plane image: 448 16-byte pieces, 408 used (6 halo rows x 34 voxels x 2 channel halves)
#define CM_XROW (34 * 32)
#define CM_MAXN 16
#ifndef CM_ABL
#define CM_ABL 0                       // timing ablations: 1 no activation, 2 no DMA, 4 no stores, 8 no MFMAs
#endif
template <int NP> __host__ __device__ constexpr int cm_plb() { return 2 * NP * CM_CHB; }          // bytes of a plane image
template <int NP> __host__ __device__ constexpr int cm_depth() { return NP == 1 ? 6 : 4; }          // plane images in the ring
template <int NP> __host__ __device__ constexpr int cm_tab_off() { return cm_depth<NP>() * cm_plb<NP>(); }
template <int NP> __host__ __device__ constexpr int cm_scr_off() { return cm_tab_off<NP>() + CM_MAXN * NP * 256; }
template <int NP> __host__ __device__ constexpr int cm_sx_off() { return cm_scr_off<NP>() + (NP == 2 ? 2 * 8 * 2 * 1024 : 0); }      // behind the pair exchange buffers [2][8 waves][2 rows][1 KiB]
template <int NP> __host__ __device__ constexpr int cm_dummy_off() { return cm_sx_off<NP>() + 1024; }
template <int NP> __host__ __device__ constexpr int cm_lds_bytes() { return cm_dummy_off<NP>() + 1024; }
int mt_conv_march16_lds_bytes(int npairs) { return npairs == 1 ? cm_lds_bytes<1>() : cm_lds_bytes<2>(); }

typedef _Float16 cm_f16x8 __attribute__((ext_vector_type(8)));
template <int MTY> __device__ __forceinline__ f32x4 cm_mfma(const bwb_bf16x8 a, const bwb_bf16x8 b, const f32x4 c) {
  if constexpr (MTY == MT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cm_f16x8, a), __builtin_bit_cast(cm_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// channel-pair exchange of conv_bf16.inc (mt_pair_exchange), restated for the 16x16 accumulator tile: lanes (co even, co odd) trade one value
// per two accumulator rows; the even lane ends with both channels of voxel j, the odd lane with both of voxel j + 1
__device__ __forceinline__ void cm_pair_exchange(float vj, float vj1, bool odd, float& a, float& b) {
  const float send = odd ? vj : vj1;
  const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  a = odd ? recv : vj;
  b = odd ? vj1 : recv;
}

// position in a workgroup's work list: planes [da, db) of column col; p = input plane of the step (da - LEAD .. db - 1 + LEAD)
struct CmCursor {
  int col, p, da, db, on, left;
};

template <int KD, int NP, int XS>
__global__ __launch_bounds__(CM_THREADS) void conv_march16_kernel(const ConvKParams P) {
  static_assert(XS == MT_F16 || XS == MT_BF16, "16-bit storage");
  static_assert(KD == 1 || KD == 3, "3x3x3 or 1x3x3");
  constexpr int LEAD = KD == 3 ? 1 : 0, NTAP = KD * 9, NWH = NP, D = cm_depth<NP>(), PLB = cm_plb<NP>(), KDMA = 2 * NP;
  constexpr int NPIECE = 2 * NP;                   // activation pieces per thread and plane (the last one partly)
  extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
  char* const L = (char*)ldsw;
  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int coh = wave & 1, rp = (wave >> 1) & 1, third = wave >> 2;
  const int pair = NP == 2 ? third : 0;
  const int cot = blockIdx.y;
  const int TW_ = P.tilesW, TH_ = P.tilesH;

  // ---- work range
  const int cols = c.N * TH_ * TW_;
  const long T = (long)cols * c.Do;
  const int wg = mt_xcd_remap(blockIdx.x, gridDim.x);
  const long t0 = T * wg / gridDim.x, t1 = T * (wg + 1) / gridDim.x;
  auto seg_open = [&](CmCursor& q, int col, int da, int left) {
    q.col = col; q.da = da;
    const int room = c.Do - da;
    q.db = da + (left < room ? left : room);
    q.p = da - LEAD;
    q.left = left - (q.db - da);
  };
  auto cur_next = [&](CmCursor& q) -> bool {     // returns true when the column changed
    if (!q.on) return false;
    if (q.p < q.db - 1 + LEAD) { q.p += 1; return false; }
    if (q.left > 0) { seg_open(q, q.col + 1, 0, q.left); return true; }
    q.on = 0;
    return false;
  };
  CmCursor F, A, M;
  F.on = t0 < t1 ? 1 : 0; F.col = 0; F.p = 0; F.da = 0; F.db = 0; F.left = 0;
  if (F.on) seg_open(F, (int)(t0 / c.Do), (int)(t0 % c.Do), (int)(t1 - t0));
  A = F; M = F;
  int nsteps = 0;
  {
    long t = t0;
    while (t < t1) { const long room = c.Do - t % c.Do; const long n = (t1 - t < room) ? t1 - t : room; nsteps += (int)n + 2 * LEAD; t += n; }
  }

  // ---- weights: 27 fragments of this wave's (pair, cout half), pack layout 3 / 4 ([cout tile][chunk][tap][lane 64][8 x 16 bit]: lane L holds
  // input channels 8 (L >> 5) .. +7 of the chunk for output channel L & 31)
  bwb_bf16x8 wreg[NTAP];
  {
    const int chunk = 2 * pair + (lg >> 1);
    const int Lsrc = (lg & 1) * 32 + 16 * coh + li;
    const uint4* wp = (const uint4*)c.wpack + ((size_t)(cot * P.nchunks + chunk) * NTAP) * 64 + Lsrc;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      uint4 v = uint4{0, 0, 0, 0};
      if (chunk < P.nchunks) v = wp[(size_t)t * 64];
      wreg[t] = __builtin_bit_cast(bwb_bf16x8, v);
    }
  }

  // ---- activation table [sample][chunk of the workgroup][scale | shift][16]
  float* const TAB = (float*)(L + cm_tab_off<NP>());
  for (int i = tid; i < c.N * NP * 64; i += CM_THREADS) {
    const int ch = i & 15, sel = (i >> 4) & 1, cq = (i >> 5) % (2 * NP), nb = i / (64 * NP);
    float v = 0.f;
    if (cq < P.nchunks) {
      const ConvChunk k_ = P.chunk[cq];
      const mt_src_t& Sq = c.src[k_.src];
      if (ch < k_.ck) v = Sq.scale != nullptr ? (sel ? Sq.shift[(size_t)nb * Sq.C + k_.c0 + ch] : Sq.scale[(size_t)nb * Sq.C + k_.c0 + ch]) : (sel ? 0.f : 1.f);
    }
    TAB[i] = v;
  }
  float slope_c[2 * NP];                           // LeakyReLU slope per chunk (1: identity source)
#pragma unroll
  for (int q = 0; q < 2 * NP; ++q) {
    slope_c[q] = 1.f;
    if (q < P.nchunks) { const mt_src_t& Sq = c.src[P.chunk[q].src]; if (Sq.scale != nullptr) slope_c[q] = Sq.slope; }
  }

  // ---- DMA roles: wave w issues blocks m = w + 8 k (k < KDMA) of the plane's 14 NP blocks of 64 pieces (7 per chunk); blocks past the end
  // go to a dummy area with all lanes out of range (every wave issues the same number of instructions: the waits below count them)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)ldsw;
  int ring_f = 0;
  auto wrap = [&](int& r) { r += PLB; if (r >= D * PLB) r = 0; };
  const int Di_ = __builtin_amdgcn_readfirstlane(c.Di);
  int dvo[KDMA];                                   // byte offset of the lane's piece inside a plane of its chunk's source (or out of range)
  mt_i32x4 drs[KDMA];
  int dxp[KDMA];                                   // bytes of a plane of the instruction's source
  unsigned dlds[KDMA];
  auto dma_setup = [&]() {                         // F.col changed (or first use)
    int r_ = F.col;
    const int tw = r_ % TW_; r_ /= TW_;
    const int th = r_ % TH_;
    const int nb = r_ / TH_;
    const int uh0 = th * 4 - 1, uw0 = tw * 32 - 1;
#pragma unroll
    for (int k = 0; k < KDMA; ++k) {
      const int m = wave + 8 * k;
      const int cidx = m / 7, j = (m % 7) * 64 + lane;
      dvo[k] = (int)0x80000000;
      dlds[k] = (unsigned)cm_dummy_off<NP>();
      drs[k] = mt_i32x4{0, 0, 0, 0x00020000};
      dxp[k] = 0;
      if (m < 14 * NP) {
        dlds[k] = (unsigned)(cidx * CM_CHB + (m % 7) * 1024);
        if (cidx < P.nchunks) {
          const ConvChunk k_ = P.chunk[cidx];
          const mt_src_t& Sq = c.src[k_.src];
          const size_t xsample = (size_t)c.Di * c.Hi * c.Wi * Sq.cs * 2;
          dxp[k] = __builtin_amdgcn_readfirstlane(c.Hi * c.Wi * Sq.cs * 2);
          const size_t ba = (size_t)Sq.ptr + (size_t)nb * xsample;          // raw buffer descriptor: base, no stride, num_records in bytes
          drs[k] = mt_i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)ba), __builtin_amdgcn_readfirstlane((int)(unsigned)((ba >> 32) & 0xffffu)),
                            __builtin_amdgcn_readfirstlane((int)xsample), 0x00020000};
          const int vx = j >> 1, hs = j & 1, row = vx / 34, cx = vx - row * 34, hx = hs ^ ((cx >> 3) & 1);
          const int uh = uh0 + row, uw = uw0 + cx;
          const bool ok = (j < 408) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi) && (8 * hx < k_.ck);
          if (ok) dvo[k] = ((uh * c.Wi + uw) * Sq.cs + k_.c0 + 8 * hx) * 2;
        }
      }
    }
  };
  auto dma_issue = [&]() {                         // the plane of cursor F into ring slot ring_f; advances F and the slot
    const int p = F.p;
    const bool pv = F.on && (unsigned)p < (unsigned)Di_;
    const unsigned base = (unsigned)ring_f;
    if (!(CM_ABL & 2)) {
#pragma unroll
      for (int k = 0; k < KDMA; ++k) {
        const int m = wave + 8 * k;
        const int vo = pv ? dvo[k] + p * dxp[k] : (int)0x80000000;
        mt_lds_dma16(drs[k], (dvo[k] < 0) ? (int)0x80000000 : vo, lds_base + (m < 14 * NP ? base : 0u) + dlds[k]);
      }
    }
    if (cur_next(F)) dma_setup();
    wrap(ring_f);
  };

  // ---- activation roles: thread t activates pieces t + 512 k of the plane's 816 NP pieces (chunk = piece / 408), in place
  unsigned avm = 0;                                // bit k: piece k is a voxel inside the tensor (and a channel half of its chunk)
  int anb = 0;
  auto act_setup = [&]() {                         // A.col changed
    int r_ = A.col;
    const int tw = r_ % TW_; r_ /= TW_;
    const int th = r_ % TH_;
    anb = r_ / TH_;
    const int uh0 = th * 4 - 1, uw0 = tw * 32 - 1;
    avm = 0;
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int pc = tid + CM_THREADS * k, cidx = pc / 408, j = pc - cidx * 408;
      const int vx = j >> 1, row = vx / 34, cx = vx - row * 34, hx = (j & 1) ^ ((cx >> 3) & 1);
      const int uh = uh0 + row, uw = uw0 + cx;
      const bool ok = (pc < 816 * NP) && (cidx < P.nchunks) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi) &&
                      (8 * hx < (cidx < P.nchunks ? P.chunk[cidx].ck : 0));
      avm |= (ok ? 1u : 0u) << k;
    }
  };
  int apo[NPIECE], atb[NPIECE];                    // byte offset of the piece inside a plane image; float index of its scale row in a sample's table
  float asl[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int pc = tid + CM_THREADS * k, cidx = pc / 408, j = pc - cidx * 408;
    const int vx = j >> 1, row = vx / 34, cx = vx - row * 34, hx = (j & 1) ^ ((cx >> 3) & 1);
    apo[k] = pc < 816 * NP ? cidx * CM_CHB + j * 16 : -1;
    atb[k] = (cidx < 2 * NP ? cidx : 0) * 32 + 8 * hx;
    asl[k] = 1.f;
#pragma unroll
    for (int q = 0; q < 2 * NP; ++q) if (q == cidx) asl[k] = slope_c[q];
  }
  // activation of one piece: the plain form (prologue) and the task form below share these
  auto act_piece = [&](int k, int ring, unsigned vm_, int nb_, bool pv) {
    if (apo[k] < 0) return;
    uint4* pp = (uint4*)(L + ring + apo[k]);
    const uint4 raw = *pp;
    const float4* tb = (const float4*)(TAB + (size_t)nb_ * NP * 64 + atb[k]);
    const float4 s0 = tb[0], s1 = tb[1], h0 = tb[4], h1 = tb[5];
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    const unsigned m = (pv && ((vm_ >> k) & 1u)) ? 0xffffffffu : 0u;
    const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
    uint4 o; unsigned od[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (CM_ABL & 1) od[d] = rw[d] & m;
      else {
        const float a = fmaf(mt_lo16<XS>(rw[d]), sc[2 * d], sh[2 * d]), b = fmaf(mt_hi16<XS>(rw[d]), sc[2 * d + 1], sh[2 * d + 1]);
        od[d] = mt_pk16<XS>(fmaxf(a, a * asl[k]), fmaxf(b, b * asl[k])) & m;
      }
    }
    o.x = od[0]; o.y = od[1]; o.z = od[2]; o.w = od[3];
    *pp = o;
  };

  // ---- operand addresses: lane (voxel li, k-group lg: chunk lg >> 1 of the pair, channel half lg & 1) of w half q under shift kw
  int aoff[NWH][3];
#pragma unroll
  for (int q = 0; q < NWH; ++q)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wh = NP == 1 ? third : q;
      const int cx = 16 * wh + li + kw;
      aoff[q][kw] = (2 * pair + (lg >> 1)) * CM_CHB + rp * 2 * CM_XROW + cx * 32 + (((lg & 1) ^ ((cx >> 3) & 1)) * 16);
    }

  // ---- output side
  const size_t out_sample = (size_t)c.Do * c.Ho * c.Wo * c.ocs0 * 2;
  const int co2 = cot * 32 + coh * 16 + (li & ~1);
  const bool odd = li & 1;
  // bias: the C operand of a plane's first MFMA (pair 0 only: the pairs' sums are added)
  const float bv = (c.bias != nullptr && pair == 0 && cot * 32 + coh * 16 + li < c.Cout) ? c.bias[cot * 32 + coh * 16 + li] : 0.f;
  f32x4 acc[3][2][NWH];                            // [slot = step of the plane's first contribution % 3][row][w half]
  f32x4 outv[2];                                   // the finished plane's tiles this wave stores: rows 0, 1 of its w half
  int st_valid = 0, st_nb = -1, st_oh0 = 0, st_ow0 = 0;      // the plane in outv
  float q1[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};    // statistics of the stored values: channels co2, co2 + 1 over this lane's voxels
  int stat_nb = -1;                                // sample the running statistics belong to
  unsigned flushed = 0;                            // samples whose partial row this workgroup has written
  float* const SCR = (float*)(L + cm_scr_off<NP>());
  int step_no = 0;
  int ring_m = 0, ring_a = PLB;                  // byte offsets of the ring slots of the step being multiplied / activated (ring_f: fetched) — wrapped adds, no division
  const int own_q = NP == 2 ? pair : 0;            // the w half (index into acc[..][q]) this wave finishes

  // statistics of a sample: lanes -> channel owner -> the four lane groups -> the waves of the cout half; one row per (sample, workgroup)
  auto flush_stats = [&](int nb) {                 // executed by ALL waves at the same step
    float s1 = 0.f, s2 = 0.f;
    {
      const float t0_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1[0]), 0xB1, 0xF, 0xF, true));
      const float t1_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1[1]), 0xB1, 0xF, 0xF, true));
      const float u0_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q2[0]), 0xB1, 0xF, 0xF, true));
      const float u1_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q2[1]), 0xB1, 0xF, 0xF, true));
      s1 = odd ? q1[1] + t1_ : q1[0] + t0_;
      s2 = odd ? q2[1] + u1_ : q2[0] + u0_;
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    float* sx = (float*)(L + cm_sx_off<NP>());                 // 8 waves x 16 channels x 2
    __syncthreads();
    if (lg == 0) { sx[(wave * 16 + li) * 2] = s1; sx[(wave * 16 + li) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 32 && c.stats_part != nullptr && nb >= 0) {
      const int h = tid >> 4, ch = tid & 15, co = cot * 32 + h * 16 + ch;
      if (co < c.Cout) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) if ((w & 1) == h) { a += sx[(w * 16 + ch) * 2]; b += sx[(w * 16 + ch) * 2 + 1]; }
        float* sp = c.stats_part + ((size_t)((size_t)nb * P.nsb + blockIdx.x) * c.Cout + co) * 2;
        sp[0] = a; sp[1] = b;
      }
    }
    if (nb >= 0) flushed |= 1u << nb;
    q1[0] = q1[1] = q2[0] = q2[1] = 0.f;
    __syncthreads();
  };

  // store tasks of the plane in outv: tile (row r): part 0 exchange (+ the other pair's partial sums), part 1 rounding + stores + statistics.
  // Everything that depends on the column only is a scalar refreshed when the M cursor changes column; the lane part of a store offset
  // is a constant of the thread.
  const int Ho_ = __builtin_amdgcn_readfirstlane(c.Ho), Wo_ = __builtin_amdgcn_readfirstlane(c.Wo);
  const int orow_ = __builtin_amdgcn_readfirstlane(c.Wo * c.ocs0 * 2), oplane_ = __builtin_amdgcn_readfirstlane(c.Ho * c.Wo * c.ocs0 * 2);
  int m_oh0 = 0, m_ow0 = 0, m_nb = 0;              // of cursor M's column
  auto m_setup = [&]() {
    int r_ = M.col;
    const int tw = r_ % TW_; r_ /= TW_;
    m_oh0 = (r_ % TH_) * 4; m_ow0 = tw * 32; m_nb = r_ / TH_;
  };
  const int wh_own = NP == 1 ? third : own_q;
  int lofs[2], lcol[2];                            // byte offset / column inside the tile row of the lane's two stores
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    lcol[jj] = 16 * wh_own + 4 * lg + 2 * jj + (odd ? 1 : 0);
    lofs[jj] = (co2 + 1 < c.Cout) ? (lcol[jj] * c.ocs0 + co2) * 2 : (int)0x80000000;
  }
  __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc((void*)c.out0, 0, 0, 0x00020000);
  int st_base = 0;                                 // byte offset of (plane, tile row 0, column ow0) of the plane in outv
  float e_a[2], e_b[2];
  auto store_task = [&](int r, int part) {
    if (CM_ABL & 4) return;
    if (part == 0) {
      if constexpr (NP == 2) {                     // the other pair's partial sums of this tile, written at the end of the previous step
        const float4 o = *(const float4*)(SCR + (((((step_no + 1) & 1) * 8 + (wave ^ 4)) * 2 + r) * 64 + lane) * 4);
        outv[r][0] += o.x; outv[r][1] += o.y; outv[r][2] += o.z; outv[r][3] += o.w;
      }
      cm_pair_exchange(outv[r][0], outv[r][1], odd, e_a[0], e_b[0]);
      cm_pair_exchange(outv[r][2], outv[r][3], odd, e_a[1], e_b[1]);
    } else {
      const int rr = 2 * rp + r;
      const bool rowok = st_valid && (st_oh0 + rr < Ho_);      // scalar
      const int soff = st_base + rr * orow_;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const bool ok = !(CM_ABL & 16) && rowok && (st_ow0 + lcol[jj] < Wo_) && lofs[jj] >= 0;
        const unsigned pk = mt_pk16<XS>(e_a[jj], e_b[jj]);
        __builtin_amdgcn_raw_buffer_store_b32(pk, st_rs, ok ? lofs[jj] : (int)0x80000000, soff, 0);
        const unsigned pm = ok ? pk : 0u;          // statistics of the values as stored (nothing for a voxel outside the tensor)
        const float ar = mt_lo16<XS>(pm), br = mt_hi16<XS>(pm);
        q1[0] += ar; q2[0] = fmaf(ar, ar, q2[0]);
        q1[1] += br; q2[1] = fmaf(br, br, q2[1]);
      }
    }
  };

  // ---- one step (PH = step % 3 fixes the accumulator slots; every LDS address has a dynamic ring base)
  bwb_bf16x8 xop[3];
  auto step = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    constexpr int NSLOT = NWH * KD * 18;           // MFMAs of a wave and step
    constexpr int NST = 4;                         // store instructions of a wave and step (two tiles x two voxel pairs)
    const int s = step_no;
    const char* const xb = L + ring_m;
    // the step's own bookkeeping (all scalar): statistics flush on a sample change of the plane about to be stored
    if (st_valid && st_nb != stat_nb) { if (stat_nb >= 0) flush_stats(stat_nb); stat_nb = st_nb; }
    const bool a_pv = A.on && (unsigned)A.p < (unsigned)Di_;
    const unsigned a_vm = avm; const int a_nb = anb;
    // tasks: activation of plane s + 1 (13 per piece: fetch, 12 thirds are folded into 4 dword tasks here), the stores of the finished plane, the DMA
    uint4 t_raw; unsigned t_od[4]; float2 t_sc, t_sh;     // (scale / shift of a dword's two channels are requested one task ahead of their use)
    auto task = [&](int q) {
      constexpr int NACT = NPIECE * 5;
      if (q < NACT) {
        if (CM_ABL & 2) return;
        const int k = q / 5, part = q % 5;
        if (apo[k] < 0) return;
        uint4* pp = (uint4*)(L + ring_a + apo[k]);
        const float2* tb = (const float2*)(TAB + (size_t)a_nb * NP * 64 + atb[k]);
        if (part == 0) { t_raw = *pp; t_sc = tb[0]; t_sh = tb[8]; }
        else {
          const int d = part - 1;
          const unsigned rw = d == 0 ? t_raw.x : d == 1 ? t_raw.y : d == 2 ? t_raw.z : t_raw.w;
          const float2 sc2 = t_sc, sh2 = t_sh;
          if (d < 3) { t_sc = tb[d + 1]; t_sh = tb[8 + d + 1]; }
          const unsigned m = (a_pv && ((a_vm >> k) & 1u)) ? 0xffffffffu : 0u;
          if (CM_ABL & 1) t_od[d] = rw & m;
          else {
            const float a = fmaf(mt_lo16<XS>(rw), sc2.x, sh2.x), b = fmaf(mt_hi16<XS>(rw), sc2.y, sh2.y);
            t_od[d] = mt_pk16<XS>(fmaxf(a, a * asl[k]), fmaxf(b, b * asl[k])) & m;
          }
          if (d == 3) { uint4 o; o.x = t_od[0]; o.y = t_od[1]; o.z = t_od[2]; o.w = t_od[3]; *pp = o; }
        }
      } else if (q < NACT + 4) {
        const int e = q - NACT;
        store_task(e >> 1, e & 1);
      } else if (q == NACT + 4) {
        dma_issue();
      }
    };
    constexpr int NTASKS = NPIECE * 5 + 5;
    static_assert(NTASKS <= NSLOT - 6 || KD == 1, "tasks must fit in front of the barrier");
    constexpr int TPS = KD == 3 ? 1 : (NTASKS + NSLOT - 7) / (NSLOT - 6);
    // operand ring: group g = (xr, kw, q) in xop[g % 3], requested two groups ahead
    auto opnd = [&](int g) -> bwb_bf16x8 {
      const int q = g % NWH, kw = (g / NWH) % 3, xr = g / (3 * NWH);
      return *(const bwb_bf16x8*)(xb + aoff[q][kw] + xr * CM_XROW);
    };
    constexpr int NG = 12 * NWH;
    xop[0] = opnd(0); xop[1] = opnd(1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 2 < NG) xop[(g + 2) % 3] = opnd(g + 2);
      __builtin_amdgcn_sched_barrier(0);
      const int q = g % NWH, kw = (g / NWH) % 3, xr = g / (3 * NWH);
      const int nr = (xr == 0 || xr == 3) ? 1 : 2;                 // output rows this halo row serves
      const int s0 = KD * ((xr == 0 ? 0 : xr == 1 ? 1 : xr == 2 ? 3 : 5) * 3 * NWH + (kw * NWH + q) * nr);      // MFMAs in front of this group
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int kh = xr - r;
        if (kh < 0 || kh > 2) continue;
        const int ri = (xr == 0 || xr == 3) ? 0 : r;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
          const int sl = KD == 3 ? (PH + 3 - kd) % 3 : 0;      // the plane whose kd-th plane of taps this is started kd steps ago
          const int slot = s0 + ri * KD + kd;                  // a constant after unrolling
          if (!(CM_ABL & 8)) {
            if (kd == 0 && kh == 0 && kw == 0) acc[sl][r][q] = cm_mfma<XS>(xop[g % 3], wreg[(kd * 3 + kh) * 3 + kw], f32x4{bv, bv, bv, bv});
            else acc[sl][r][q] = cm_mfma<XS>(xop[g % 3], wreg[(kd * 3 + kh) * 3 + kw], acc[sl][r][q]);
          }
#pragma unroll
          for (int tq = 0; tq < TPS; ++tq)
            if (slot * TPS + tq < NTASKS) task(slot * TPS + tq);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the plane that took its last contribution in this step (KD = 3: started two steps ago) leaves the accumulators
    {
      constexpr int sl = KD == 3 ? (PH + 1) % 3 : 0;
      const int q_od = M.p - LEAD;                 // output plane completed by input plane M.p
      st_valid = (M.on && q_od >= M.da && q_od < M.db) ? 1 : 0;
      st_oh0 = m_oh0; st_ow0 = m_ow0;
      if (st_nb != m_nb || step_no == 0) {
        st_nb = m_nb;
        st_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)c.out0 + (size_t)m_nb * out_sample), 0, (int)out_sample, 0x00020000);
      }
      st_base = q_od * oplane_ + m_oh0 * orow_ + m_ow0 * c.ocs0 * 2;
      if constexpr (NP == 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const f32x4 o = acc[sl][r][1 - own_q];
          *(float4*)(SCR + ((((s & 1) * 8 + wave) * 2 + r) * 64 + lane) * 4) = float4{o[0], o[1], o[2], o[3]};
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) outv[r] = acc[sl][r][own_q];
    }
    if (cur_next(M)) m_setup();
    if (cur_next(A)) act_setup();
    step_no = s + 1;
    wrap(ring_m); wrap(ring_a);
    // the raw plane of step s + 2 must be complete — this wave's part — before the barrier publishes it.  vmcnt counts every vector-memory
    // instruction in issue order, stores included: behind the DMAs of step s + 2 (issued D - 3 steps ago, after that step's stores) come D - 3
    // steps of NST stores + KDMA DMAs each, and exactly those may still be in flight (a smaller count would drain the ring: the first
    // form waited with (D - 3) KDMA and the stores alone cost 165 of 310 us).  The store tasks issue their instruction on every step
    // (out-of-range offsets for an invalid plane), so the count holds from step 0 on.
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 3) * (KDMA + NST)) : "memory");
    __syncthreads();
  };

  // ---- prologue
  __syncthreads();                                 // the activation table
  if (nsteps > 0) {
    dma_setup();
    act_setup();
    m_setup();
#pragma unroll
    for (int t = 0; t < D - 1; ++t) dma_issue();   // planes of steps 0 .. D - 2 (ring_f ends at slot D - 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // all of them (once per workgroup: the steady-state count below needs no special cases)
    __syncthreads();
    {                                              // plane 0 activated in place, plainly
      const bool pv = A.on && (unsigned)A.p < (unsigned)Di_;
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) act_piece(k, 0, avm, anb, pv);
      if (cur_next(A)) act_setup();
    }
    __syncthreads();
    for (int s = 0; s < nsteps + 1; s += 3) {      // (+ 1: the last plane is stored during the step after its last contribution)
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
    }
  }
  flush_stats(stat_nb);
  // partial rows of the samples this workgroup never stored to: zeros
  if (c.stats_part != nullptr && tid < 32) {
    const int co = cot * 32 + tid;
    if (co < c.Cout)
      for (int nb = 0; nb < c.N; ++nb)
        if (!((flushed >> nb) & 1u)) { float* sp = c.stats_part + ((size_t)((size_t)nb * P.nsb + blockIdx.x) * c.Cout + co) * 2; sp[0] = 0.f; sp[1] = 0.f; }
  }
}

int mt_launch_conv_march16(const ConvKParams& P, int nwg, hipStream_t st) {
  const int xdt = P.c.src[0].dtype, KD = P.c.KD, NP = (P.nchunks + 1) / 2;
  MT_REQUIRE((KD == 1 || KD == 3) && (xdt == MT_F16 || xdt == MT_BF16) && (NP == 1 || NP == 2) && P.c.N <= CM_MAXN,
             "conv3d (march16): KD %d / storage type %d / %d chunks / %d samples", KD, xdt, P.nchunks, P.c.N);
  void (*kfn)(const ConvKParams) = nullptr;
  int idx = 0;
#define CM_PICK(KD_, NP_, XS_, I_) if (KD == KD_ && NP == NP_ && xdt == XS_) { kfn = conv_march16_kernel<KD_, NP_, XS_>; idx = I_; }
  CM_PICK(3, 1, MT_F16, 0) CM_PICK(3, 1, MT_BF16, 1) CM_PICK(3, 2, MT_F16, 2) CM_PICK(3, 2, MT_BF16, 3)
  CM_PICK(1, 1, MT_F16, 4) CM_PICK(1, 1, MT_BF16, 5) CM_PICK(1, 2, MT_F16, 6) CM_PICK(1, 2, MT_BF16, 7)
#undef CM_PICK
  const int ldsb = mt_conv_march16_lds_bytes(NP);
  static std::atomic<uint64_t> attr_s[8];
  const int devid = mt_current_device();
  if (mt_device_pending(attr_s[idx], devid)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    if (e != hipSuccess) { mt_set_error("conv3d: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
    mt_mark_device_done(attr_s[idx], devid);
  }
  hipLaunchKernelGGL(kfn, dim3(nwg, mt_cdiv(P.c.Cout, 32), 1), dim3(CM_THREADS), ldsb, st, P);
  MT_CHECK_LAUNCH("conv_march16");
  return MT_OK;
}
