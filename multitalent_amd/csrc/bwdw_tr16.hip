// bwdw_tr16.hip — backward-weight of 3x3x3 / 1x3x3 stride-1 convolutions in mixed precision WITHOUT a Winograd transform (round 5).
//
//   dW[co][ci][kd][kh][kw] = sum over voxels v of dY[v][co] * act(X)[v + tap][ci]          (backward of nn.Conv3d, generic_UNet.py:57,67)
//
// The contraction index is the voxel, NDHWC keeps the channels fastest — the opposite of what a bf16 MFMA operand wants (a lane holds
// 8 consecutive K values of ONE row).  conv_bwdw_wino_bf16s_kernel escaped that through the Winograd domain (operands produced by
// transform threads: 334 VALU + 94 LDS + 24 MFMA instructions per plane, 0.16 of the bf16 matrix peak).  gfx950's LDS transpose read
// removes the problem at its root: ds_read_b64_tr_b16 hands lane (channel i, k-group g) the four values [voxel 4g..4g+3][channel i] out of
// a CHANNEL-fastest 16-bit image, so the raw NDHWC planes — X activated once (fp32 scale / shift / LeakyReLU, one rounding to bf16), dY
// copied — ARE the operand images, for X under every tap shift: a tap is an immediate byte offset, no transform, no gather.
//
// Work decomposition (one workgroup of EIGHT waves per CU, two per SIMD):
//   * a workgroup owns 32 input channels (a PAIR of 16-channel chunks), 32 output channels and a 4 x 32 (h, w) column, and marches along D;
//     wave (cih, rp, coh) owns chunk cih, the output rows 2rp, 2rp+1 of the column and the cout half coh: K (voxels) is split over the row
//     pairs, every wave accumulates all 27 taps of its 16 ci x 16 co block — 27 accumulator tiles, 108 registers.
//     Why two waves per SIMD and not one with twice the tile (the first form of this kernel: 216 accumulator registers, measured 241 us for
//     30->30 @ 2x48x192x192): INSIDE one wave nothing overlaps a v_mfma_f32_16x16x32_bf16 — tools/ubench/mfma16x16x32_stream.hip: a bare
//     stream runs at 16.2 clocks per MFMA (0.86 of the 2.5 PFLOP/s figure), with the step's 32 transpose reads interleaved 26.1, with three
//     vector instructions behind every MFMA 25.3, with both 38 — the phases of a wave ADD (ablation of that kernel: MFMA stream 130 us +
//     staging 76 us + fixed 42 us).  A second wave on the SIMD hides them: reads cost nothing (16.2 again), vector work only its issue slots;
//   * step p brings X plane p and dY plane p+1.  The dY operands of the wave's rows live in a REGISTER window of three planes
//     (p+1, p, p-1 <-> kd = 0, 1, 2), so an X operand (halo row u, shift kw: 2 transpose reads) feeds 3 planes x (1 or 2 rows) MFMAs:
//     54 v_mfma_f32_16x16x32_bf16 per step and wave for 28 transpose reads;
//   * both tensors are read from HBM once (X with its 6/4 x 34/32 halo, shared between neighbouring columns through the XCD's L2): the
//     workgroup sees every input channel pair-chunk it owns and all 32 couts at once;
//   * the three phases rotate statically (template parameter PH = step mod 3): register window slot, raw register set (global loads
//     three steps ahead of their use) and LDS image all have index step mod 3, so every LDS address is lane base + immediate;
//   * the work list of a workgroup is a contiguous range of (column, plane) pairs in linear order — perfectly balanced for any volume,
//     a new column only zeroes the register window — and the ranges are laid out per XCD (mt_xcd_remap).
// One barrier per step.  Partials [chunk][cot][workgroup][tap][16 ci][32 co] -> bwdw_reduce_kernel as for every backward-weight kernel.
#include "bwdw_common.h"
#include <atomic>
#include <type_traits>

#define BWT_THREADS 512
#define BWT_XROW (34 * 32)             // bytes of one halo row of a chunk image: 34 voxels x 16 channels x 2 bytes
#define BWT_XCH (6 * BWT_XROW)         // a chunk's plane: 6 halo rows
#define BWT_XBUF (2 * BWT_XCH)         // both chunks
#define BWT_YROW (32 * 32)             // one row of one cout half: 32 voxels x 16 channels x 2 bytes
#define BWT_YHALF (4 * BWT_YROW)
#define BWT_YSLOT (2 * BWT_YHALF)
#define BWT_YI_OFF (3 * BWT_XBUF)
#define BWT_IMG_BYTES (3 * BWT_XBUF + 3 * BWT_YSLOT)      // 39 168 + 24 576 = 63 744 (every offset fits the 16-bit immediate of a DS instruction)
#define BWT_MAXN 16                    // samples per launch (host-checked): table [sample][chunk of the pair][scale | shift][16 channels] behind the images
#define BWT_LDS_BYTES (BWT_IMG_BYTES + BWT_MAXN * 256)
#ifndef BWT_EARLY
#define BWT_EARLY 1                    // KD = 3: two tasks per slot and the fetch behind MFMA 12 (0: one per slot, fetch behind MFMA 30)
#endif
#ifndef BWT_ABL
#define BWT_ABL 0                      // timing ablations: 1 no X activation (copy), 2 no staging at all, 4 no global loads (stale registers), 8 no MFMAs
#endif
typedef short bwt_s4 __attribute__((ext_vector_type(4)));

// one MFMA operand: 32 consecutive voxels x 16 channels out of a channel-fastest image (32 bytes per voxel), lane (i = channel, g = k-group).
// K index 8g + j <-> voxel 4g + (j & 3) + 16 (j >> 2): the two half-wave groups of a read cover 256 contiguous bytes (conflict-free)
__device__ __forceinline__ bwb_bf16x8 bwt_operand(const char* lane_base, int off) {
  const bwt_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bwt_s4 __attribute__((address_space(3)))*)(lane_base + off));
  const bwt_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bwt_s4 __attribute__((address_space(3)))*)(lane_base + off + 512));
  return __builtin_bit_cast(bwb_bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int KD, int XS>
__global__ __launch_bounds__(BWT_THREADS) void conv_bwdw_tr16_kernel(const BwdWParams P) {
  static_assert(XS == MT_F16 || XS == MT_BF16, "16-bit X");
  static_assert(KD == 1 || KD == 3, "3x3x3 or 1x3x3");
  constexpr int LEAD = KD == 3 ? 1 : 0;          // step p brings dY plane p + LEAD
  constexpr int NT = KD * 9;
  extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
  char* const L = (char*)ldsw;
  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int cih = wave >> 2, rp = (wave >> 1) & 1, coh = wave & 1;
  const int cot = blockIdx.y, pz = blockIdx.z;
  const int chi = 2 * pz + cih;
  const bool chv = chi < P.nchunks;               // (odd chunk count: the second half of the last pair multiplies zeros)
  const ConvChunk cc = P.chunk[chv ? chi : 2 * pz];
  const int ck = chv ? cc.ck : 0;
  const mt_src_t& Y = P.y;
  const mt_src_t& S = c.src[cc.src];

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  bwb_bf16x8 win[3][2];                           // dY operands [plane slot = step % 3][row of the pair] (cout half coh)

  // ---- this workgroup's range of (column, plane) pairs
  const int cols = c.N * P.tilesH * P.tilesW;
  const long T = (long)cols * c.Do;
  const int wg = mt_xcd_remap(blockIdx.x, gridDim.x);
  const long t0 = T * wg / gridDim.x, t1 = T * (wg + 1) / gridDim.x;

  const size_t xsample = (size_t)c.Di * c.Hi * c.Wi * S.cs * 2, ysample = (size_t)c.Do * c.Ho * c.Wo * Y.cs * 2;      // bytes
  const int xplane_bytes = __builtin_amdgcn_readfirstlane(c.Hi * c.Wi * S.cs * 2);
  const int yplane_bytes = __builtin_amdgcn_readfirstlane(c.Ho * c.Wo * Y.cs * 2);
  const float xslope = S.scale != nullptr ? S.slope : 1.f;

  // ---- staging roles.  X: waves 4c .. 4c+3 stage chunk c: piece i = (tid & 255) + 256 k (k = 0, 1) of the chunk plane's 408 16-byte pieces
  // (voxel i >> 1, channel half i & 1 = tid & 1).  dY: piece tid of 512, LDS order [cout half][row][voxel][channel quarter & 1].
  const int hx = tid & 1;
  int xlo[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) xlo[k] = cih * BWT_XCH + ((tid & 255) + 256 * k) * 16;        // byte offset inside a plane image
  const bool x1 = (tid & 255) + 256 < 408;        // the second piece exists for 152 threads of a chunk's 256

  // ---- fetch cursor (three steps ahead of the MFMAs) and its per-segment address tables
  struct Cur { int col, p, last, da, db; long left; } F;
  int xpo[2], ypo;
  unsigned xvm = 0;                                // bit k: piece k is a voxel inside the tensor
  __amdgpu_buffer_rsrc_t xrs, yrs;
  int nbF = 0;
  auto seg_setup = [&]() {                         // F.col changed
    int r_ = F.col;
    const int tw = r_ % P.tilesW; r_ /= P.tilesW;
    const int th = r_ % P.tilesH;
    nbF = r_ / P.tilesH;
    const int oh0 = th * 4, ow0 = tw * 32, uh0 = oh0 - 1, uw0 = ow0 - 1;
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nbF * xsample), 0, (int)xsample, 0x00020000);
    yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Y.ptr + (size_t)nbF * ysample), 0, (int)ysample, 0x00020000);
    xvm = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = (tid & 255) + 256 * k, vx = i >> 1, row = vx / 34, cx = vx - row * 34;
      const int uh = uh0 + row, uw = uw0 + cx;
      const bool ok = (i < 408) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi) && (8 * hx < ck);
      xpo[k] = ok ? ((uh * c.Wi + uw) * S.cs + cc.c0 + 8 * hx) * 2 : (int)0x80000000;
      xvm |= (ok ? 1u : 0u) << k;
    }
    {
      const int i = tid, row = (i >> 6) & 3, vox = (i >> 1) & 31, q = 2 * (i >> 8) + (i & 1);
      const int oh = oh0 + row, ow = ow0 + vox;
      const bool ok = (oh < c.Ho) && (ow < c.Wo) && (cot * 32 + 8 * q < c.Cout);
      ypo = ok ? ((oh * c.Wo + ow) * Y.cs + cot * 32 + 8 * q) * 2 : (int)0x80000000;
    }
  };
  auto seg_begin = [&](int col, int da, long left) {      // a new segment: planes [da, db) of column col
    F.col = col; F.da = da;
    const long room = c.Do - da;
    F.db = da + (int)(left < room ? left : room);
    F.p = da - LEAD; F.last = F.db - 1 + LEAD;
    F.left = left - (F.db - da);
    seg_setup();
  };
  bool Fon = t0 < t1;
  F.col = 0; F.p = 0; F.last = 0; F.da = 0; F.db = 0; F.left = 0;
  xpo[0] = xpo[1] = ypo = (int)0x80000000;
  xrs = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, 0, 0x00020000);
  yrs = __builtin_amdgcn_make_buffer_rsrc((void*)Y.ptr, 0, 0, 0x00020000);
  if (Fon) seg_begin((int)(t0 / c.Do), (int)(t0 % c.Do), t1 - t0);
  int nsteps = 0;                                 // total steps of this workgroup: every segment costs its planes + 2 LEAD
  {
    long t = t0;
    while (t < t1) { const long room = c.Do - t % c.Do; const long n = (t1 - t < room) ? t1 - t : room; nsteps += (int)n + 2 * LEAD; t += n; }
  }

  // raw register sets: data of step t lives in set t % 3
  uint4 rx[3][2], ry[3];
  unsigned rvm[3] = {0, 0, 0};                     // validity bits of the set's X pieces (0 for a plane outside the tensor)
  int rnb[3] = {0, 0, 0};                          // sample of the set (scale / shift of the activation)
  int rfirst[3] = {0, 0, 0};                       // the set opens a segment: the register window restarts from zero
  int first_now = 0;                               // ... for the step whose images are in LDS (copied when its set is activated)

  auto fetch = [&](uint4 (&gx)[2], uint4& gy, unsigned& gvm, int& gnb, int& gfirst) {     // issue only; advances the cursor
    // (the three loads are issued on EVERY call — past the end of the work list with out-of-range offsets, which return zeros without touching
    // memory: a step that issued fewer vector loads on some path would make the compiler's vmcnt bookkeeping fall back to vmcnt(0), i.e. drain
    // the three-step prefetch at every step)
    const int p = F.p, q = p + LEAD;
    const bool pv = Fon && (unsigned)p < (unsigned)c.Di, qv = Fon && q >= F.da && q < F.db;
    const int poff = __builtin_amdgcn_readfirstlane(pv ? p * xplane_bytes : 0), xoob = __builtin_amdgcn_readfirstlane(pv ? 0 : (int)0x80000000);
    const int qoff = __builtin_amdgcn_readfirstlane(qv ? q * yplane_bytes : 0), yoob = __builtin_amdgcn_readfirstlane(qv ? 0 : (int)0x80000000);
    if (!(BWT_ABL & 6)) {
#pragma unroll
      for (int k = 0; k < 2; ++k) gx[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xpo[k] | xoob, poff, 0));
      gy = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ypo | yoob, qoff, 0));
    }
    gvm = pv ? xvm : 0u;
    gnb = nbF;
    gfirst = (Fon && p == F.da - LEAD) ? 1 : 0;
    if (Fon) {
      if (p < F.last) F.p = p + 1;
      else if (F.left > 0) seg_begin(F.col + 1, 0, F.left);
      else Fon = false;
    }
  };

  // activation constants: an LDS table of every sample's scale / shift for the pair's two chunks (0 / 0 for channels past the chunk: they
  // activate to exactly 0), filled once; a step reads the 8 + 8 values of its thread (chunk cih, channel half hx) for the sample of the set it
  // activates — LDS reads, so that no step waits on the vector-memory counter for anything but its own raw set
  float* const TAB = (float*)(L + BWT_IMG_BYTES);
  for (int i = tid; i < c.N * 64; i += BWT_THREADS) {
    const int ch = i & 15, sel = (i >> 4) & 1, cq = (i >> 5) & 1, nb = i >> 6;
    const int chq = 2 * pz + cq;
    float v = sel ? 0.f : 1.f;
    if (chq < P.nchunks) {
      const ConvChunk cq_ = P.chunk[chq];
      const mt_src_t& Sq = c.src[cq_.src];
      if (ch >= cq_.ck) v = 0.f;
      else if (Sq.scale != nullptr) v = sel ? Sq.shift[(size_t)nb * Sq.C + cq_.c0 + ch] : Sq.scale[(size_t)nb * Sq.C + cq_.c0 + ch];
    } else v = 0.f;
    TAB[i] = v;
  }
  float sc[8], sh[8];
  auto load_affine = [&](int nb) {
    const float4* t = (const float4*)(TAB + (nb * 2 + cih) * 32 + 8 * hx);
    const float4 a0 = t[0], a1 = t[1], b0 = t[4], b1 = t[5];
    sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
    sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
  };
  auto put_piece = [&](const uint4& raw, unsigned ok, char* dst) {       // activate 8 channels, round once to bf16, 16-byte store (prologue)
    const unsigned m = ok ? 0xffffffffu : 0u;        // zero padding applies AFTER the activation
    const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
    uint4 o;
    unsigned od[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (BWT_ABL & 1) od[d] = rw[d] & m;
      else {
        const float t0_ = fmaf(mt_lo16<XS>(rw[d]), sc[2 * d], sh[2 * d]), t1_ = fmaf(mt_hi16<XS>(rw[d]), sc[2 * d + 1], sh[2 * d + 1]);
        od[d] = mt_pk_bf16(fmaxf(t0_, t0_ * xslope), fmaxf(t1_, t1_ * xslope)) & m;       // LeakyReLU as max(t, slope t): 0 <= slope <= 1
      }
    }
    o.x = od[0]; o.y = od[1]; o.z = od[2]; o.w = od[3];
    *(uint4*)dst = o;
  };

  const int xlane = (4 * lk + (li >> 2)) * 32 + (li & 3) * 8;
  const char* const xbase = L + xlane + cih * BWT_XCH + rp * 2 * BWT_XROW;
  const char* const ybase = L + BWT_YI_OFF + xlane + coh * BWT_YHALF + rp * 2 * BWT_YROW;

  // ---- one step.  PH = step % 3: LDS images PH hold X plane p / dY plane p + LEAD; the barrier that publishes them sits near the END of the
  // previous step, so that the first two X operands and the dY operands of this step (window slot PH) are already requested under the
  // previous step's last MFMAs.  The raw set of step s + 1 (set NX = (PH + 1) % 3, loads issued three steps ago) is activated into images
  // NX one small task per MFMA slot (pinned by sched_barrier: left alone the compiler puts the whole activation in front of the MFMAs),
  // then refilled with the loads of step s + 4.
  bwb_bf16x8 xop[4];                               // X operands: group g in xop[g % 4], requested two groups ahead (its registers were last read by group g - 4)
  float a_t0 = 0.f, a_t1 = 0.f, a_u0 = 0.f;        // activation task state
  unsigned a_od[4] = {0, 0, 0, 0}, a_m = 0;
  auto step = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    constexpr int NX = (PH + 1) % 3;
    constexpr int NSLOT_A = KD * 15;               // MFMA slots of the groups xr = 0..2
    if (KD == 3 && first_now) {                    // a new column: the two older planes of the window are not this column's
#pragma unroll
      for (int s = 1; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) win[(PH + s) % 3][r] = __builtin_bit_cast(bwb_bf16x8, uint4{0, 0, 0, 0});
    }
    load_affine(rnb[NX]);
    first_now = rfirst[NX];                        // (set NX is refilled by the fetch task below)
    char* const xdst = L + NX * BWT_XBUF;
    char* const ydst = L + BWT_YI_OFF + NX * BWT_YSLOT + tid * 16;
    // tasks: q = 0..23 the activation of the set's two X pieces in thirds of a dword (the dY store rides on the first), 24 the fetch
    auto task = [&](int q) {                       // q is a compile-time constant after unrolling
      if (q == 24) { fetch(rx[NX], ry[NX], rvm[NX], rnb[NX], rfirst[NX]); return; }
      if (BWT_ABL & 2) return;
      const int d = q / 3, k = d >> 2, dw = d & 3, part = q % 3;
      const unsigned raw = dw == 0 ? rx[NX][k].x : dw == 1 ? rx[NX][k].y : dw == 2 ? rx[NX][k].z : rx[NX][k].w;
      if (part == 0) {
        if (q == 0) *(uint4*)ydst = ry[NX];
        if (dw == 0) a_m = (unsigned)(-(int)((rvm[NX] >> k) & 1u));            // zero padding applies AFTER the activation
        a_t0 = fmaf(mt_lo16<XS>(raw), sc[2 * dw], sh[2 * dw]); a_t1 = fmaf(mt_hi16<XS>(raw), sc[2 * dw + 1], sh[2 * dw + 1]);
        a_u0 = a_t0 * xslope;
      } else if (part == 1) {
        const float u1 = a_t1 * xslope;
        a_t0 = fmaxf(a_t0, a_u0); a_t1 = fmaxf(a_t1, u1);                      // LeakyReLU as max(t, slope t): 0 <= slope <= 1
      } else {
        a_od[dw] = ((BWT_ABL & 1) ? raw : mt_pk_bf16(a_t0, a_t1)) & a_m;
        if (dw == 3) {
          uint4 o; o.x = a_od[0]; o.y = a_od[1]; o.z = a_od[2]; o.w = a_od[3];
          if (k == 0 || x1) *(uint4*)(xdst + xlo[k]) = o;
        }
      }
    };
    // slot = position of an MFMA in the step's sequence (a constant after unrolling: pure arithmetic on the loop indices).  KD = 3: one task
    // behind each of the first 24 MFMAs, the fetch behind MFMA 30; KD = 1 (15 + 3 slots, staging-bound anyway): two tasks per slot.
    // (Measured and not kept: the two waves of a SIMD taking their tasks in ALTERNATE slots, task q behind MFMA 2q + cih, so that one wave's
    // vector work meets the other's bare MFMA — 254 against 212 us for 30->30 @ 2x48x192x192: the uniform branch per slot costs more.)
    auto mfma_slot = [&](int slot, int g, int kw, int kh, int r, int kd) {
      const int ws = KD == 3 ? (PH + 3 - kd) % 3 : PH;                         // plane p + 1 - kd was brought by step s - kd
      if (!(BWT_ABL & 8))
        acc[(kd * 3 + kh) * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xop[g % 4], win[ws][r], acc[(kd * 3 + kh) * 3 + kw], 0, 0, 0);
      if (KD == 3 && !BWT_EARLY) {
        if (slot < 24) task(slot);
        else if (slot == 30) task(24);
      } else {
        if (slot < 12) { task(2 * slot); task(2 * slot + 1); }
        else if (slot == 12) task(24);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // groups g = xr * 3 + kw; halo row xr of the pair serves output row r with kh = xr - r
#pragma unroll
    for (int xr = 0; xr < 3; ++xr)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int g = xr * 3 + kw, gn = g + 2;
        xop[gn % 4] = bwt_operand(xbase, PH * BWT_XBUF + (gn / 3) * BWT_XROW + (gn % 3) * 32);
        __builtin_amdgcn_sched_barrier(0);
        const int nr = xr == 0 ? 1 : 2;                                         // output rows this halo row serves
        const int s0 = KD * (3 * (xr == 0 ? 0 : xr == 1 ? 1 : 3) + kw * nr);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int kh = xr - r;
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int kd = 0; kd < KD; ++kd) mfma_slot(s0 + (xr == 0 ? 0 : r) * KD + kd, g, kw, kh, r, kd);
        }
      }
    // the last halo row: groups 9, 10 (requested above), 11
    xop[3] = bwt_operand(xbase, PH * BWT_XBUF + 3 * BWT_XROW + 2 * 32);        // group 11
    __builtin_amdgcn_sched_barrier(0);
    if (KD == 3) {                                 // oldest plane first: its window slot is the one the next step's dY operands go to
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) mfma_slot(NSLOT_A + kw, 9 + kw, kw, 2, 1, 2);
    }
    __syncthreads();                               // images NX complete (and every wave is past its reads of images PH except xop[1..3], in registers)
    // requests of the NEXT step under the remaining MFMAs of this one: its dY operands into window slot NX, its first two X operands
#pragma unroll
    for (int r = 0; r < 2; ++r) win[NX][r] = bwt_operand(ybase, NX * BWT_YSLOT + r * BWT_YROW);
    xop[0] = bwt_operand(xbase, NX * BWT_XBUF);                                // group 0 of the next step (xop[0] was group 8: done)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int kd = 0; kd < (KD == 3 ? 2 : 1); ++kd) mfma_slot(NSLOT_A + 3 + kw * 2 + kd, 9 + kw, kw, 2, 1, kd);
      if (kw == 0) {
        xop[1] = bwt_operand(xbase, NX * BWT_XBUF + 32);                       // group 1 of the next step (xop[1] was group 9: done)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: sets 0, 1, 2 in flight, set 0 activated into images 0, set 0 refilled with step 3
  if (nsteps > 0) {
    fetch(rx[0], ry[0], rvm[0], rnb[0], rfirst[0]);
    fetch(rx[1], ry[1], rvm[1], rnb[1], rfirst[1]);
    fetch(rx[2], ry[2], rvm[2], rnb[2], rfirst[2]);
    __syncthreads();                               // the activation table
    load_affine(rnb[0]);
    put_piece(rx[0][0], rvm[0] & 1u, L + xlo[0]);
    if (x1) put_piece(rx[0][1], (rvm[0] >> 1) & 1u, L + xlo[1]);
    *(uint4*)(L + BWT_YI_OFF + tid * 16) = ry[0];
    first_now = rfirst[0];
    fetch(rx[0], ry[0], rvm[0], rnb[0], rfirst[0]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) win[0][r] = bwt_operand(ybase, r * BWT_YROW);
    xop[0] = bwt_operand(xbase, 0);
    xop[1] = bwt_operand(xbase, 32);
    // (steps run in whole triples — up to two trailing steps multiply zeros: past the end of the work list the fetch returns zeros — so that
    // the loop body is straight-line code with the same number of vector loads on every path: a path that skipped a step and returned to the
    // loop head would make the compiler's vmcnt bookkeeping wait for younger raw sets than the one a step consumes)
    for (int s = 0; s < nsteps; s += 3) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
    }
  }

  // ---- the two row pairs of a (chunk, cout half) are summed through LDS, nine taps at a time; one partial per (chunk, cout tile, workgroup)
  float* const red = (float*)ldsw;                 // [cih * 2 + coh][tap 9][j][lane]
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) {
    __syncthreads();
    if (rp == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(((cih * 2 + coh) * 9 + t) * 4 + j) * 64 + lane] = acc[kd * 9 + t][j];
    }
    __syncthreads();
    if (rp == 0 && chv) {
      float* pp = P.part + ((size_t)((size_t)(chi * P.ncot + cot) * gridDim.x + blockIdx.x) * NT) * 512;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pp[(size_t)(kd * 9 + t) * 512 + (lk * 4 + j) * 32 + coh * 16 + li] = acc[kd * 9 + t][j] + red[(((cih * 2 + coh) * 9 + t) * 4 + j) * 64 + lane];
    }
  }
}

int mt_launch_bwdw_tr16(const BwdWParams& P, int KD, int xdt, hipStream_t st) {
  MT_REQUIRE((KD == 1 || KD == 3) && (xdt == MT_F16 || xdt == MT_BF16), "bwd_weight (tr16): KD %d / X storage type %d", KD, xdt);
  MT_REQUIRE(P.c.N <= BWT_MAXN, "bwd_weight (tr16): %d samples (the activation table holds %d)", P.c.N, BWT_MAXN);
  void (*kfn)(const BwdWParams) = KD == 3 ? (xdt == MT_F16 ? conv_bwdw_tr16_kernel<3, MT_F16> : conv_bwdw_tr16_kernel<3, MT_BF16>)
                                          : (xdt == MT_F16 ? conv_bwdw_tr16_kernel<1, MT_F16> : conv_bwdw_tr16_kernel<1, MT_BF16>);
  static std::atomic<uint64_t> attr_s[4];
  const int devid = mt_current_device();
  std::atomic<uint64_t>& at = attr_s[(KD == 3 ? 0 : 2) + (xdt == MT_F16 ? 0 : 1)];
  if (mt_device_pending(at, devid)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWT_LDS_BYTES);
    if (e != hipSuccess) { mt_set_error("bwd_weight: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
    mt_mark_device_done(at, devid);
  }
  hipLaunchKernelGGL(kfn, dim3(P.nsg, P.ncot, (P.nchunks + 1) / 2), dim3(BWT_THREADS), BWT_LDS_BYTES, st, P);
  MT_CHECK_LAUNCH("conv_bwdw_tr16");
  return MT_OK;
}
