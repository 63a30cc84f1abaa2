// conv_x16s.hip — the 16-bit stride-1 3x3x3 / 1x3x3 convolution as ONE software pipeline per wave (round 6, second form).
//
// conv_x16_kernel (conv_x16.hip) solved the operand path — its MFMA phase alone runs at 0.91 - 0.97 of the 16-bit matrix peak — but its
// phases (patch loads, conversion, MFMAs, epilogue) are separated by barriers and two workgroups per CU overlap almost nothing: the
// MFMA phase is 23 - 30 % of a wave's time (profiles/r06_x16_phase_cycles_ablations_sq.txt).  Here every phase lives INSIDE the wave's
// own MFMA stream: ONE workgroup per CU, four waves with the whole register file (512 per lane), patch image AND weight image double
// buffered in LDS (2 x 39 + 2 x 27 KiB).  While step s multiplies out of buffers s & 1, the same wave
//   * converts the raw pieces of chunk s + 1 (lazy InstanceNorm + LeakyReLU, one rounding) into patch buffer (s + 1) & 1,
//   * moves the weight fragments of chunk s + 1 (requested at the start of the step) into weight buffer (s + 1) & 1,
//   * requests the raw patch of the chunk PAIR two pairs ahead (both chunks of a 32-channel pair in one go: the L2 -> L1 line rate),
// slotted between the MFMAs of the nine (kd, kw) groups (a 32-cycle v_mfma_f32_32x32x16 hides <= 5 other instructions), and ONE
// barrier ends the step.  Same tile (4 x 4 x 32 outputs x 32 channels, wave = output plane), same LDS images, same epilogue
// (conv_x16_epi.inc) as conv_x16_kernel.  Reference: the layers are generic_UNet.py:28-70 / conv_blocks.py:116-213 under autocast
// (nnUNetTrainerV2.py:236-249).
#include "bwdw_common.h"
#include "conv16_common.h"

#ifndef XS_ABL
#define XS_ABL 0     // timing ablations: 1 no global loads, 2 no conversion / LDS writes, 4 no epilogue, 8 no MFMAs
#endif
#define X16_ABL ((XS_ABL & 4) ? 0 : 0)
#include "conv_x16_epi.inc"

struct XsPair { X16Geo g; int ch0, ch1, last, valid; };

template <int ST> __device__ __forceinline__ unsigned xs_act2(unsigned d, float sc0, float sh0, float sc1, float sh1, float slope) {
  const float t0 = __builtin_fmaf(mt_lo16<ST>(d), sc0, sh0), t1 = __builtin_fmaf(mt_hi16<ST>(d), sc1, sh1);
  return mt_pk16<ST>(fmaxf(t0, t0 * slope), fmaxf(t1, t1 * slope));
}

// AFF: the sources may carry a lazy activation (forward; a plain source of such a launch gets scale 1, shift 0, slope 1: the identity,
// exact).  false: every source is plain (backward-data over gradients): the conversion is a masked copy.
template <int KD, int ST, bool ACC, bool AFF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_x16s_kernel(const X16Params P) {
  constexpr int TD = 4, TH = 4, TW = 32;
  constexpr int LD = TD + KD - 1, LH = TH + 2, LW = TW + 2;
  constexpr int ROWS = LD * LH, RPW = ROWS / 4;               // 36 / 9 (3x3x3), 24 / 6 (1x3x3)
  constexpr int NTAP = KD * 9, PD = (KD - 1) / 2, NG = KD * 3;
  constexpr int A_BYTES = ROWS * LW * 32, B_BYTES = NTAP * 1024, NBP = (B_BYTES / 16 + 255) / 256;
  constexpr int RPG = (RPW + NG - 1) / NG;                    // rows converted per MFMA group: 1 (3x3x3), 2 (1x3x3)
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_lds[];
  unsigned char* const ldsA0 = xs_lds;                        // patch images [2]
  unsigned char* const ldsB0 = xs_lds + 2 * A_BYTES;          // weight images [2]
  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const stg = ldsB0 + 2 * B_BYTES + wave * 2048;          // wave-private epilogue image
  const int DUMMY = 2 * A_BYTES + 2 * B_BYTES + 4 * 2048;               // byte offset of 4 KiB nobody reads: stores of lanes without a piece
  float* const tab = (float*)(xs_lds + DUMMY + 4096);                    // [chunk][scale | shift][16] of one sample
  float* const tabb = tab + P.nchunks * 32;                              // bias of every output channel
  const int li = lane & 31, lhalf = lane >> 5;

  const int G = (int)gridDim.x;
  const int lb = mt_xcd_remap((int)blockIdx.x, G);
  const int it0 = (int)((long)lb * P.nitems / G), it1 = (int)((long)(lb + 1) * P.nitems / G);
  if (it0 >= it1) return;
  auto decode = [&](int item, X16Geo& g) __attribute__((always_inline)) {
    int tile = item / P.ncot;
    g.ntile = item - tile * P.ncot;
    const int td = tile % P.tilesD; tile /= P.tilesD;
    const int th = tile % P.tilesH; tile /= P.tilesH;
    const int tw = tile % P.tilesW;
    g.nb = tile / P.tilesW;
    g.od0 = td * TD; g.oh0 = th * TH; g.ow0 = tw * TW;
    g.sb = (td * P.tilesH + th) * P.tilesW + tw;
  };
  // the stream of chunk pairs this workgroup walks: (item, pair of the item) in order
  int item_c = it0, pr_c = 0;
  auto next_pair = [&](XsPair& d) __attribute__((always_inline)) {
    if (item_c >= it1) { d.valid = 0; d.last = 0; d.ch0 = 0; d.ch1 = -1; return; }
    decode(item_c, d.g);
    d.ch0 = P.pair[pr_c][0]; d.ch1 = P.pair[pr_c][1];
    d.last = pr_c + 1 == P.npairs;
    d.valid = 1;
    if (++pr_c == P.npairs) { pr_c = 0; ++item_c; }
  };

  // ---- per-lane constants of the staging pass (as conv_x16_kernel): main piece = (column 1 + lane / 2, channel half lane & 1) of the
  // wave's rows w, w + 4, ...; halo piece = columns 0 and LW - 1, one piece for each of the first 4 * ROWS threads (the others: DUMMY)
  const int hf = lane & 1;
  const int lwm = 1 + (lane >> 1);
  const int ldsA_main = lwm * 32 + (((hf ^ (lwm >> 3)) & 1) * 16);
  const bool has_halo = tid < ROWS * 4;
  const int hrow = has_halo ? (tid >> 2) : 0;
  const int lwh = ((tid >> 1) & 1) ? (LW - 1) : 0;
  const int hld = hrow / LH, hlh = hrow - hld * LH;
  const int ldsA_halo = (hrow * LW + lwh) * 32 + (((hf ^ (lwh >> 3)) & 1) * 16);
  const int abase_w = wave * LH * LW * 32;
  int abase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = li + kw;
    abase[kw] = abase_w + col * 32 + (((lhalf ^ (col >> 3)) & 1) * 16);
  }

  // ---- raw pieces of two chunk pairs: R[parity of the pair][chunk of the pair][row 0 .. RPW-1 | halo]
  x16_u32x4 R[2][2][RPW + 1];
  XsPair pd[2];
  x16_u32x4 rbw[NBP];                       // the next step's weight fragments on their way to LDS

  auto issue_rows = [&](const XsPair& d, x16_u32x4 (&r)[2][RPW + 1], int q0, int q1, bool with_halo) __attribute__((always_inline)) {
    const ConvChunk cc = P.chunk[d.ch0];
    const mt_src_t& S = c.src[cc.src];
    const int cs = S.cs;
    const size_t sample_bytes = (size_t)c.Di * c.Hi * c.Wi * cs * 2;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)d.g.nb * sample_bytes), 0, (int)sample_bytes, 0x00020000);
    const int ud0 = d.g.od0 - PD, uh0 = d.g.oh0 - 1, uw0 = d.g.ow0 - 1;
    const int rowbytes = c.Wi * cs * 2;
    const int uwm = uw0 + lwm;
    const bool live = d.valid != 0 && !(XS_ABL & 1);
    const int voffm = (live && (unsigned)uwm < (unsigned)c.Wi) ? (uwm * cs + cc.c0 + 8 * hf) * 2 : (int)0x80000000;
    const int second = (d.ch1 >= 0) ? 0 : (int)0x80000000;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      if (q < q0 || q >= q1) continue;
      const int row = wave + 4 * q;
      const int ld = row / LH, lh = row - ld * LH;
      const int ud = ud0 + ld, uh = uh0 + lh;
      const int rm = (((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi)) ? -1 : 0;
      const int soff = ((ud * c.Hi + uh) & rm) * rowbytes;
      const int vo = voffm | (~rm & (int)0x80000000);
      r[0][q] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
      r[1][q] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (vo + 32) | second, soff, 0));
    }
    if (with_halo) {
      const int ud = ud0 + hld, uh = uh0 + hlh, uw = uw0 + lwh;
      const bool ok = live && has_halo && ((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi);
      const int voff = ok ? (((ud * c.Hi + uh) * c.Wi + uw) * cs + cc.c0 + 8 * hf) * 2 : (int)0x80000000;
      r[0][RPW] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
      r[1][RPW] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (voff + 32) | second, 0, 0));
    }
  };
  auto issue_weights = [&](int ntile, int ch, bool live) __attribute__((always_inline)) {
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)c.wpack + (size_t)(ntile * P.nchunks + ch) * B_BYTES), 0, B_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < NBP; ++i)
      rbw[i] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (live && !(XS_ABL & 1)) ? (tid + 256 * i) * 16 : (int)0x80000000, 0, 0));
  };
  auto store_weights = [&](unsigned char* ldsB, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      if (i < i0 || i >= i1) continue;
      const int off = (tid + 256 * i) * 16;
      unsigned char* dst = ((i + 1) * 256 * 16 <= B_BYTES || off < B_BYTES) ? ldsB + off : xs_lds + DUMMY + tid * 16;
      if (!(XS_ABL & 2)) *(x16_u32x4*)dst = rbw[i];
    }
  };

  // conversion constants of a chunk (scale / shift of the lane's eight channels from the table, channel-tail masks) and of a tile (row /
  // column validity as masks) — plain data, no control flow, so that the conversion can sit between MFMAs
  struct CvtC { float sc[8], sh[8]; unsigned cm[4]; float slope; int ud0, uh0; unsigned colm, halom; };
  auto cvt_setup = [&](const XsPair& d, int ch, CvtC& k) __attribute__((always_inline)) {
    const ConvChunk cc = P.chunk[ch];
    const mt_src_t& S = c.src[cc.src];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ce = 8 * hf + 2 * e;
      k.cm[e] = ((ce < cc.ck) ? 0x0000ffffu : 0u) | ((ce + 1 < cc.ck) ? 0xffff0000u : 0u);
    }
    if constexpr (AFF) {
      const f32x4* tp = (const f32x4*)(tab + ch * 32 + 8 * hf);
      const f32x4 s0 = tp[0], s1 = tp[1], h0 = tp[4], h1 = tp[5];
#pragma unroll
      for (int e = 0; e < 4; ++e) { k.sc[e] = s0[e]; k.sc[4 + e] = s1[e]; k.sh[e] = h0[e]; k.sh[4 + e] = h1[e]; }
      k.slope = S.scale != nullptr ? S.slope : 1.f;
    }
    k.ud0 = d.g.od0 - PD; k.uh0 = d.g.oh0 - 1;
    const int uw0 = d.g.ow0 - 1;
    k.colm = ((unsigned)(uw0 + lwm) < (unsigned)c.Wi) ? 0xffffffffu : 0u;
    const int ud = k.ud0 + hld, uh = k.uh0 + hlh, uw = uw0 + lwh;
    k.halom = (((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi)) ? 0xffffffffu : 0u;
  };
  auto cvt_put = [&](const CvtC& k, x16_u32x4 v, unsigned vm, unsigned char* dst) __attribute__((always_inline)) {
    if constexpr (AFF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xs_act2<ST>(v[e], k.sc[2 * e], k.sh[2 * e], k.sc[2 * e + 1], k.sh[2 * e + 1], k.slope);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] &= k.cm[e] & vm;
    if (!(XS_ABL & 2)) *(x16_u32x4*)dst = v;
  };
  auto cvt_row = [&](const CvtC& k, const x16_u32x4 (&xa)[RPW + 1], int q, unsigned char* ldsA) __attribute__((always_inline)) {
    const int row = wave + 4 * q;
    const int ld = row / LH, lh = row - ld * LH;
    const int ud = k.ud0 + ld, uh = k.uh0 + lh;
    const unsigned rm = (((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi)) ? 0xffffffffu : 0u;
    cvt_put(k, xa[q], rm & k.colm, ldsA + row * (LW * 32) + ldsA_main);
  };
  auto cvt_halo = [&](const CvtC& k, const x16_u32x4 (&xa)[RPW + 1], unsigned char* ldsA) __attribute__((always_inline)) {
    cvt_put(k, xa[RPW], k.halom, has_halo ? ldsA + ldsA_halo : xs_lds + DUMMY + tid * 16);
  };

  f32x16 acc[1][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0][m][j] = 0.f;
  };
  zero_acc();

  int tab_nb = -1;
  auto table = [&](int nb) __attribute__((always_inline)) {
    if (nb == tab_nb) return;
    __syncthreads();                       // (nobody converts with the old sample's constants any more)
    for (int e = tid; e < P.nchunks * 32; e += 256) {
      const int ch = e >> 5, k = e & 15;
      const ConvChunk cc = P.chunk[ch];
      const mt_src_t& S = c.src[cc.src];
      float v = 0.f;
      if (k < cc.ck) {
        if (S.scale != nullptr) v = ((e >> 4) & 1) ? S.shift[(size_t)nb * S.C + cc.c0 + k] : S.scale[(size_t)nb * S.C + cc.c0 + k];
        else v = ((e >> 4) & 1) ? 0.f : 1.f;
      }
      tab[e] = v;
    }
    tab_nb = nb;
    __syncthreads();
  };

  // ---- one step: the MFMAs of chunk `s` out of buffers lbuf, and between them everything step s + 1 needs.
  //   LASTP: this is the last chunk of its pair -> the next chunk is the first of the OTHER register set's pair, and this set is free:
  //   the pair two ahead is requested into it.
  int lbuf = 0;
  auto step = [&](auto PAR_, auto LASTP_) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PAR_)::value;
    constexpr bool LASTP = decltype(LASTP_)::value;
    const XsPair nxt = LASTP ? pd[PAR ^ 1] : pd[PAR];
    const int nch = LASTP ? nxt.ch0 : nxt.ch1;
    if constexpr (AFF) table(nxt.g.nb);
    const unsigned char* const rdA = ldsA0 + lbuf * A_BYTES;
    const unsigned char* const rdB = ldsB0 + lbuf * B_BYTES;
    unsigned char* const wrA = ldsA0 + (lbuf ^ 1) * A_BYTES;
    unsigned char* const wrB = ldsB0 + (lbuf ^ 1) * B_BYTES;
    if constexpr (LASTP) next_pair(pd[PAR]);           // (the descriptor of the pair that now goes into this register set)
    CvtC k;
    cvt_setup(nxt, nch, k);
    const x16_u32x4 (&src)[RPW + 1] = LASTP ? R[PAR ^ 1][0] : R[PAR][1];
    bf16x8 a[6], b[2][3];
    auto lda = [&](int g, int r) __attribute__((always_inline)) { return *(const bf16x8*)(rdA + abase[g % 3] + ((g / 3) * LH + r) * (LW * 32)); };
    auto ldb = [&](int g, int kh) __attribute__((always_inline)) { return *(const bf16x8*)(rdB + ((g / 3) * 9 + kh * 3 + (g % 3)) * 1024 + lane * 16); };
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = lda(0, r);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) b[0][kh] = ldb(0, kh);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // ---- fillers of this group
      if (g == 0) issue_weights(nxt.g.ntile, nch, nxt.valid != 0);
      if constexpr (LASTP) {
        if (g == 0) issue_rows(pd[PAR], R[PAR], 0, (RPW + 1) / 2, false);
        if (g == 1) issue_rows(pd[PAR], R[PAR], (RPW + 1) / 2, RPW, true);
      }
#pragma unroll
      for (int q = g * RPG; q < (g + 1) * RPG && q < RPW; ++q) cvt_row(k, src, q, wrA);
      if (g == NG - 1) cvt_halo(k, src, wrA);
      if (g >= NG - 3) {
        constexpr int per = (NBP + 2) / 3;
        store_weights(wrB, (g - (NG - 3)) * per, (g - (NG - 3) + 1) * per);
      }
      // ---- the group's 12 MFMAs; an input-row register is refilled right behind its last use
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if (r - kh >= 0 && r - kh < 4 && !(XS_ABL & 8)) acc[0][r - kh] = mt_mfma16<ST>(a[r], b[g & 1][kh], acc[0][r - kh]);
        if (g + 1 < NG) {
          a[r] = lda(g + 1, r);
          if (r < 3) b[(g + 1) & 1][r] = ldb(g + 1, r);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    lbuf ^= 1;
  };

  auto run_pair = [&](auto PAR_) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PAR_)::value;
    const XsPair cur = pd[PAR];
    if (cur.ch1 >= 0) {
      step(PAR_, std::false_type());
      step(PAR_, std::true_type());
    } else {
      step(PAR_, std::true_type());
    }
    if (cur.last) {
      if (!(XS_ABL & 4)) x16_epilogue<ST, ACC>(c, acc, stg, tabb, cur.g, wave, lane);
      else if (acc[0][0][0] == 12345.678f) ((float*)c.out0)[0] = acc[0][3][3];
      zero_acc();
      if (c.stats_part != nullptr && !(XS_ABL & 4)) {
        __syncthreads();
        if (tid < 32) {
          const int co = cur.g.ntile * 32 + tid;
          if (co < c.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float* sp = (const float*)(ldsB0 + 2 * B_BYTES + w * 2048);
              t1 += sp[tid * 2]; t2 += sp[tid * 2 + 1];
            }
            float* dst = c.stats_part + ((size_t)((size_t)cur.g.nb * P.nsb + cur.g.sb) * c.Cout + co) * 2;
            dst[0] = t1; dst[1] = t2;
          }
        }
      }
    }
  };

  // ---- prologue: the first two pairs requested, the first chunk converted, its weights in place
  for (int e = tid; e < P.ncot * 32; e += 256) tabb[e] = (c.bias != nullptr && e < c.Cout) ? c.bias[e] : 0.f;
  pd[0].g = X16Geo{0, 0, 0, 0, 0, 0}; pd[1].g = pd[0].g;
  next_pair(pd[0]);
  issue_rows(pd[0], R[0], 0, RPW, true);
  next_pair(pd[1]);
  issue_rows(pd[1], R[1], 0, RPW, true);
  issue_weights(pd[0].g.ntile, pd[0].ch0, true);
  if constexpr (AFF) table(pd[0].g.nb); else __syncthreads();
  {
    CvtC k;
    cvt_setup(pd[0], pd[0].ch0, k);
#pragma unroll
    for (int q = 0; q < RPW; ++q) cvt_row(k, R[0][0], q, ldsA0);
    cvt_halo(k, R[0][0], ldsA0);
    store_weights(ldsB0, 0, NBP);
  }
  __syncthreads();
  while (true) {
    if (!pd[0].valid) break;
    run_pair(std::integral_constant<int, 0>());
    if (!pd[1].valid) break;
    run_pair(std::integral_constant<int, 1>());
  }
}

static size_t xs_lds_bytes(int KD, int nchunks, int ncot) {
  const int rows = (4 + KD - 1) * 6;
  return 2 * ((size_t)rows * 34 * 32 + (size_t)KD * 9 * 1024) + 4 * 2048 + 4096 + (size_t)nchunks * 128 + (size_t)ncot * 128;
}

bool mt_conv_x16s_fits(int KD, int nchunks, int ncot) { return xs_lds_bytes(KD, nchunks, ncot) <= 160 * 1024; }

int mt_conv_x16s_workgroups(int nitems) {
  const int cap = mt_device_cus(mt_current_device());
  return nitems < cap ? nitems : cap;
}

int mt_launch_conv_x16s(const X16Params& P, int KD, int dt, bool aff, hipStream_t st) {
  const size_t ldsb = xs_lds_bytes(KD, P.nchunks, P.ncot);
  MT_REQUIRE(ldsb <= 160 * 1024, "conv3d (x16s): %zu bytes of LDS", ldsb);
  MT_REQUIRE((KD == 1 || KD == 3) && (dt == MT_F16 || dt == MT_BF16), "conv3d (x16s): KD %d, type %d", KD, dt);
  const bool acc = P.c.accumulate != 0;
  void (*kfn)(const X16Params) = nullptr;
#define XS_PICK(KD_, ST_) \
  if (KD == KD_ && dt == ST_) kfn = acc ? (aff ? conv_x16s_kernel<KD_, ST_, true, true> : conv_x16s_kernel<KD_, ST_, true, false>) \
                                        : (aff ? conv_x16s_kernel<KD_, ST_, false, true> : conv_x16s_kernel<KD_, ST_, false, false>);
  XS_PICK(3, MT_F16) XS_PICK(3, MT_BF16) XS_PICK(1, MT_F16) XS_PICK(1, MT_BF16)
#undef XS_PICK
  static std::atomic<uint64_t> done[16];
  std::atomic<uint64_t>& d = done[(KD == 3 ? 0 : 8) + (dt == MT_F16 ? 0 : 4) + (acc ? 2 : 0) + (aff ? 1 : 0)];
  const int dev = mt_current_device();
  if (mt_device_pending(d, dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { mt_set_error("conv3d (x16s): cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
    mt_mark_device_done(d, dev);
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)P.nwg), dim3(256), ldsb, st, P);
  MT_CHECK_LAUNCH("conv3d_x16s");
  return MT_OK;
}
