// conv_x16.hip — stride-1 3x3x3 / 1x3x3 convolution with 16-bit storage on all operands (fp16 forward over activations, bf16
// backward-data over gradients), v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulation.  Mixed-precision mode of the reference (autocast:
// nnUNetTrainerV2.py:236-249, MultiTalent_Trainer_DDP.py:340-352; the layers are generic_UNet.py:28-70 ConvDropoutNormNonlin and
// conv_blocks.py:116-213 BasicResidualBlock).
//
// Why a second kernel beside conv_bf16_kernel (conv_bf16.inc): that kernel's three phases ADD because every one of them
// loads the vector-memory path: per 16-channel chunk and workgroup its four waves fetch the SAME 27 weight fragments from L1 (108 KB
// through a 64 B/clk path = half the chunk's matrix time), stage the input patch as 8-byte pieces and leave through 128 dword stores per
// tile; and every weight-fragment wait inside the MFMA loop (vmcnt is in-order) forbids a register prefetch of the next chunk.  Here
//   * the chunk's weight fragments go to LDS once per workgroup (27 KiB, ds_read_b128 = 256 B/clk): the MFMA loop issues no vector-memory
//     instruction, so the raw patch AND the weights of the NEXT (tile, chunk) step are in flight in registers while this one multiplies
//     (persistent workgroups: the prefetch runs across tile boundaries);
//   * an input-row fragment is read once per (kd, kw) and used for the three kh taps of the wave's four output rows (6 fragment reads
//     per 12 MFMAs instead of 12);
//   * the patch is fetched as 16-byte pieces (two lanes per voxel: 10 loads per lane and chunk instead of 20);
//   * the epilogue goes through a wave-private LDS image: a tile row leaves as 16-byte stores (8 per wave and tile instead of 32).
// Tile 4 x 4 x 32 outputs x 32 output channels, four waves (wave = output plane), two workgroups per CU.
#include "bwdw_common.h"
#include "conv16_common.h"


#ifndef X16_ABL
#define X16_ABL 0    // compile-time timing ablations: 1 no patch / weight loads, 2 no conversion + LDS writes, 4 no epilogue, 8 no MFMAs, 16 no wait for the weight DMA, 32 no global stores, 64 epilogue without transposed reads + stores, 128 epilogue computes the values only
#endif

#include "conv_x16_epi.inc"

#ifndef X16_TS
#define X16_TS 0     // 1: per-phase s_memtime totals of every wave -> (long long*)c.out1 [workgroup][wave][8] (tools/bench_fwd16.py --ts)
#endif
#if X16_TS
#define X16_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[k] += t_ - ts_last; ts_last = t_; }
#else
#define X16_STAMP(k)
#endif

template <int ST> __device__ __forceinline__ unsigned x16_act2(unsigned d, float sc0, float sh0, float sc1, float sh1, float slope) {
  const float t0 = __builtin_fmaf(mt_lo16<ST>(d), sc0, sh0), t1 = __builtin_fmaf(mt_hi16<ST>(d), sc1, sh1);
  return mt_pk16<ST>(fmaxf(t0, t0 * slope), fmaxf(t1, t1 * slope));
}

template <int KD, int ST, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_x16_kernel(const X16Params P) {
  constexpr int TD = 4, TH = 4, TW = 32;
  constexpr int LD = TD + KD - 1, LH = TH + 2, LW = TW + 2;
  constexpr int ROWS = LD * LH, RPW = ROWS / 4;               // 36 / 9 (3x3x3), 24 / 6 (1x3x3)
  constexpr int NTAP = KD * 9, PD = (KD - 1) / 2;
  constexpr int A_BYTES = ROWS * LW * 32, B_BYTES = NTAP * 1024, NBP = (B_BYTES / 16 + 255) / 256;
  static_assert(ROWS % 4 == 0, "rows split over the four waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char x16_lds[];
  unsigned char* const ldsA = x16_lds;
  unsigned char* const ldsB = x16_lds + A_BYTES;
  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const stg = ldsB + B_BYTES + wave * 2048;    // wave-private epilogue image: 32 voxels x 32 channels x 2 bytes
  float* const tab = (float*)(ldsB + B_BYTES + 4 * 2048);     // [chunk][scale | shift][16] of the current sample
  float* const tabb = tab + P.nchunks * 32;                    // bias of every output channel (zeros behind Cout up to the next multiple of 32)
  const int li = lane & 31, lhalf = lane >> 5;

  // ---- this workgroup's contiguous range of (spatial tile, cout tile) items; an XCD walks a contiguous range (shared halos and weights)
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

  // ---- per-lane constants of the staging pass.  Main piece: lane = (voxel column 1 + lane/2, channel half lane & 1) of the wave's rows
  // w, w + 4, ...; halo piece: the columns 0 and LW - 1 of all rows, one piece for each of the first 4 * ROWS threads.
  const int hf = lane & 1;
  const int lwm = 1 + (lane >> 1);
  const int ldsA_main = lwm * 32 + (((hf ^ (lwm >> 3)) & 1) * 16);
  const bool has_halo = tid < ROWS * 4;
  const int hrow = tid >> 2;
  const int lwh = ((tid >> 1) & 1) ? (LW - 1) : 0;
  const int hld = hrow / LH, hlh = hrow - hld * LH;
  const int ldsA_halo = (hrow * LW + lwh) * 32 + (((hf ^ (lwh >> 3)) & 1) * 16);
  // MFMA phase: byte offset of (plane = wave, row 0, column li + kw, half) — the compact image of conv_bf16.inc: the two 16-byte channel
  // halves of a voxel are swapped where bit 3 of its column is set (conflict-free ds_read_b128 without padding)
  int abase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = li + kw;
    abase[kw] = (wave * LH * LW + col) * 32 + (((lhalf ^ (col >> 3)) & 1) * 16);
  }

  // ---- the raw patch of a chunk PAIR (32 consecutive channels of one source = 64 bytes of every voxel) is requested at once: the two
  // chunks' pieces of a voxel lie in the same cache lines, and what bounds this kernel is the number of lines that move from L2 to L1
  // (measured: the second piece of a line costs 0.1 - 0.35 of the first; fetched a step apart it costs the same again).  Set 0 is
  // written to LDS in the pair's first step, set 1 waits in registers for the second.
  x16_u32x4 ra[2][RPW], rh[2];
  auto issue = [&](const X16Geo& g, int pr, bool live) __attribute__((always_inline)) {
    const int ch0 = P.pair[pr][0], ch1 = P.pair[pr][1];
    const ConvChunk cc = P.chunk[ch0];
    const mt_src_t& S = c.src[cc.src];
    const int cs = S.cs;
    const size_t sample_bytes = (size_t)c.Di * c.Hi * c.Wi * cs * 2;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)g.nb * sample_bytes), 0, (int)sample_bytes, 0x00020000);
    const int ud0 = g.od0 - PD, uh0 = g.oh0 - 1, uw0 = g.ow0 - 1;
    const int rowbytes = c.Wi * cs * 2;
    const int uwm = uw0 + lwm;
    const int voffm = (live && (unsigned)uwm < (unsigned)c.Wi) ? (uwm * cs + cc.c0 + 8 * hf) * 2 : (int)0x80000000;
    const int second = (ch1 >= 0) ? 0 : (int)0x80000000;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int row = wave + 4 * q;
      const int ld = row / LH, lh = row - ld * LH;
      const int ud = ud0 + ld, uh = uh0 + lh;
      const int rm = (((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi)) ? -1 : 0;
      const int soff = ((ud * c.Hi + uh) & rm) * rowbytes;
      const int vo = voffm | (~rm & (int)0x80000000);
      ra[0][q] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
      ra[1][q] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (vo + 32) | second, soff, 0));
    }
    {
      const int ud = ud0 + hld, uh = uh0 + hlh, uw = uw0 + lwh;
      const bool ok = live && has_halo && ((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi);
      const int voff = ok ? (((ud * c.Hi + uh) * c.Wi + uw) * cs + cc.c0 + 8 * hf) * 2 : (int)0x80000000;
      rh[0] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
      rh[1] = __builtin_bit_cast(x16_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (voff + 32) | second, 0, 0));
    }
  };
  // the chunk's weight fragments: HBM / L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at base + 16 l), no
  // registers.  Issued behind the trailing barrier of the previous step, complete (vmcnt(0)) before the barrier that opens the MFMA phase.
  auto issue_weights = [&](int ntile, int ch) __attribute__((always_inline)) {
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)c.wpack + (size_t)(ntile * P.nchunks + ch) * B_BYTES), 0, B_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < NBP; ++i)
      if ((i + 1) * 256 * 16 <= B_BYTES || (wave * 64 + 256 * i) * 16 < B_BYTES)        // (wave-uniform: the last instruction covers the first waves only)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(ldsB + (wave * 64 + 256 * i) * 16), 16, (tid + 256 * i) * 16, 0, 0, 0);
  };

  auto convert = [&](const X16Geo& g, int ch, const x16_u32x4 (&xa)[RPW], const x16_u32x4& xh) {
    const ConvChunk cc = P.chunk[ch];
    const mt_src_t& S = c.src[cc.src];
    const bool aff = S.scale != nullptr;
    const float slope = aff ? S.slope : 1.f;
    unsigned cm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int ce = 8 * hf + 2 * d;
      cm[d] = ((ce < cc.ck) ? 0x0000ffffu : 0u) | ((ce + 1 < cc.ck) ? 0xffff0000u : 0u);
    }
    float sc[8], sh[8];
    if (aff) {
      const f32x4* tp = (const f32x4*)(tab + ch * 32 + 8 * hf);
      const f32x4 s0 = tp[0], s1 = tp[1], h0 = tp[4], h1 = tp[5];
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
    }
    const int ud0 = g.od0 - PD, uh0 = g.oh0 - 1, uw0 = g.ow0 - 1;
    auto put = [&](x16_u32x4 v, unsigned vm, int byteoff) __attribute__((always_inline)) {
      if (aff) {
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = x16_act2<ST>(v[d], sc[2 * d], sh[2 * d], sc[2 * d + 1], sh[2 * d + 1], slope);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) v[d] &= cm[d] & vm;
      *(x16_u32x4*)(ldsA + byteoff) = v;
    };
    const unsigned colm = ((unsigned)(uw0 + lwm) < (unsigned)c.Wi) ? 0xffffffffu : 0u;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int row = wave + 4 * q;
      const int ld = row / LH, lh = row - ld * LH;
      const int ud = ud0 + ld, uh = uh0 + lh;
      const unsigned rm = (((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi)) ? 0xffffffffu : 0u;
      put(xa[q], rm & colm, row * (LW * 32) + ldsA_main);
    }
    if (has_halo) {
      const int ud = ud0 + hld, uh = uh0 + hlh, uw = uw0 + lwh;
      const bool ok = ((unsigned)ud < (unsigned)c.Di) && ((unsigned)uh < (unsigned)c.Hi) && ((unsigned)uw < (unsigned)c.Wi);
      put(xh, ok ? 0xffffffffu : 0u, ldsA_halo);
    }
  };

  f32x16 acc[1][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0][m][j] = 0.f;
  };
  zero_acc();

  // ---- one chunk: groups g = (kd, kw); the six input rows of the group feed 3 (kh) x 4 (output rows) MFMAs
  auto mfma_phase = [&]() __attribute__((always_inline)) {
    constexpr int NG = KD * 3;
    bf16x8 a[6], b[2][3];
    auto lda = [&](int g, int r) __attribute__((always_inline)) { return *(const bf16x8*)(ldsA + abase[g % 3] + ((g / 3) * LH + r) * (LW * 32)); };
    auto ldb = [&](int g, int kh) __attribute__((always_inline)) { return *(const bf16x8*)(ldsB + ((g / 3) * 9 + kh * 3 + (g % 3)) * 1024 + lane * 16); };
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = lda(0, r);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) b[0][kh] = ldb(0, kh);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {        // input row r feeds output row m = r - kh of tap row kh; its register is refilled right behind its last use
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if (r - kh >= 0 && r - kh < 4) acc[0][r - kh] = mt_mfma16<ST>(a[r], b[g & 1][kh], acc[0][r - kh]);
        if (g + 1 < NG) {
          a[r] = lda(g + 1, r);
          if (r < 3) b[(g + 1) & 1][r] = ldb(g + 1, r);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

#if X16_TS
  unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts_last = __builtin_amdgcn_s_memtime();
#endif
  X16Geo cur, nxt;
  decode(it0, cur);
  nxt = cur;
  int tab_nb = -1;
  for (int e = tid; e < P.ncot * 32; e += 256) tabb[e] = (c.bias != nullptr && e < c.Cout) ? c.bias[e] : 0.f;      // (published by the first table barrier)
  if (!(X16_ABL & 1)) { issue(cur, 0, true); issue_weights(cur.ntile, P.pair[0][0]); }
  for (int item = it0; item < it1; ++item) {
    if (cur.nb != tab_nb) {
      // (every wave is past the previous step's trailing barrier: nobody reads the table any more)
      for (int e = tid; e < P.nchunks * 32; e += 256) {
        const int ch = e >> 5, k = e & 15;
        const ConvChunk cc = P.chunk[ch];
        const mt_src_t& S = c.src[cc.src];
        float v = 0.f;
        if (S.scale != nullptr && k < cc.ck) v = ((e >> 4) & 1) ? S.shift[(size_t)cur.nb * S.C + cc.c0 + k] : S.scale[(size_t)cur.nb * S.C + cc.c0 + k];
        tab[e] = v;
      }
      tab_nb = cur.nb;
      __syncthreads();
    }
    const bool more = item + 1 < it1;
    for (int pr = 0; pr < P.npairs; ++pr) {
      const int ch0 = P.pair[pr][0], ch1 = P.pair[pr][1];
      const bool lastpair = pr + 1 == P.npairs;
      // one step = one 16-channel chunk: front (conversion of the chunk's raw pieces -> LDS, barrier, MFMAs) and back (trailing barrier,
      // weight DMA of the next step).  SUB = 0: the pair's first chunk (register set 0); SUB = 1: its second (set 1).  The step that ends
      // a pair requests the next pair's patch (of this tile, or of the next tile) before its MFMAs.
      const bool last = lastpair;                             // (of the step that ends this pair): last step of the tile
      const bool live = !last || more;
      auto front = [&](auto SUB, int ch, bool ends_pair) __attribute__((always_inline)) {
        constexpr int sub = decltype(SUB)::value;
        X16_STAMP(7)
        if (!(X16_ABL & 2)) convert(cur, ch, ra[sub], rh[sub]);
        X16_STAMP(0)
        if (!(X16_ABL & 16)) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this step's weight fragments have landed (LDS-DMA).  The BUILTIN, not inline asm: the compiler's wait-count pass must see it, or it drains vmcnt again in front of every later LDS read that might alias the DMA (it did: four times per epilogue, each time waiting for the stores just issued)
        __syncthreads();
        X16_STAMP(1)
        if (ends_pair) {
          if (last && live) decode(item + 1, nxt);
          if (!(X16_ABL & 1)) issue(last ? nxt : cur, last ? 0 : pr + 1, live);
        }
        __builtin_amdgcn_sched_barrier(0);
        X16_STAMP(2)
        if (!(X16_ABL & 8)) mfma_phase();
        X16_STAMP(3)
      };
      auto back = [&](int ntile_next, int ch_next, bool have_next) __attribute__((always_inline)) {
        X16_STAMP(4)
        __syncthreads();
        X16_STAMP(5)
        if (have_next && !(X16_ABL & 1)) issue_weights(ntile_next, ch_next);
      };
      front(std::integral_constant<int, 0>(), ch0, ch1 < 0);
      if (ch1 >= 0) {
        back(cur.ntile, ch1, true);
        front(std::integral_constant<int, 1>(), ch1, true);
      }
      if (last) {
        if (!(X16_ABL & 4)) x16_epilogue<ST, ACC>(c, acc, stg, tabb, cur, wave, lane);
        else if (acc[0][0][0] == 12345.678f) ((float*)c.out0)[0] = acc[0][3][3];
        zero_acc();
      }
      back(last ? nxt.ntile : cur.ntile, P.pair[last ? 0 : pr + 1][0], live);
      if (last && c.stats_part != nullptr && tid < 32 && !(X16_ABL & 4)) {
        const int co = cur.ntile * 32 + tid;
        if (co < c.Cout) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float* sp = (const float*)(ldsB + B_BYTES + w * 2048);
            t1 += sp[tid * 2]; t2 += sp[tid * 2 + 1];
          }
          float* dst = c.stats_part + ((size_t)((size_t)cur.nb * P.nsb + cur.sb) * c.Cout + co) * 2;
          dst[0] = t1; dst[1] = t2;
        }
      }
    }
    cur = nxt;
  }
#if X16_TS
  if (lane == 0 && c.out1 != nullptr)
    for (int k = 0; k < 8; ++k) ((unsigned long long*)c.out1)[((size_t)blockIdx.x * 4 + wave) * 8 + k] = ts_acc[k];
#endif
}

static size_t x16_lds_bytes(int KD, int nchunks, int ncot) {
  const int rows = (4 + KD - 1) * 6;
  return (size_t)rows * 34 * 32 + (size_t)KD * 9 * 1024 + 4 * 2048 + (size_t)nchunks * 128 + (size_t)ncot * 128;
}

int mt_conv_x16_workgroups(int nitems) {
  const int cap = 2 * mt_device_cus(mt_current_device());
  return nitems < cap ? nitems : cap;
}

int mt_launch_conv_x16(const X16Params& P, int KD, int dt, hipStream_t st) {
  const size_t ldsb = x16_lds_bytes(KD, P.nchunks, P.ncot);
  MT_REQUIRE(ldsb <= 160 * 1024, "conv3d (x16): %zu bytes of LDS", ldsb);
  MT_REQUIRE((KD == 1 || KD == 3) && (dt == MT_F16 || dt == MT_BF16), "conv3d (x16): KD %d, type %d", KD, dt);
  const bool acc = P.c.accumulate != 0;
  void (*kfn)(const X16Params);
  if (KD == 3) kfn = dt == MT_F16 ? (acc ? conv_x16_kernel<3, MT_F16, true> : conv_x16_kernel<3, MT_F16, false>)
                                  : (acc ? conv_x16_kernel<3, MT_BF16, true> : conv_x16_kernel<3, MT_BF16, false>);
  else kfn = dt == MT_F16 ? (acc ? conv_x16_kernel<1, MT_F16, true> : conv_x16_kernel<1, MT_F16, false>)
                          : (acc ? conv_x16_kernel<1, MT_BF16, true> : conv_x16_kernel<1, MT_BF16, false>);
  static std::atomic<uint64_t> done[8];
  std::atomic<uint64_t>& d = done[(KD == 3 ? 0 : 4) + (dt == MT_F16 ? 0 : 2) + (acc ? 1 : 0)];
  const int dev = mt_current_device();
  if (mt_device_pending(d, dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { mt_set_error("conv3d (x16): cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
    mt_mark_device_done(d, dev);
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)P.nwg), dim3(256), ldsb, st, P);
  MT_CHECK_LAUNCH("conv3d_x16");
  return MT_OK;
}
