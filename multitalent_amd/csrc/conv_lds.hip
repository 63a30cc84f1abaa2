// conv_lds.hip — direct (im2col-free) 3D convolution on fp32 MFMA for gfx950, NDHWC.
//
// Replaces nn.Conv3d on the hot path of the reference (generic_UNet.py:57,67 ConvDropoutNormNonlin,
// conv_blocks.py:49-85,116-213) and, through flipped/transposed packed weights, its autograd
// backward-data; conv_bwdw_kernel replaces the autograd backward-weight.
//
// Design (MI355X_MICROARCH.md: fp32 MFMA = 64 FLOP/clk/SIMD, 64-cycle v_mfma_f32_32x32x2_f32):
//  * a workgroup (4 waves) owns a TD x TH x TW block of output voxels and 32 output channels;
//  * per <=CK input channels it stages the haloed input tile into LDS once (zero padding,
//    zero-insertion for transposed convs, and InstanceNorm+LeakyReLU applied ON LOAD so the
//    normalised activation never round-trips through HBM), channel stride CK+1 (odd) so the
//    per-voxel A-fragment reads are bank-conflict free;
//  * every (tap, channel pair) is one v_mfma_f32_32x32x2_f32: A = 32 voxels x 2 channels from LDS,
//    B = 2 channels x 32 couts straight from the L2-resident pre-packed weights (256 B coalesced,
//    prefetched one tap ahead), accumulating MT 32x32 tiles per wave in registers;
//  * epilogue adds bias, writes NDHWC (optionally into two destinations = split of a concat
//    gradient, optionally accumulating) and emits per-block (sum, sumsq) partials for InstanceNorm.
#include "mt_common.h"

struct ConvChunk { short src, c0, ck, cglob; };

struct ConvKParams {
  mt_conv3d_t c;
  int tilesD, tilesH, tilesW, nsb;
  int nchunks, ntaps;
  ConvChunk chunk[MT_MAX_CHUNKS];
};

// Split the concatenated input channels (C0 | C1) into chunks of <= ck channels that never straddle
// the two sources.  Shared by packing and kernels so the packed order always matches.
static int mt_build_chunks(int C0, int C1, int ck, ConvChunk* out) {
  int n = 0;
  const int Cs[2] = {C0, C1};
  int cglob = 0;
  for (int s = 0; s < 2; ++s) {
    for (int c0 = 0; c0 < Cs[s]; c0 += ck) {
      if (n >= MT_MAX_CHUNKS) return -1;
      const int k = (Cs[s] - c0 < ck) ? (Cs[s] - c0) : ck;
      out[n].src = (short)s; out[n].c0 = (short)c0; out[n].ck = (short)k; out[n].cglob = (short)(cglob + c0);
      ++n;
    }
    cglob += Cs[s];
  }
  return n;
}

// ------------------------------------------------------------------------------------------------
// Input-tile staging shared by forward and backward-weight kernels.
// Tile origin (ud0,uh0,uw0) in VIRTUAL input coordinates, extent LD x LH x LW, channel slots CK.
template <int CK>
__device__ __forceinline__ void mt_stage_input(float* __restrict__ lds, const mt_conv3d_t& c,
                                               const ConvChunk ch, int nb, int ud0, int uh0, int uw0,
                                               int LD, int LH, int LW, int lane, int wave) {
  constexpr int CKP = CK + 1;
  constexpr int VPS = 64 / CK;  // voxels per 64-lane step
  constexpr int U = 4;
  const mt_src_t& S = c.src[ch.src];
  const int cl = lane % CK, vl = lane / CK;
  const bool cvalid = cl < ch.ck;
  const bool has_aff = S.scale != nullptr;
  float sc = 1.f, sh = 0.f;
  if (has_aff && cvalid) {
    sc = S.scale[(size_t)nb * S.C + ch.c0 + cl];
    sh = S.shift[(size_t)nb * S.C + ch.c0 + cl];
  }
  const float slope = S.slope;
  const int nrows = LD * LH;
  for (int row = wave; row < nrows; row += 4) {
    const int ld = row / LH, lhh = row - ld * LH;
    const int ud = ud0 + ld, uh = uh0 + lhh;
    int sd = ud, shh = uh;
    bool rvalid = (ud >= 0) && (uh >= 0);
    if (c.dilD == 2) { rvalid = rvalid && !(ud & 1); sd = ud >> 1; }
    if (c.dilH == 2) { rvalid = rvalid && !(uh & 1); shh = uh >> 1; }
    rvalid = rvalid && (sd < c.Di) && (shh < c.Hi);
    const float* rowp = S.ptr + ((size_t)((size_t)nb * c.Di + sd) * c.Hi + shh) * c.Wi * S.cs + ch.c0 + cl;
    float* ldsrow = lds + (size_t)row * LW * CKP + cl;
    for (int lw0 = 0; lw0 < LW; lw0 += VPS * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int lw = lw0 + u * VPS + vl;
        const int uw = uw0 + lw;
        int sw = uw;
        bool ok = rvalid && cvalid && (lw < LW) && (uw >= 0);
        if (c.dilW == 2) { ok = ok && !(uw & 1); sw = uw >> 1; }
        ok = ok && (sw < c.Wi);
        float x = 0.f;
        if (ok) {
          x = rowp[(size_t)sw * S.cs];
          if (has_aff) x = mt_lrelu(fmaf(x, sc, sh), slope);
        }
        v[u] = x;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int lw = lw0 + u * VPS + vl;
        if (lw < LW) ldsrow[lw * CKP] = v[u];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int MW, int RH, int TD, int CK>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvKParams P) {
  constexpr int MH = 32 / MW;
  constexpr int TH = MH * RH;
  constexpr int TW = MW;
  constexpr int NMT = TD * RH;
  static_assert(NMT % 4 == 0, "M tiles must split over 4 waves");
  constexpr int MT = NMT / 4;
  constexpr int CKP = CK + 1;
  constexpr int NKP = CK / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;

  int tile = mt_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = blockIdx.y;
  const int tw = tile % P.tilesW; tile /= P.tilesW;
  const int th = tile % P.tilesH; tile /= P.tilesH;
  const int td = tile % P.tilesD;
  const int nb = tile / P.tilesD;
  const int sb = (td * P.tilesH + th) * P.tilesW + tw;

  const int LD = (TD - 1) * c.SD + c.KD, LH = (TH - 1) * c.SH + c.KH, LW = (TW - 1) * c.SW + c.KW;
  const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
  const int ud0 = od0 * c.SD - c.PD, uh0 = oh0 * c.SH - c.PH, uw0 = ow0 * c.SW - c.PW;

  // per-lane LDS base of each M tile: voxel (dm, row, col) of the tile, channel half lhalf
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wave * MT + m;
    const int dm = mt / RH, rh = mt % RH;
    const int r = li / MW, col = li % MW;
    const int row = rh * MH + r;
    abase[m] = ((dm * c.SD * LH + row * c.SH) * LW + col * c.SW) * CKP + lhalf;
  }

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;

  const int ntaps = P.ntaps;
  for (int ch = 0; ch < P.nchunks; ++ch) {
    const ConvChunk cc = P.chunk[ch];
    const float* wq = c.wpack + ((size_t)(ntile * P.nchunks + ch) * ntaps) * (NKP * 64) + lane;
    float bcur[NKP], bnxt[NKP];
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) bcur[kp] = wq[kp * 64];

    __syncthreads();  // previous chunk's LDS reads are done
    mt_stage_input<CK>(lds, c, cc, nb, ud0, uh0, uw0, LD, LH, LW, lane, wave);
    __syncthreads();

    const int nkp = (cc.ck + 1) >> 1;
    int tap = 0;
    for (int kd = 0; kd < c.KD; ++kd)
      for (int kh = 0; kh < c.KH; ++kh)
        for (int kw = 0; kw < c.KW; ++kw) {
          const int tapoff = ((kd * LH + kh) * LW + kw) * CKP;
          const int tnext = (tap + 1 < ntaps) ? tap + 1 : tap;
          const float* wn = wq + (size_t)tnext * (NKP * 64);
#pragma unroll
          for (int kp = 0; kp < NKP; ++kp) bnxt[kp] = wn[kp * 64];
#pragma unroll
          for (int kp = 0; kp < NKP; ++kp) {
            if (kp < nkp) {
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                const float a = lds[abase[m] + tapoff + 2 * kp];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[kp], acc[m], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int kp = 0; kp < NKP; ++kp) bcur[kp] = bnxt[kp];
          ++tap;
        }
  }

  // ---- epilogue: bias, store, InstanceNorm partial statistics
  const int co = ntile * 32 + li;
  const bool covalid = co < c.Cout;
  const float bv = (c.bias != nullptr && covalid) ? c.bias[co] : 0.f;
  float* optr;
  int ocs, cofs;
  if (co < c.csplit) { optr = c.out0; ocs = c.ocs0; cofs = co; }
  else               { optr = c.out1; ocs = c.ocs1; cofs = co - c.csplit; }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wave * MT + m;
    const int dm = mt / RH, rh = mt % RH;
    const int od = od0 + dm;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf;
      const int r = iv / MW, col = iv % MW;
      const int oh = oh0 + rh * MH + r, ow = ow0 + col;
      const bool ok = covalid && (od < c.Do) && (oh < c.Ho) && (ow < c.Wo);
      if (ok) {
        const size_t idx = ((size_t)((size_t)((size_t)nb * c.Do + od) * c.Ho + oh) * c.Wo + ow) * ocs + cofs;
        float v = acc[m][j] + bv;
        if (c.accumulate) v += optr[idx];
        optr[idx] = v;
        s1 += v;
        s2 += v * v;
      }
    }
  }
  if (c.stats_part != nullptr) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();  // LDS tile no longer needed
    if (lhalf == 0) { lds[(wave * 32 + li) * 2] = s1; lds[(wave * 32 + li) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 32 && (ntile * 32 + tid) < c.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += lds[(w * 32 + tid) * 2]; t2 += lds[(w * 32 + tid) * 2 + 1]; }
      float* sp = c.stats_part + ((size_t)((size_t)nb * P.nsb + sb) * c.Cout + ntile * 32 + tid) * 2;
      sp[0] = t1; sp[1] = t2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side of mt_conv3d_fwd
struct ConvCfg { int MW, RH, TD, CK; };
static const ConvCfg kCfgs[] = {
  {32, 4, 2, 16}, {16, 2, 2, 16}, {8, 2, 2, 16}, {32, 4, 2, 8}, {16, 2, 2, 8}, {8, 2, 2, 8},
};

static void cfg_tile(const ConvCfg& g, int* TD, int* TH, int* TW) {
  *TD = g.TD; *TH = (32 / g.MW) * g.RH; *TW = g.MW;
}
static size_t cfg_lds(const ConvCfg& g, const mt_conv3d_t* p) {
  int TD, TH, TW; cfg_tile(g, &TD, &TH, &TW);
  const size_t LD = (TD - 1) * p->SD + p->KD, LH = (TH - 1) * p->SH + p->KH, LW = (TW - 1) * p->SW + p->KW;
  size_t b = LD * LH * LW * (g.CK + 1) * sizeof(float);
  return b < 1024 ? 1024 : b;
}
// choose the tile shape with the least padded work that fits LDS; prefer >=2 workgroups per CU
static int pick_cfg(const mt_conv3d_t* p) {
  int best = -1; double bestcost = 1e300;
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const ConvCfg& g = kCfgs[i];
    const size_t l = cfg_lds(g, p);
    if (l > 160 * 1024) continue;
    int TD, TH, TW; cfg_tile(g, &TD, &TH, &TW);
    double vol = (double)mt_cdiv(p->Do, TD) * TD * mt_cdiv(p->Ho, TH) * TH * mt_cdiv(p->Wo, TW) * TW;
    double cost = vol;
    if (l > 80 * 1024) cost *= 1.25;       // one workgroup per CU: no load/compute overlap
    if (g.CK == 8) cost *= 1.10;           // more staging passes
    if (cost < bestcost - 1e-9) { bestcost = cost; best = i; }
  }
  return best;
}

extern "C" int mt_conv3d_ck(const mt_conv3d_t* p) {
  const int i = pick_cfg(p);
  return i < 0 ? -1 : kCfgs[i].CK;
}
extern "C" int mt_conv3d_stats_blocks(const mt_conv3d_t* p) {
  const int i = pick_cfg(p);
  if (i < 0) return -1;
  int TD, TH, TW; cfg_tile(kCfgs[i], &TD, &TH, &TW);
  return mt_cdiv(p->Do, TD) * mt_cdiv(p->Ho, TH) * mt_cdiv(p->Wo, TW);
}

static int conv_validate(const mt_conv3d_t* p) {
  MT_REQUIRE(p != nullptr, "conv3d: null params");
  MT_REQUIRE(p->nsrc == 1 || p->nsrc == 2, "conv3d: nsrc must be 1 or 2 (got %d)", p->nsrc);
  MT_REQUIRE(p->KD >= 1 && p->KD <= 3 && p->KH >= 1 && p->KH <= 3 && p->KW >= 1 && p->KW <= 3, "conv3d: kernel size must be 1..3");
  MT_REQUIRE(p->SD >= 1 && p->SD <= 2 && p->SH >= 1 && p->SH <= 2 && p->SW >= 1 && p->SW <= 2, "conv3d: stride must be 1 or 2");
  MT_REQUIRE((p->dilD == 1 || p->dilD == 2) && (p->dilH == 1 || p->dilH == 2) && (p->dilW == 1 || p->dilW == 2), "conv3d: dilation (zero insertion) must be 1 or 2");
  int csum = p->src[0].C + (p->nsrc == 2 ? p->src[1].C : 0);
  MT_REQUIRE(csum == p->Cin, "conv3d: source channels (%d) != Cin (%d)", csum, p->Cin);
  MT_REQUIRE(p->N > 0 && p->Do > 0 && p->Ho > 0 && p->Wo > 0 && p->Cout > 0 && p->Cin > 0, "conv3d: empty problem");
  MT_REQUIRE(p->wpack != nullptr && p->out0 != nullptr, "conv3d: null pointers");
  MT_REQUIRE(p->csplit >= p->Cout || p->out1 != nullptr, "conv3d: out1 required when csplit < Cout");
  return MT_OK;
}

template <int MW, int RH, int TD, int CK>
static int launch_conv(const mt_conv3d_t* p, const ConvCfg& g, hipStream_t st) {
  ConvKParams P;
  P.c = *p;
  if (P.c.nsrc == 1) { P.c.src[1] = P.c.src[0]; P.c.src[1].C = 0; }
  int TDv, TH, TW; cfg_tile(g, &TDv, &TH, &TW);
  P.tilesD = mt_cdiv(p->Do, TDv); P.tilesH = mt_cdiv(p->Ho, TH); P.tilesW = mt_cdiv(p->Wo, TW);
  P.nsb = P.tilesD * P.tilesH * P.tilesW;
  P.ntaps = p->KD * p->KH * p->KW;
  P.nchunks = mt_build_chunks(p->src[0].C, p->nsrc == 2 ? p->src[1].C : 0, CK, P.chunk);
  MT_REQUIRE(P.nchunks > 0, "conv3d: too many channel chunks (Cin=%d, ck=%d)", p->Cin, CK);
  const size_t ldsb = cfg_lds(g, p);
  dim3 grid((unsigned)(P.nsb * p->N), (unsigned)mt_cdiv(p->Cout, 32), 1);
  auto kfn = conv_fwd_kernel<MW, RH, TD, CK>;
  if (ldsb > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) { mt_set_error("conv3d: cannot raise dynamic LDS to %zu: %s", ldsb, hipGetErrorString(e)); return MT_EHIP; }
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), ldsb, st, P);
  MT_CHECK_LAUNCH("conv3d_fwd");
  return MT_OK;
}

extern "C" int mt_conv3d_fwd(const mt_conv3d_t* p, mt_stream_t stream) {
  int rc = conv_validate(p);
  if (rc != MT_OK) return rc;
  const int i = pick_cfg(p);
  MT_REQUIRE(i >= 0, "conv3d: no tile configuration fits LDS");
  const ConvCfg& g = kCfgs[i];
  hipStream_t st = (hipStream_t)stream;
  switch (i) {
    case 0: return launch_conv<32, 4, 2, 16>(p, g, st);
    case 1: return launch_conv<16, 2, 2, 16>(p, g, st);
    case 2: return launch_conv<8, 2, 2, 16>(p, g, st);
    case 3: return launch_conv<32, 4, 2, 8>(p, g, st);
    case 4: return launch_conv<16, 2, 2, 8>(p, g, st);
    case 5: return launch_conv<8, 2, 2, 8>(p, g, st);
  }
  return MT_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// Weight packing into B-fragment order [ntile][chunk][tap][ck/2][64]:
//   lane l holds W_eff[tap][ci = chunk.cglob + 2*kp + (l>>5)][co = ntile*32 + (l&31)]
struct PackParams {
  const float* w; float* dst;
  int Cout, KD, KH, KW, nkp, nchunks, ntiles, flip;
  long s_ci, s_co, s_kd, s_kh, s_kw;
  ConvChunk chunk[MT_MAX_CHUNKS];
};
__global__ void pack_weights_kernel(const PackParams P) {
  const long total = (long)P.ntiles * P.nchunks * P.KD * P.KH * P.KW * P.nkp * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int l = (int)(r % 64); r /= 64;
    const int kp = (int)(r % P.nkp); r /= P.nkp;
    const int kw = (int)(r % P.KW); r /= P.KW;
    const int kh = (int)(r % P.KH); r /= P.KH;
    const int kd = (int)(r % P.KD); r /= P.KD;
    const int ch = (int)(r % P.nchunks); r /= P.nchunks;
    const int nt = (int)r;
    const ConvChunk cc = P.chunk[ch];
    const int cin_local = 2 * kp + (l >> 5);
    const int co = nt * 32 + (l & 31);
    float v = 0.f;
    if (cin_local < cc.ck && co < P.Cout) {
      const int ci = cc.cglob + cin_local;
      const int zd = P.flip ? P.KD - 1 - kd : kd, zh = P.flip ? P.KH - 1 - kh : kh, zw = P.flip ? P.KW - 1 - kw : kw;
      v = P.w[ci * P.s_ci + co * P.s_co + zd * P.s_kd + zh * P.s_kh + zw * P.s_kw];
    }
    P.dst[i] = v;
  }
}

extern "C" int mt_pack_conv_weights(const float* w, float* dst, size_t* packed_floats, int C0, int C1, int Cout,
                                    int KD, int KH, int KW, long s_ci, long s_co, long s_kd, long s_kh, long s_kw,
                                    int flip, int ck, mt_stream_t stream) {
  MT_REQUIRE(ck >= 2 && (ck % 2) == 0, "pack: ck must be even (got %d)", ck);
  PackParams P;
  P.nchunks = mt_build_chunks(C0, C1, ck, P.chunk);
  MT_REQUIRE(P.nchunks > 0, "pack: too many chunks");
  P.ntiles = mt_cdiv(Cout, 32);
  P.nkp = ck / 2;
  const size_t total = (size_t)P.ntiles * P.nchunks * KD * KH * KW * P.nkp * 64;
  if (packed_floats) *packed_floats = total;
  if (dst == nullptr) return MT_OK;
  MT_REQUIRE(w != nullptr, "pack: null weights");
  P.w = w; P.dst = dst; P.Cout = Cout; P.KD = KD; P.KH = KH; P.KW = KW; P.flip = flip;
  P.s_ci = s_ci; P.s_co = s_co; P.s_kd = s_kd; P.s_kh = s_kh; P.s_kw = s_kw;
  int blocks = mt_cdiv((long)total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("pack_weights");
  return MT_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward-weight:  dW[tap][ci][co] = sum_{n,o} X[n, o*S + t - P, ci] * Y[n, o, co]
// v_mfma_f32_16x16x4_f32: M = 16 input channels (one chunk), N = 16 output channels, K = 4 voxels.
// A workgroup owns (ci chunk, 32 couts) and walks a strided list of spatial tiles, keeping all taps'
// accumulators in registers (taps are dealt round-robin to the 4 waves); it writes ONE partial per
// workgroup, reduced deterministically by bwdw_reduce_kernel straight into the torch weight layout.
struct BwdWParams {
  mt_conv3d_t c;      // X geometry (src), conv geometry; Do/Ho/Wo = Y dims
  mt_src_t y;         // Y source (C = Cout)
  int TD, TH, TW;     // spatial tile (TW % 4 == 0)
  int tilesD, tilesH, tilesW, ntiles_total;
  int nchunks, ntaps, ncot, nsg;
  float* part;        // [chunk][cot][sg][tap][16][32]
  ConvChunk chunk[MT_MAX_CHUNKS];
};

#define BW_CK 16
#define BW_YP 48
#define BW_MAXT 7

__global__ __launch_bounds__(256) void conv_bwdw_kernel(const BwdWParams P) {
  constexpr int CK = BW_CK, CKP = CK + 1, YP = BW_YP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const mt_conv3d_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int sg = blockIdx.x, cot = blockIdx.y, chi = blockIdx.z;
  const ConvChunk cc = P.chunk[chi];
  const int TD = P.TD, TH = P.TH, TW = P.TW, TV = TD * TH * TW;
  const int LD = (TD - 1) * c.SD + c.KD, LH = (TH - 1) * c.SH + c.KH, LW = (TW - 1) * c.SW + c.KW;
  float* xl = lds;
  float* yl = lds + (size_t)LD * LH * LW * CKP;

  // taps handled by this wave: wave, wave+4, ...
  int tapoff[BW_MAXT];
  int mytaps = 0;
#pragma unroll
  for (int t = 0; t < BW_MAXT; ++t) {
    const int tap = wave + 4 * t;
    tapoff[t] = 0;
    if (tap < P.ntaps) {
      const int kw = tap % c.KW, kh = (tap / c.KW) % c.KH, kd = tap / (c.KW * c.KH);
      tapoff[t] = ((kd * LH + kh) * LW + kw) * CKP;
      mytaps = t + 1;
    }
  }
  f32x4 acc[BW_MAXT][2];
#pragma unroll
  for (int t = 0; t < BW_MAXT; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][h][j] = 0.f;

  const mt_src_t& Y = P.y;
  for (int tile = sg; tile < P.ntiles_total; tile += P.nsg) {
    int r = tile;
    const int tw = r % P.tilesW; r /= P.tilesW;
    const int th = r % P.tilesH; r /= P.tilesH;
    const int td = r % P.tilesD;
    const int nb = r / P.tilesD;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
    __syncthreads();
    mt_stage_input<CK>(xl, c, cc, nb, od0 * c.SD - c.PD, oh0 * c.SH - c.PH, ow0 * c.SW - c.PW, LD, LH, LW, lane, wave);
    // stage Y tile: [TV][32 couts]
    {
      const int col = tid & 31;
      const int co = cot * 32 + col;
      const bool cok = co < c.Cout;
      float ysc = 1.f, ysh = 0.f;
      const bool yaff = Y.scale != nullptr;
      if (yaff && cok) { ysc = Y.scale[(size_t)nb * Y.C + co]; ysh = Y.shift[(size_t)nb * Y.C + co]; }
      for (int v = tid >> 5; v < TV; v += 8) {
        const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
        const int od = od0 + d, oh = oh0 + h, ow = ow0 + w;
        float x = 0.f;
        if (cok && od < c.Do && oh < c.Ho && ow < c.Wo) {
          x = Y.ptr[((size_t)((size_t)((size_t)nb * c.Do + od) * c.Ho + oh) * c.Wo + ow) * Y.cs + co];
          if (yaff) x = mt_lrelu(fmaf(x, ysc, ysh), Y.slope);
        }
        yl[v * YP + col] = x;
      }
    }
    __syncthreads();
    for (int v0 = 0; v0 < TV; v0 += 4) {
      const int v = v0 + lk;  // this lane's voxel (K index)
      const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
      const int xb = ((d * c.SD * LH + h * c.SH) * LW + w * c.SW) * CKP + li;
      const float b0 = yl[v * YP + li];
      const float b1 = yl[v * YP + 16 + li];
#pragma unroll
      for (int t = 0; t < BW_MAXT; ++t) {
        if (t < mytaps) {
          const float a = xl[xb + tapoff[t]];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[t][1], 0, 0, 0);
        }
      }
    }
  }
  // write partial: D layout of 16x16x4: row (ci) = (lane>>4)*4 + j, col (co) = lane&15
  float* pp = P.part + ((size_t)((size_t)(chi * P.ncot + cot) * P.nsg + sg) * P.ntaps) * (16 * 32);
#pragma unroll
  for (int t = 0; t < BW_MAXT; ++t) {
    const int tap = wave + 4 * t;
    if (tap < P.ntaps) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) pp[(size_t)tap * 512 + (lk * 4 + j) * 32 + h * 16 + li] = acc[t][h][j];
    }
  }
}

struct BwdWReduceParams {
  const float* part; float* dw;
  int Cin, Cout, KD, KH, KW, nchunks, ncot, nsg, ntaps, accumulate;
  long s_ci, s_co, s_kd, s_kh, s_kw;
  ConvChunk chunk[MT_MAX_CHUNKS];
};
__global__ void bwdw_reduce_kernel(const BwdWReduceParams P) {
  // one thread per (chunk, cot, tap, ci16, co32)
  const long total = (long)P.nchunks * P.ncot * P.ntaps * 512;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int col = (int)(r % 32); r /= 32;
    const int cil = (int)(r % 16); r /= 16;
    const int tap = (int)(r % P.ntaps); r /= P.ntaps;
    const int cot = (int)(r % P.ncot); r /= P.ncot;
    const int chi = (int)r;
    const ConvChunk cc = P.chunk[chi];
    const int co = cot * 32 + col;
    if (cil >= cc.ck || co >= P.Cout) continue;
    const float* pp = P.part + ((size_t)(chi * P.ncot + cot) * P.nsg * P.ntaps + tap) * 512 + cil * 32 + col;
    double s = 0.0;
    for (int g = 0; g < P.nsg; ++g) s += (double)pp[(size_t)g * P.ntaps * 512];
    const int kw = tap % P.KW, kh = (tap / P.KW) % P.KH, kd = tap / (P.KW * P.KH);
    const long o = (long)(cc.cglob + cil) * P.s_ci + (long)co * P.s_co + kd * P.s_kd + kh * P.s_kh + kw * P.s_kw;
    if (P.accumulate) P.dw[o] += (float)s; else P.dw[o] = (float)s;
  }
}

static void bwdw_plan(const mt_conv3d_t* p, BwdWParams* P) {
  // tile: rows of up to 32 voxels in W (multiple of 4), ~128 voxels per tile
  int TW = p->Wo >= 32 ? 32 : ((p->Wo + 3) / 4) * 4;
  int TH = 128 / TW; if (TH > p->Ho) TH = p->Ho; if (TH < 1) TH = 1;
  int TD = 128 / (TW * TH); if (TD > p->Do) TD = p->Do; if (TD < 1) TD = 1;
  // keep the haloed X tile within ~48 KiB for strided convs
  for (;;) {
    const size_t LD = (TD - 1) * p->SD + p->KD, LH = (TH - 1) * p->SH + p->KH, LW = (TW - 1) * p->SW + p->KW;
    const size_t b = (LD * LH * LW * (BW_CK + 1) + (size_t)TD * TH * TW * BW_YP) * sizeof(float);
    if (b <= 72 * 1024 || (TD == 1 && TH == 1)) break;
    if (TD > 1) TD = (TD + 1) / 2; else TH = (TH + 1) / 2;
  }
  P->TD = TD; P->TH = TH; P->TW = TW;
  P->tilesD = mt_cdiv(p->Do, TD); P->tilesH = mt_cdiv(p->Ho, TH); P->tilesW = mt_cdiv(p->Wo, TW);
  P->ntiles_total = P->tilesD * P->tilesH * P->tilesW * p->N;
  P->ntaps = p->KD * p->KH * p->KW;
  P->nchunks = mt_build_chunks(p->src[0].C, p->nsrc == 2 ? p->src[1].C : 0, BW_CK, P->chunk);
  P->ncot = mt_cdiv(p->Cout, 32);
  // spatial groups: fill ~2 workgroups per CU over all (chunk, cot) pairs
  int pairs = P->nchunks * P->ncot; if (pairs < 1) pairs = 1;
  int nsg = (512 + pairs - 1) / pairs;
  if (nsg > P->ntiles_total) nsg = P->ntiles_total;
  if (nsg < 1) nsg = 1;
  P->nsg = nsg;
}

extern "C" size_t mt_conv3d_bwd_weight_workspace(const mt_conv3d_t* p) {
  if (p == nullptr) return 0;
  BwdWParams P; bwdw_plan(p, &P);
  if (P.nchunks <= 0) return 0;
  return (size_t)P.nchunks * P.ncot * P.nsg * P.ntaps * 512 * sizeof(float);
}

extern "C" int mt_conv3d_bwd_weight(const mt_conv3d_t* p, const mt_src_t* ysrc, float* dw, long s_ci, long s_co,
                                    long s_kd, long s_kh, long s_kw, int accumulate, void* workspace,
                                    size_t workspace_bytes, mt_stream_t stream) {
  MT_REQUIRE(p != nullptr && ysrc != nullptr && dw != nullptr, "bwd_weight: null argument");
  MT_REQUIRE(p->nsrc == 1 || p->nsrc == 2, "bwd_weight: nsrc must be 1 or 2");
  MT_REQUIRE(p->KD >= 1 && p->KD <= 3 && p->KH >= 1 && p->KH <= 3 && p->KW >= 1 && p->KW <= 3, "bwd_weight: kernel size must be 1..3");
  MT_REQUIRE(p->dilD == 1 && p->dilH == 1 && p->dilW == 1, "bwd_weight: dilation unsupported");
  MT_REQUIRE(ysrc->C == p->Cout, "bwd_weight: ysrc.C (%d) != Cout (%d)", ysrc->C, p->Cout);
  BwdWParams P;
  P.c = *p;
  if (P.c.nsrc == 1) { P.c.src[1] = P.c.src[0]; P.c.src[1].C = 0; }
  P.y = *ysrc;
  bwdw_plan(p, &P);
  MT_REQUIRE(P.nchunks > 0, "bwd_weight: too many channel chunks");
  MT_REQUIRE(P.ntaps <= 4 * BW_MAXT, "bwd_weight: too many taps");
  const size_t need = (size_t)P.nchunks * P.ncot * P.nsg * P.ntaps * 512 * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) { mt_set_error("bwd_weight: workspace %zu < %zu", workspace_bytes, need); return MT_EWORKSPACE; }
  P.part = (float*)workspace;
  const size_t LD = (P.TD - 1) * p->SD + p->KD, LH = (P.TH - 1) * p->SH + p->KH, LW = (P.TW - 1) * p->SW + p->KW;
  const size_t ldsb = (LD * LH * LW * (BW_CK + 1) + (size_t)P.TD * P.TH * P.TW * BW_YP) * sizeof(float);
  MT_REQUIRE(ldsb <= 160 * 1024, "bwd_weight: LDS tile too large (%zu)", ldsb);
  hipStream_t st = (hipStream_t)stream;
  if (ldsb > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_bwdw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) { mt_set_error("bwd_weight: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
  }
  hipLaunchKernelGGL(conv_bwdw_kernel, dim3(P.nsg, P.ncot, P.nchunks), dim3(256), ldsb, st, P);
  MT_CHECK_LAUNCH("conv_bwdw");
  BwdWReduceParams R;
  R.part = P.part; R.dw = dw; R.Cin = p->Cin; R.Cout = p->Cout; R.KD = p->KD; R.KH = p->KH; R.KW = p->KW;
  R.nchunks = P.nchunks; R.ncot = P.ncot; R.nsg = P.nsg; R.ntaps = P.ntaps; R.accumulate = accumulate;
  R.s_ci = s_ci; R.s_co = s_co; R.s_kd = s_kd; R.s_kh = s_kh; R.s_kw = s_kw;
  for (int i = 0; i < P.nchunks; ++i) R.chunk[i] = P.chunk[i];
  const long total = (long)P.nchunks * P.ncot * P.ntaps * 512;
  int blocks = mt_cdiv(total, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bwdw_reduce_kernel, dim3(blocks), dim3(256), 0, st, R);
  MT_CHECK_LAUNCH("bwdw_reduce");
  return MT_OK;
}

