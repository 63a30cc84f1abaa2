// pointwise.hip — 1x1x1 convolutions and kernel==stride transposed convolutions on fp32 MFMA.
//
// Replaces: seg heads nn.Conv3d(C, num_classes, 1) (generic_UNet.py:349-351; generic_modular_UNet.py:244,251),
// strided 1x1x1 skip projections (conv_blocks.py:192-197), nn.ConvTranspose3d(k == stride, bias=False)
// (generic_UNet.py:335-336; generic_modular_UNet.py:236-237) and, with transposed packed weights,
// the backward-data of the 1x1x1 convs.
//
// For a transposed conv every tap is an independent GEMM whose rows are scattered to out[base*so + tap] — written
// directly into the first half of the skip-concat buffer (ocs).  Weights: mt_pack_conv_weights(layout 1, ck 16).
#include "mt_common.h"
#include <stdlib.h>
#include <string.h>

struct PwKParams {
  mt_pointwise_t c;
  int ntaps, nchunks, nsb;
  long Vb;
  int wide;      // pw_fast_kernel: transposed-conv outputs leave through LDS as 16-byte stores (see the wide epilogue)
};

#define PW_MAXC 1024   // largest Cin (rounded up to a chunk) whose scale/shift fit the LDS copy
#define PW_CK 16   // channels per K chunk (packed weight layout 1, ck = 16 — the conv kernels' layout)

// ---- storage types (mt_src_t.dtype, odtype; mt_common.h).  The matrix arithmetic of this file is fp32 either way: a 16-bit source
// is widened on load (8 channels = ONE 16-byte load instead of two), a 16-bit destination rounded on store.
// 16-bit output of a 32x32 accumulator tile as channel-pair dwords: the lanes of a channel pair (li even, li odd) trade one value per
// two accumulator rows, so the EVEN lane holds both channels of voxel row j and the ODD lane both channels of row j + 1.
__device__ __forceinline__ void pw_pair_exchange(float vj, float vj1, bool odd, float& a, float& b) {
  const float send = odd ? vj : vj1;
  const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  a = odd ? recv : vj;
  b = odd ? vj1 : recv;
}
__device__ __forceinline__ float pw_pair_combine(float s0, float s1, bool odd) {       // per-channel total of sums kept per pair member
  const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
  const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
  return odd ? s1 + t1 : s0 + t0;
}
// 8 consecutive channels of one voxel (byte offset o inside the buffer): two 16-byte loads (fp32) or one (16-bit)
template <int XS>
__device__ __forceinline__ void pw_load8(__amdgpu_buffer_rsrc_t r, int o, float (&x)[8]) {
  if constexpr (XS == MT_F32) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + g * 16, 0, 0));
      x[4 * g] = t[0]; x[4 * g + 1] = t[1]; x[4 * g + 2] = t[2]; x[4 * g + 3] = t[3];
    }
  } else {
    const uint4 t = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
    x[0] = mt_lo16<XS>(t.x); x[1] = mt_hi16<XS>(t.x); x[2] = mt_lo16<XS>(t.y); x[3] = mt_hi16<XS>(t.y);
    x[4] = mt_lo16<XS>(t.z); x[5] = mt_hi16<XS>(t.z); x[6] = mt_lo16<XS>(t.w); x[7] = mt_hi16<XS>(t.w);
  }
}

// One wave = 32 base voxels x 32 output channels x NT taps.  Lane (i, h) holds channels 8h..8h+7 of voxel i for the current
// 16-channel chunk (one 32-byte vector straight from global memory, lazy InstanceNorm+LeakyReLU applied in registers), so a
// chunk costs 8 MFMAs per tap with no LDS traffic at all; every tap of a transposed conv accumulates into its own
// accumulator tile (NT*16 AGPRs) and the input is read exactly once.  Outputs leave through buffer stores whose per-row
// offsets are computed once (out-of-range rows carry the hardware-masked offset).
#ifndef PW_ABL
#define PW_ABL 0      // timing ablations of pw_fast_kernel: 1 no stores, 2 no weight-fragment loads, 4 no MFMAs
#endif
// M16 (mixed precision, fp16 source, mt_pointwise_t.mma == 1): the activated fragment is rounded to fp16 and multiplied by pack-layout-4
// weights — one v_mfma_f32_32x32x16_f16 per (chunk, tap) instead of eight fp32 MFMAs (the forward transposed convs ran AT the fp32 matrix rate)
template <int NT, int VEC, int XS = MT_F32, int OS = MT_F32, bool M16 = false>
__global__ __launch_bounds__(256) void pw_fast_kernel(const PwKParams P) {
  constexpr int XE = mt_ebytes<XS>(), OE = mt_ebytes<OS>();     // bytes per stored element
  const mt_pointwise_t& c = P.c;
  __shared__ float red[4 * 32 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;
  const int bx = mt_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bx / P.nsb, sb = bx % P.nsb;
  const int ntile = blockIdx.y;
  const long m0 = (long)sb * 128 + wave * 32;
  const mt_src_t& S = c.src;

  // ---- A operand: this lane's base voxel, channels 8*lhalf .. +7 of each chunk
  const long bv = m0 + li;
  const bool vok = bv < P.Vb;
  // (d, h, w) of the wave's first voxel by ONE wave-uniform division; a row 0..31 further only carries once into h and once into
  // d when the rows are at least 32 voxels long (otherwise: the general division per lane).  The per-row divisions of the first
  // version (1 + 16 of them per lane, ~40 VALU each) cost more than the 16 MFMAs of the tile.
  const int w0 = (int)(m0 % c.Wb), h0 = (int)((m0 / c.Wb) % c.Hb), d0 = (int)(m0 / ((long)c.Wb * c.Hb));
  const bool longrows = c.Wb >= 32;
  auto row_dhw = [&](int iv, int& d, int& h, int& w) {
    if (longrows) {
      w = w0 + iv; h = h0; d = d0;
      const bool cw = w >= c.Wb;
      w = cw ? w - c.Wb : w; h = cw ? h + 1 : h;
      const bool chh = h >= c.Hb;
      h = chh ? h - c.Hb : h; d = chh ? d + 1 : d;
    } else {
      const long v = m0 + iv;
      w = (int)(v % c.Wb); h = (int)((v / c.Wb) % c.Hb); d = (int)(v / ((long)c.Wb * c.Hb));
    }
  };
  int wb, hb, db;
  row_dhw(li, db, hb, wb);
  const size_t in_sample = (size_t)c.Di * c.Hi * c.Wi * S.cs;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * in_sample * XE), 0,
                                                                (int)(in_sample * XE), 0x00020000);
  const int aoff = vok ? ((((db * c.siD) * c.Hi + hb * c.siH) * c.Wi + wb * c.siW) * S.cs + 8 * lhalf) * XE : (int)0x80000000;
  const bool aff = S.scale != nullptr;
  const float slope = S.slope;
  const bool lrelu_ok = (slope >= 0.f) && (slope <= 1.f);
  // the producer's per-(sample, channel) scale / shift once per workgroup in LDS: fetching them per element through the vector
  // memory path cost 16 extra VMEM instructions per chunk and lane (measured: 30 -> 47 head 3.1 ms)
  __shared__ __attribute__((aligned(16))) float ssc[PW_MAXC], ssh[PW_MAXC];
  if (aff) {
    for (int i = tid; i < P.nchunks * PW_CK; i += 256) {
      ssc[i] = i < S.C ? S.scale[(size_t)nb * S.C + i] : 0.f;
      ssh[i] = i < S.C ? S.shift[(size_t)nb * S.C + i] : 0.f;
    }
    __syncthreads();
  }

  auto load_a = [&](int ch, float (&x)[8]) {
    const int o = aoff + ch * (PW_CK * XE);
    if constexpr (XS != MT_F32) {
      pw_load8<XS>(ra, o, x);
    } else if constexpr (VEC == 4) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, o + g * 16, 0, 0));
        x[4 * g] = t[0]; x[4 * g + 1] = t[1]; x[4 * g + 2] = t[2]; x[4 * g + 3] = t[3];
      }
    } else if constexpr (VEC == 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ra, o + g * 8, 0, 0));
        x[2 * g] = t.x; x[2 * g + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) x[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, o + g * 4, 0, 0));
    }
  };
  // channels beyond Cin inside the last chunk may hold neighbouring data: zero them (and apply the lazy activation)
  auto finish_a = [&](int ch, float (&x)[8]) {
    const int cb = ch * PW_CK + 8 * lhalf;
    if (aff) {
      const f32x4 sc0 = *(const f32x4*)(ssc + cb), sc1 = *(const f32x4*)(ssc + cb + 4);
      const f32x4 sh0 = *(const f32x4*)(ssh + cb), sh1 = *(const f32x4*)(ssh + cb + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(x[e], e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
        x[e] = lrelu_ok ? fmaxf(t, t * slope) : mt_lrelu(t, slope);
      }
    }
    if (cb + 8 > c.Cin || !vok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (vok && cb + e < c.Cin) ? x[e] : 0.f;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  // blockIdx.z: which NT of the P.ntaps taps this workgroup computes (8 taps as two workgroups of 4: half the accumulator
  // registers, twice the resident workgroups — their store phases overlap the others' multiplications)
  const int tap0 = (int)blockIdx.z * NT;

  float xa[8], xn[8];
  load_a(0, xa);
  for (int ch = 0; ch < P.nchunks; ++ch) {
    if (ch + 1 < P.nchunks) load_a(ch + 1, xn);
    finish_a(ch, xa);
    if constexpr (M16) {
      typedef _Float16 pw_f16x8 __attribute__((ext_vector_type(8)));
      uint4 af;
      af.x = mt_pk16<MT_F16>(xa[0], xa[1]); af.y = mt_pk16<MT_F16>(xa[2], xa[3]); af.z = mt_pk16<MT_F16>(xa[4], xa[5]); af.w = mt_pk16<MT_F16>(xa[6], xa[7]);
      const unsigned* wq16 = (const unsigned*)c.wpack + ((size_t)(ntile * P.nchunks + ch) * P.ntaps + tap0) * 256 + lane * 4;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint4 b = *(const uint4*)(wq16 + t * 256);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pw_f16x8, af), __builtin_bit_cast(pw_f16x8, b), acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) xa[e] = xn[e];
      continue;
    }
    const float* wq = c.wpack + ((size_t)(ntile * P.nchunks + ch) * P.ntaps + tap0) * 512 + lane * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 b0, b1;
      if (PW_ABL & 2) { b0 = f32x4{xa[0], xa[1], xa[2], xa[3]}; b1 = f32x4{xa[4], xa[5], xa[6], xa[7]}; }
      else { b0 = *(const f32x4*)(wq + t * 512); b1 = *(const f32x4*)(wq + t * 512 + 256); }
      if (PW_ABL & 4) { acc[t][0] += xa[t & 7] * b0[0] + b1[1]; continue; }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], b0[e], acc[t], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[4 + e], b1[e], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) xa[e] = xn[e];
  }
  if (PW_ABL & 1) {
    float sacc = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) sacc += acc[t][j];
    if (sacc == 1234.5678f) c.out[0] = sacc;
    return;
  }

  // ---- epilogue
  const int co = ntile * 32 + li;
  const bool covalid = co < c.Cout;
  const float bias = (c.bias != nullptr && covalid) ? c.bias[co] : 0.f;
  const int Ho = c.Hb * c.soH, Wo = c.Wb * c.soW, Do = c.Db * c.soD;
  const size_t out_sample = (size_t)Do * Ho * Wo * c.ocs;
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)c.out + (size_t)nb * out_sample * OE), 0,
                                                                (int)(out_sample * OE), 0x00020000);
  if constexpr (NT >= 4 && VEC == 4) {
    // Wide epilogue of the transposed convolutions (soW == 2, the wave's 32 base voxels in one row, <= 32 even output channels):
    // the two kw taps of a (kd, kh) pair are 64 CONSECUTIVE output voxels.  The dword stores of the plain epilogue (one per
    // accumulator element: 128 store instructions per wave, each two 120-byte runs) were 300 of the 460 us of the 60 -> 30
    // launch (PW_ABL=1); here the pair goes through a wave-private LDS tile [64 voxels][32] and leaves as 16-byte stores.
    if (P.wide) {
      __shared__ __attribute__((aligned(16))) float wst[4][64 * 32];
      float* sg = wst[wave];
      int wb0, hb0, db0;
      row_dhw(0, db0, hb0, wb0);                             // the row of the wave's first voxel (the whole wave is in it)
      const bool wok = m0 < P.Vb;
      const int pv = lane >> 3, pc = (lane & 7) * 4;         // piece (lane & 7) of output voxel pv + 8 k
#pragma unroll
      for (int pr = 0; pr < NT / 2; ++pr) {                  // (kd, kh) pairs: taps 2 pr (kw = 0) and 2 pr + 1 (kw = 1)
        const int prg = pr + tap0 / 2;
        const int th = prg % c.soH, tdd = prg / c.soH;
#pragma unroll
        for (int tw = 0; tw < 2; ++tw)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf;
            sg[(2 * iv + tw) * 32 + li] = acc[2 * pr + tw][j] + bias;
          }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (wok && P.wide == 2) {
          // dense output (channel stride == Cout): the pair's 64 voxels are ONE run of 64 * Cout floats
          const int rowbase = ((((db0 * c.soD + tdd) * Ho + hb0 * c.soH + th) * Wo + wb0 * 2) * c.Cout) * OE;
          const int n2 = 32 * c.Cout;                        // channel pairs: 8-byte units (fp32) / dwords (16-bit)
          for (int u = lane; u < n2; u += 64) {
            const int e = 2 * u, ov = e / c.Cout, cc = e - ov * c.Cout;      // (Cout even: a unit never straddles two voxels)
            const float2 h = *(const float2*)(sg + ov * 32 + cc);
            if constexpr (OS == MT_F32) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, h), ro, rowbase + u * 8, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(mt_pk16<OS>(h.x, h.y), ro, rowbase + u * 4, 0, 0);
          }
        } else if (wok) {
          const int rowbase = ((((db0 * c.soD + tdd) * Ho + hb0 * c.soH + th) * Wo + wb0 * 2) * c.ocs) * OE;      // bytes
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int ov = pv + 8 * k;
            const f32x4 v = *(const f32x4*)(sg + ov * 32 + pc);
            const int o = rowbase + (ov * c.ocs + pc) * OE;
            if constexpr (OS == MT_F32) {
              if (pc + 4 <= c.Cout)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ro, o, 0, 0);
              else if (pc + 2 <= c.Cout) {
                float2 h; h.x = v[0]; h.y = v[1];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, h), ro, o, 0, 0);
              }
            } else {
              if (pc + 4 <= c.Cout) {
                uint2 h; h.x = mt_pk16<OS>(v[0], v[1]); h.y = mt_pk16<OS>(v[2], v[3]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, h), ro, o, 0, 0);
              } else if (pc + 2 <= c.Cout) {
                __builtin_amdgcn_raw_buffer_store_b32(mt_pk16<OS>(v[0], v[1]), ro, o, 0, 0);
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();                     // the tile is rewritten by the next pair
      }
      return;
    }
  }
  float s1 = 0.f, s2 = 0.f;
  if constexpr (OS != MT_F32) {
    // 16-bit destination: channel-pair dwords (pw_pair_exchange) — even lanes store accumulator row j, odd lanes row j + 1
    const bool odd = li & 1;
    const int coe = co & ~1;
    const bool pvalid = coe + 1 < c.Cout;                 // (Cout, ocs even: mt_pointwise_io_supported)
    int pbase[8];
#pragma unroll
    for (int jp = 0; jp < 8; ++jp) {
      const int j = 2 * jp;
      const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf + (odd ? 1 : 0);
      const long v = m0 + iv;
      int w2, h2, d2;
      row_dhw(iv, d2, h2, w2);
      const bool ok = pvalid && v < P.Vb;
      pbase[jp] = ok ? ((((d2 * c.soD) * Ho + h2 * c.soH) * Wo + w2 * c.soW) * c.ocs + coe) * 2 : (int)0x80000000;
    }
    float q1[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tg = tap0 + t;
      const int tw = tg % c.soW, th = (tg / c.soW) % c.soH, tdd = tg / (c.soW * c.soH);
      const int toff = ((tdd * Ho + th) * Wo + tw) * c.ocs * 2;
#pragma unroll
      for (int jp = 0; jp < 8; ++jp) {
        float a, b;
        pw_pair_exchange(acc[t][2 * jp] + bias, acc[t][2 * jp + 1] + bias, odd, a, b);
        if (c.accumulate) { const unsigned pv = __builtin_amdgcn_raw_buffer_load_b32(ro, pbase[jp], toff, 0); a += mt_lo16<OS>(pv); b += mt_hi16<OS>(pv); }
        const unsigned pk = mt_pk16<OS>(a, b);
        __builtin_amdgcn_raw_buffer_store_b32(pk, ro, pbase[jp], toff, 0);
        if (c.stats_part != nullptr && pbase[jp] >= 0) {
          const float ar = mt_lo16<OS>(pk), br = mt_hi16<OS>(pk);
          q1[0] += ar; q2[0] = fmaf(ar, ar, q2[0]); q1[1] += br; q2[1] = fmaf(br, br, q2[1]);
        }
      }
    }
    s1 = pw_pair_combine(q1[0], q1[1], odd);
    s2 = pw_pair_combine(q2[0], q2[1], odd);
  } else {
  int obase[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf;
    const long v = m0 + iv;
    int w2, h2, d2;
    row_dhw(iv, d2, h2, w2);
    const bool ok = covalid && v < P.Vb;
    obase[j] = ok ? ((((d2 * c.soD) * Ho + h2 * c.soH) * Wo + w2 * c.soW) * c.ocs + co) * 4 : (int)0x80000000;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tg = tap0 + t;
    const int tw = tg % c.soW, th = (tg / c.soW) % c.soH, tdd = tg / (c.soW * c.soH);
    const int toff = ((tdd * Ho + th) * Wo + tw) * c.ocs * 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float val = acc[t][j] + bias;
      if (c.accumulate) val += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ro, obase[j], toff, 0));
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, obase[j], toff, 0);
      if (c.stats_part != nullptr && obase[j] >= 0) { s1 += val; s2 = fmaf(val, val, s2); }
    }
  }
  }
  if (c.stats_part != nullptr) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lhalf == 0) { red[(wave * 32 + li) * 2] = s1; red[(wave * 32 + li) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 32 && (ntile * 32 + tid) < c.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * 32 + tid) * 2]; t2 += red[(w * 32 + tid) * 2 + 1]; }
      float* sp = c.stats_part + ((size_t)((size_t)nb * P.nsb + sb) * c.Cout + ntile * 32 + tid) * 2;
      sp[0] = t1; sp[1] = t2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Inference: segmentation head + nonlinearity + un-flip + accumulation in ONE kernel (neural_network.py:502-591 does
// pred = nonlin(net(flip(x))); result += flip^-1(pred) / num_results per mirror combination).  The 1x1x1 head is computed with the
// MFMA operand roles SWAPPED (weights as the row operand), so a lane owns a VOXEL and its registers are output channels: the
// channel-major accumulator acc[C][D][H][W] is then written with 32 consecutive voxels per channel row — 128-byte aligned runs —
// instead of 47-channel NDHWC rows of 188 bytes, and the logits never exist in HBM (the separate head wrote 2.7 GB per batch of
// eight tiles at 0.9 TB/s and flip_accumulate read them back).  The register contents of both operands are exactly those of
// pw_fast_kernel; only their order in the MFMA changes.
struct HeadAccParams {
  mt_pointwise_t c;
  int nchunks, nsb, sample, fD, fH, fW, nonlin, first;
  long V;
  float weight;
  float* acc;
};
// XS: storage type of the source (fp32, or 16-bit activations of the mixed mode: a lane's 8 channels are ONE 16-byte load)
template <int VEC, int XS = MT_F32>
__global__ __launch_bounds__(256) void head_flip_accumulate_kernel(const HeadAccParams P) {
  constexpr int XE = mt_ebytes<XS>();
  const mt_pointwise_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;
  const int sb = mt_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = P.sample;
  const long m0 = (long)sb * 128 + wave * 32;
  const mt_src_t& S = c.src;
  const long bv = m0 + li;
  const bool vok = bv < P.V;
  const size_t in_sample = (size_t)P.V * S.cs;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * in_sample * XE), 0, (int)(in_sample * XE), 0x00020000);
  const int aoff = vok ? (int)((bv * S.cs + 8 * lhalf) * XE) : (int)0x80000000;
  const bool aff = S.scale != nullptr;
  const float slope = S.slope;
  const bool lrelu_ok = (slope >= 0.f) && (slope <= 1.f);
  __shared__ __attribute__((aligned(16))) float ssc[PW_MAXC], ssh[PW_MAXC];
  if (aff) {
    for (int i = tid; i < P.nchunks * PW_CK; i += 256) {
      ssc[i] = i < S.C ? S.scale[(size_t)nb * S.C + i] : 0.f;
      ssh[i] = i < S.C ? S.shift[(size_t)nb * S.C + i] : 0.f;
    }
    __syncthreads();
  }
  auto load_a = [&](int ch, float (&x)[8]) {        // (16-bit: raw dwords in x[0..3], widened in finish_a — a conversion here would wait for the prefetch)
    const int o = aoff + ch * (PW_CK * XE);
    if constexpr (XS != MT_F32) {
      const uint4 t = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, o, 0, 0));
      x[0] = __builtin_bit_cast(float, t.x); x[1] = __builtin_bit_cast(float, t.y); x[2] = __builtin_bit_cast(float, t.z); x[3] = __builtin_bit_cast(float, t.w);
    } else if constexpr (VEC == 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ra, o + g * 8, 0, 0));
        x[2 * g] = t.x; x[2 * g + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) x[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, o + g * 4, 0, 0));
    }
  };
  auto finish_a = [&](int ch, float (&x)[8]) {
    const int cb = ch * PW_CK + 8 * lhalf;
    if constexpr (XS != MT_F32) {
      const unsigned r0 = __builtin_bit_cast(unsigned, x[0]), r1 = __builtin_bit_cast(unsigned, x[1]), r2 = __builtin_bit_cast(unsigned, x[2]), r3 = __builtin_bit_cast(unsigned, x[3]);
      x[0] = mt_lo16<XS>(r0); x[1] = mt_hi16<XS>(r0); x[2] = mt_lo16<XS>(r1); x[3] = mt_hi16<XS>(r1);
      x[4] = mt_lo16<XS>(r2); x[5] = mt_hi16<XS>(r2); x[6] = mt_lo16<XS>(r3); x[7] = mt_hi16<XS>(r3);
    }
    if (aff) {
      const f32x4 sc0 = *(const f32x4*)(ssc + cb), sc1 = *(const f32x4*)(ssc + cb + 4);
      const f32x4 sh0 = *(const f32x4*)(ssh + cb), sh1 = *(const f32x4*)(ssh + cb + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(x[e], e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
        x[e] = lrelu_ok ? fmaxf(t, t * slope) : mt_lrelu(t, slope);
      }
    }
    if (cb + 8 > c.Cin || !vok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (vok && cb + e < c.Cin) ? x[e] : 0.f;
    }
  };
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  const bool two = c.Cout > 32;                      // block-uniform
  float xa[8], xn[8];
  load_a(0, xa);
  for (int ch = 0; ch < P.nchunks; ++ch) {
    if (ch + 1 < P.nchunks) load_a(ch + 1, xn);
    finish_a(ch, xa);
    const float* wq = c.wpack + (size_t)ch * 512 + lane * 4;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      if (n == 1 && !two) break;
      const float* wn = wq + (size_t)n * P.nchunks * 512;
      const f32x4 b0 = *(const f32x4*)(wn), b1 = *(const f32x4*)(wn + 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[e], xa[e], acc[n], 0, 0, 0);      // rows = channels
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[e], xa[4 + e], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < (XS != MT_F32 ? 4 : 8); ++e) xa[e] = xn[e];
  }
  // ---- epilogue: this lane's voxel, channels n*32 + (j&3) + 8*(j>>2) + 4*lhalf
  if (c.bias != nullptr) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int cj = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf;
        acc[n][j] += cj < c.Cout ? c.bias[cj] : 0.f;
      }
  }
  if (P.nonlin == 2) {                                // softmax over ALL channels of the voxel: own registers + the partner lane's
    float mx = -3.0e38f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf < c.Cout) mx = fmaxf(mx, acc[n][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float se = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool cv = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf < c.Cout;
        acc[n][j] = cv ? expf(acc[n][j] - mx) : 0.f;
        se += acc[n][j];
      }
    se += __shfl_xor(se, 32, 64);
    const float inv = 1.f / se;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[n][j] *= inv;
  } else if (P.nonlin == 1) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[n][j] = 1.f / (1.f + expf(-acc[n][j]));
  }
  if (!vok) return;
  const int w = (int)(bv % c.Wb), h = (int)((bv / c.Wb) % c.Hb), d = (int)(bv / ((long)c.Wb * c.Hb));
  const long dv = ((long)(P.fD ? c.Db - 1 - d : d) * c.Hb + (P.fH ? c.Hb - 1 - h : h)) * c.Wb + (P.fW ? c.Wb - 1 - w : w);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (n == 1 && !two) break;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cj = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf;
      if (cj < c.Cout) {
        float* a = P.acc + (size_t)cj * P.V + dv;
        const float v = acc[n][j] * P.weight;
        *a = P.first ? v : *a + v;
      }
    }
  }
}
extern "C" int mt_head_flip_accumulate(const mt_pointwise_t* p, int sample, int flipD, int flipH, int flipW, int nonlin, float weight,
                                       float* acc, int first, mt_stream_t stream) {
  MT_REQUIRE(p != nullptr && acc != nullptr, "head_flip_accumulate: null pointers");
  MT_REQUIRE(mt_dtype_ok(p->src.dtype), "head_flip_accumulate: bad source storage type %d", p->src.dtype);
  MT_REQUIRE(p->src.dtype == MT_F32 || (!(p->src.cs & 1) && !(((uintptr_t)p->src.ptr) & 3)), "head_flip_accumulate: a 16-bit source needs an even channel stride");
  MT_REQUIRE(p->siD == 1 && p->siH == 1 && p->siW == 1 && p->soD == 1 && p->soH == 1 && p->soW == 1 && p->Db == p->Di && p->Hb == p->Hi &&
             p->Wb == p->Wi, "head_flip_accumulate: 1x1x1 stride-1 head only");
  MT_REQUIRE(p->Cout >= 1 && p->Cout <= 64 && p->src.C == p->Cin && sample >= 0 && sample < p->N, "head_flip_accumulate: needs 1..64 output channels");
  MT_REQUIRE(nonlin >= 0 && nonlin <= 2, "head_flip_accumulate: nonlin must be 0 (none), 1 (sigmoid) or 2 (softmax)");
  HeadAccParams P;
  P.c = *p; P.nchunks = mt_cdiv(p->Cin, PW_CK); P.V = (long)p->Db * p->Hb * p->Wb; P.nsb = mt_cdiv(P.V, 128);
  MT_REQUIRE(P.nchunks * PW_CK <= PW_MAXC && (double)P.V * p->src.cs * 4.0 < 2147483648.0, "head_flip_accumulate: sample too large");
  P.sample = sample; P.fD = flipD; P.fH = flipH; P.fW = flipW; P.nonlin = nonlin; P.first = first; P.weight = weight; P.acc = acc;
  const mt_src_t& S = p->src;
  const bool v2 = (S.cs % 2) == 0 && (((uintptr_t)S.ptr) & 7) == 0;
  if (S.dtype == MT_F16) hipLaunchKernelGGL((head_flip_accumulate_kernel<2, MT_F16>), dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else if (S.dtype == MT_BF16) hipLaunchKernelGGL((head_flip_accumulate_kernel<2, MT_BF16>), dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else if (v2) hipLaunchKernelGGL(head_flip_accumulate_kernel<2>, dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else    hipLaunchKernelGGL(head_flip_accumulate_kernel<1>, dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("head_flip_accumulate");
  return MT_OK;
}

// ------------------------------------------------------------------------------------------------
// All mirror combinations of a tile in ONE kernel, straight into the volume aggregate: for output voxel v of the tile
//   agg[c][tile + v] += gauss[v] * weight * sum_k nonlin(head(features_k[flip_k(v)]))_c ,   nb[tile + v] += gauss[v]
// (neural_network.py:531-586 result += flip^-1(pred) / num_results per combination, then :384-394 result *= gaussian and the
// overlap-add).  The sum over the samples k stays in registers, so the per-tile accumulator and its 2 x 333 MB read-modify-write
// per mirror combination (plus the separate tile_accumulate pass) disappear: 8 x 212 MB of features in, one update of the
// aggregate out.
struct HeadMirParams {
  mt_pointwise_t c;
  int nchunks, nsb, sample0, nsamples, nonlin;
  int flips[8];                 // bit 0: D, bit 1: H, bit 2: W
  long V;
  float weight;
  const float* gauss;           // [D][H][W] or NULL (= 1)
  float* agg; float* nb;        // agg[C][aX][aY][aZ], nb[aX][aY][aZ] (nb may be NULL)
  long aX, aY, aZ; int x0, y0, z0;
};
template <int VEC, int XS = MT_F32>
__global__ __launch_bounds__(256) void head_mirror_accumulate_kernel(const HeadMirParams P) {
  constexpr int XE = mt_ebytes<XS>();
  const mt_pointwise_t& c = P.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;
  const int sb = mt_xcd_remap(blockIdx.x, gridDim.x);
  const long bv = (long)sb * 128 + wave * 32 + li;
  const bool vok = bv < P.V;
  const int w = (int)(bv % c.Wb), h = (int)((bv / c.Wb) % c.Hb), d = (int)(bv / ((long)c.Wb * c.Hb));
  const mt_src_t& S = c.src;
  const size_t in_sample = (size_t)P.V * S.cs;
  const bool aff = S.scale != nullptr;
  const float slope = S.slope;
  const bool lrelu_ok = (slope >= 0.f) && (slope <= 1.f);
  const bool two = c.Cout > 32;                      // block-uniform
  __shared__ __attribute__((aligned(16))) float ssc[PW_MAXC], ssh[PW_MAXC];
  f32x16 sum[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) sum[n][j] = 0.f;
  float bias[2][16];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cj = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf;
      bias[n][j] = (c.bias != nullptr && cj < c.Cout) ? c.bias[cj] : 0.f;
    }

  for (int k = 0; k < P.nsamples; ++k) {
    const int nb_ = P.sample0 + k, f = P.flips[k];
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb_ * in_sample * XE), 0, (int)(in_sample * XE), 0x00020000);
    const long sv = ((long)((f & 1) ? c.Db - 1 - d : d) * c.Hb + ((f & 2) ? c.Hb - 1 - h : h)) * c.Wb + ((f & 4) ? c.Wb - 1 - w : w);
    const int aoff = vok ? (int)((sv * S.cs + 8 * lhalf) * XE) : (int)0x80000000;
    if (aff) {
      __syncthreads();
      for (int i = tid; i < P.nchunks * PW_CK; i += 256) {
        ssc[i] = i < S.C ? S.scale[(size_t)nb_ * S.C + i] : 0.f;
        ssh[i] = i < S.C ? S.shift[(size_t)nb_ * S.C + i] : 0.f;
      }
      __syncthreads();
    }
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[n][j] = bias[n][j];
    for (int ch = 0; ch < P.nchunks; ++ch) {
      float x[8];
      const int o = aoff + ch * (PW_CK * XE);
      if constexpr (XS != MT_F32) {
        const uint4 t = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, o, 0, 0));
        x[0] = mt_lo16<XS>(t.x); x[1] = mt_hi16<XS>(t.x); x[2] = mt_lo16<XS>(t.y); x[3] = mt_hi16<XS>(t.y);
        x[4] = mt_lo16<XS>(t.z); x[5] = mt_hi16<XS>(t.z); x[6] = mt_lo16<XS>(t.w); x[7] = mt_hi16<XS>(t.w);
      } else if constexpr (VEC == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ra, o + g * 8, 0, 0));
          x[2 * g] = t.x; x[2 * g + 1] = t.y;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) x[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, o + g * 4, 0, 0));
      }
      const int cb = ch * PW_CK + 8 * lhalf;
      if (aff) {
        const f32x4 sc0 = *(const f32x4*)(ssc + cb), sc1 = *(const f32x4*)(ssc + cb + 4);
        const f32x4 sh0 = *(const f32x4*)(ssh + cb), sh1 = *(const f32x4*)(ssh + cb + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = fmaf(x[e], e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
          x[e] = lrelu_ok ? fmaxf(t, t * slope) : mt_lrelu(t, slope);
        }
      }
      if (cb + 8 > c.Cin || !vok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (vok && cb + e < c.Cin) ? x[e] : 0.f;
      }
      const float* wq = c.wpack + (size_t)ch * 512 + lane * 4;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n == 1 && !two) break;
        const float* wn = wq + (size_t)n * P.nchunks * 512;
        const f32x4 b0 = *(const f32x4*)(wn), b1 = *(const f32x4*)(wn + 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[e], x[e], acc[n], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[e], x[4 + e], acc[n], 0, 0, 0);
      }
    }
    if (P.nonlin == 2) {
      float mx = -3.0e38f;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf < c.Cout) mx = fmaxf(mx, acc[n][j]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const bool cv = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf < c.Cout;
          acc[n][j] = cv ? expf(acc[n][j] - mx) : 0.f;
          se += acc[n][j];
        }
      se += __shfl_xor(se, 32, 64);
      const float inv = 1.f / se;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum[n][j] += acc[n][j] * inv;
    } else {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum[n][j] += P.nonlin == 1 ? 1.f / (1.f + expf(-acc[n][j])) : acc[n][j];
    }
  }
  if (!vok) return;
  const float g = P.gauss ? P.gauss[bv] : 1.f;
  const float wg = P.weight * g;
  const size_t av = ((size_t)(P.x0 + d) * P.aY + (P.y0 + h)) * P.aZ + (P.z0 + w);
  const size_t AV = (size_t)P.aX * P.aY * P.aZ;
  if (P.nb != nullptr && lhalf == 0) P.nb[av] += g;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (n == 1 && !two) break;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cj = n * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhalf;
      if (cj < c.Cout) P.agg[(size_t)cj * AV + av] += sum[n][j] * wg;
    }
  }
}
extern "C" int mt_head_mirror_accumulate(const mt_pointwise_t* p, int sample0, int nsamples, const int32_t* flips, int nonlin, float weight,
                                         const float* gauss, float* agg, float* nb, long aX, long aY, long aZ, int x0, int y0, int z0,
                                         mt_stream_t stream) {
  MT_REQUIRE(p == nullptr || mt_dtype_ok(p->src.dtype), "head_mirror_accumulate: bad source storage type");
  MT_REQUIRE(p == nullptr || p->src.dtype == MT_F32 || (!(p->src.cs & 1) && !(((uintptr_t)p->src.ptr) & 3)), "head_mirror_accumulate: a 16-bit source needs an even channel stride");
  MT_REQUIRE(p != nullptr && agg != nullptr && flips != nullptr, "head_mirror_accumulate: null pointers");
  MT_REQUIRE(p->siD == 1 && p->siH == 1 && p->siW == 1 && p->soD == 1 && p->soH == 1 && p->soW == 1 && p->Db == p->Di && p->Hb == p->Hi &&
             p->Wb == p->Wi, "head_mirror_accumulate: 1x1x1 stride-1 head only");
  MT_REQUIRE(p->Cout >= 1 && p->Cout <= 64 && p->src.C == p->Cin, "head_mirror_accumulate: needs 1..64 output channels");
  MT_REQUIRE(nsamples >= 1 && nsamples <= 8 && sample0 >= 0 && sample0 + nsamples <= p->N, "head_mirror_accumulate: bad sample range");
  MT_REQUIRE(nonlin >= 0 && nonlin <= 2, "head_mirror_accumulate: nonlin must be 0, 1 (sigmoid) or 2 (softmax)");
  MT_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + p->Db <= aX && y0 + p->Hb <= aY && z0 + p->Wb <= aZ, "head_mirror_accumulate: tile outside the aggregate");
  HeadMirParams P;
  P.c = *p; P.nchunks = mt_cdiv(p->Cin, PW_CK); P.V = (long)p->Db * p->Hb * p->Wb; P.nsb = mt_cdiv(P.V, 128);
  MT_REQUIRE(P.nchunks * PW_CK <= PW_MAXC && (double)P.V * p->src.cs * 4.0 < 2147483648.0, "head_mirror_accumulate: sample too large");
  P.sample0 = sample0; P.nsamples = nsamples; P.nonlin = nonlin; P.weight = weight; P.gauss = gauss; P.agg = agg; P.nb = nb;
  for (int k = 0; k < 8; ++k) P.flips[k] = k < nsamples ? flips[k] : 0;
  P.aX = aX; P.aY = aY; P.aZ = aZ; P.x0 = x0; P.y0 = y0; P.z0 = z0;
  const mt_src_t& S = p->src;
  const bool v2 = (S.cs % 2) == 0 && (((uintptr_t)S.ptr) & 7) == 0;
  if (S.dtype == MT_F16) hipLaunchKernelGGL((head_mirror_accumulate_kernel<2, MT_F16>), dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else if (S.dtype == MT_BF16) hipLaunchKernelGGL((head_mirror_accumulate_kernel<2, MT_BF16>), dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else if (v2) hipLaunchKernelGGL(head_mirror_accumulate_kernel<2>, dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  else    hipLaunchKernelGGL(head_mirror_accumulate_kernel<1>, dim3((unsigned)P.nsb), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("head_mirror_accumulate");
  return MT_OK;
}

extern "C" int mt_pointwise_stats_blocks(const mt_pointwise_t* p) {
  if (p == nullptr) return -1;
  return mt_cdiv((long)p->Db * p->Hb * p->Wb, 128);
}

// ---- 1x1x1 head with 33..64 output channels into a DENSE [V][Cout] tensor (the 47 MultiTalent logits, generic_UNet.py:349-351) ----
// pw_fast_kernel serves it with two workgroups per 128 voxels (channels 0-31 and 32-46): the input is read twice and every voxel's
// 188-byte row is written as a 128-byte and a 60-byte piece, neither aligned to anything (measured 1.8 TB/s).  Here one workgroup
// computes both channel tiles from one read of the input, transposes its 128 x Cout block through LDS and writes it as ONE linear run
// of 16-byte stores (128 * 47 * 4 = 24 064 contiguous bytes).  Requires V % 32 == 0, unit strides, no accumulation / statistics.
#define PWH_MAXCO 64
template <int XS = MT_F32, bool M16 = false>
__global__ __launch_bounds__(256) void pw_head_kernel(const PwKParams P) {
  constexpr int XE = mt_ebytes<XS>();
  const mt_pointwise_t& c = P.c;
  __shared__ __attribute__((aligned(16))) float ssc[PW_MAXC], ssh[PW_MAXC];
  __shared__ __attribute__((aligned(16))) float stage[4][32 * PWH_MAXCO];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;
  const int bx = mt_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bx / P.nsb, sb = bx % P.nsb;
  const long m0 = (long)sb * 128 + wave * 32;
  const mt_src_t& S = c.src;
  const bool wok = m0 < P.Vb;                              // (whole waves: V % 32 == 0)
  const size_t in_sample = (size_t)P.Vb * S.cs;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * in_sample * XE), 0, (int)(in_sample * XE), 0x00020000);
  const int aoff = wok ? (int)(((m0 + li) * S.cs + 8 * lhalf) * XE) : (int)0x80000000;
  const bool aff = S.scale != nullptr;
  const float slope = S.slope;
  if (aff) {
    for (int i = tid; i < P.nchunks * PW_CK; i += 256) {
      ssc[i] = i < S.C ? S.scale[(size_t)nb * S.C + i] : 0.f;
      ssh[i] = i < S.C ? S.shift[(size_t)nb * S.C + i] : 0.f;
    }
    __syncthreads();
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  float xa[8], xn[8];
  auto load_a = [&](int ch, float (&x)[8]) { pw_load8<XS>(ra, aoff + ch * (PW_CK * XE), x); };
  load_a(0, xa);
  for (int ch = 0; ch < P.nchunks; ++ch) {
    if (ch + 1 < P.nchunks) load_a(ch + 1, xn);
    const int cb = ch * PW_CK + 8 * lhalf;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = xa[e];
    if (aff) {
      const f32x4 sc0 = *(const f32x4*)(ssc + cb), sc1 = *(const f32x4*)(ssc + cb + 4);
      const f32x4 sh0 = *(const f32x4*)(ssh + cb), sh1 = *(const f32x4*)(ssh + cb + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(x[e], e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
        x[e] = mt_lrelu(t, slope);
      }
    }
    if (cb + 8 > c.Cin) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (cb + e < c.Cin) ? x[e] : 0.f;
    }
    if constexpr (M16) {
      typedef _Float16 pw_f16x8 __attribute__((ext_vector_type(8)));
      uint4 af;
      af.x = mt_pk16<MT_F16>(x[0], x[1]); af.y = mt_pk16<MT_F16>(x[2], x[3]); af.z = mt_pk16<MT_F16>(x[4], x[5]); af.w = mt_pk16<MT_F16>(x[6], x[7]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4 b = *(const uint4*)((const unsigned*)c.wpack + (size_t)(t * P.nchunks + ch) * 256 + lane * 4);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pw_f16x8, af), __builtin_bit_cast(pw_f16x8, b), acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) xa[e] = xn[e];
      continue;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* wq = c.wpack + (size_t)(t * P.nchunks + ch) * 512 + lane * 4;
      const f32x4 b0 = *(const f32x4*)(wq);
      const f32x4 b1 = *(const f32x4*)(wq + 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], b0[e], acc[t], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[4 + e], b1[e], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) xa[e] = xn[e];
  }
  // ---- epilogue: [32 voxels][Cout] of this wave through LDS, then a linear run of 16-byte stores
  float* sg = stage[wave];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int co = t * 32 + li;
    const float bias = (c.bias != nullptr && co < c.Cout) ? c.bias[co] : 0.f;
    if (co < c.Cout) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf;
        sg[iv * c.Cout + co] = acc[t][j] + bias;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): this wave's own LDS writes (no other wave touches stage[wave])
  __builtin_amdgcn_wave_barrier();
  if (wok) {
    const size_t out_sample = (size_t)P.Vb * c.Cout;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(c.out + (size_t)nb * out_sample), 0, (int)(out_sample * 4), 0x00020000);
    const int n4 = 8 * c.Cout;                             // float4 per wave block (32 * Cout / 4)
    const int obase = (int)(m0 * c.Cout * 4);
    for (int i = lane; i < n4; i += 64) {
      const f32x4 v = *(const f32x4*)(sg + 4 * i);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ro, obase + i * 16, 0, 0);
    }
  }
}

// ---- narrow heads: 1x1x1 convolution to <= 4 output channels (the heads of the few-class single-dataset trainers: 2 logits for
// Task009) from a dense CIN-channel tensor.  The MFMA forms spend a 32-wide output tile on 2 channels and fetch their A fragments as
// 32-byte pieces of 120-byte rows (2.7 TB/s); here a thread owns a voxel: its row is one contiguous run (consecutive lanes = consecutive
// rows: the wave reads 7.5 KiB linearly), the lazy InstanceNorm+LeakyReLU and CIN x Cout multiply-adds run on the vector ALU
// (60 FMAs per 120 bytes: far below the bandwidth bound), W / scale / shift come from LDS as broadcast reads.
// a dense CIN-channel row (byte offset o) as 16-byte loads + tail; CIN % 2 == 0
template <int CIN, int XS>
__device__ __forceinline__ void pw_load_row(__amdgpu_buffer_rsrc_t r, int o, float (&x)[CIN + 2]) {
  if constexpr (XS == MT_F32) {
#pragma unroll
    for (int q = 0; q < CIN / 4; ++q) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + q * 16, 0, 0));
      x[4 * q] = t[0]; x[4 * q + 1] = t[1]; x[4 * q + 2] = t[2]; x[4 * q + 3] = t[3];
    }
    if constexpr ((CIN % 4) != 0) {
      const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, o + (CIN / 4) * 16, 0, 0));
      x[CIN - 2] = t.x; x[CIN - 1] = t.y;
    }
  } else {
    constexpr int ND = CIN / 2;            // dwords of the row
    unsigned d[ND + 3];
#pragma unroll
    for (int q = 0; q < ND / 4; ++q) {
      const uint4 t = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, o + q * 16, 0, 0));
      d[4 * q] = t.x; d[4 * q + 1] = t.y; d[4 * q + 2] = t.z; d[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int q = (ND / 4) * 4; q < ND; ++q) d[q] = __builtin_amdgcn_raw_buffer_load_b32(r, o + q * 4, 0, 0);
#pragma unroll
    for (int q = 0; q < ND; ++q) { x[2 * q] = mt_lo16<XS>(d[q]); x[2 * q + 1] = mt_hi16<XS>(d[q]); }
  }
}
template <int CIN, int OS>
__device__ __forceinline__ void pw_store_row(__amdgpu_buffer_rsrc_t r, int o, const float (&x)[CIN + 2]) {
  if constexpr (OS == MT_F32) {
#pragma unroll
    for (int q = 0; q < CIN / 4; ++q) {
      f32x4 t; t[0] = x[4 * q]; t[1] = x[4 * q + 1]; t[2] = x[4 * q + 2]; t[3] = x[4 * q + 3];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, t), r, o + q * 16, 0, 0);
    }
    if constexpr ((CIN % 4) != 0) {
      float2 t; t.x = x[CIN - 2]; t.y = x[CIN - 1];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, t), r, o + (CIN / 4) * 16, 0, 0);
    }
  } else {
    constexpr int ND = CIN / 2;
    unsigned d[ND + 3];
#pragma unroll
    for (int q = 0; q < ND; ++q) d[q] = mt_pk16<OS>(x[2 * q], x[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < ND / 4; ++q) {
      uint4 t; t.x = d[4 * q]; t.y = d[4 * q + 1]; t.z = d[4 * q + 2]; t.w = d[4 * q + 3];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, t), r, o + q * 16, 0, 0);
    }
#pragma unroll
    for (int q = (ND / 4) * 4; q < ND; ++q) __builtin_amdgcn_raw_buffer_store_b32(d[q], r, o + q * 4, 0, 0);
  }
}
template <int CIN, int XS = MT_F32>
__global__ __launch_bounds__(256) void pw_narrow_kernel(const PwKParams P) {
  constexpr int XE = mt_ebytes<XS>();
  const mt_pointwise_t& c = P.c;
  __shared__ __attribute__((aligned(16))) float sw[4][CIN + 2], ssc[CIN + 2], ssh[CIN + 2];
  const int tid = threadIdx.x;
  const int nb = blockIdx.y;
  const mt_src_t& S = c.src;
  const bool aff = S.scale != nullptr;
  for (int i = tid; i < 4 * CIN; i += 256) {
    const int co = i / CIN, ci = i - co * CIN;
    // packed layout 1 (ck = 16): channel ci = 16 ch + 8 half + 4 q + e of output co sits at [ch][q][lane = 32 half + co][e]
    const int ch = ci >> 4, cl = ci & 15, half = cl >> 3, kp = cl & 7;
    sw[co][ci] = co < c.Cout ? c.wpack[(size_t)ch * 512 + ((kp >> 2) * 64 + half * 32 + co) * 4 + (kp & 3)] : 0.f;
  }
  for (int i = tid; i < CIN; i += 256) {
    ssc[i] = aff ? S.scale[(size_t)nb * S.C + i] : 1.f;
    ssh[i] = aff ? S.shift[(size_t)nb * S.C + i] : 0.f;
  }
  __syncthreads();
  const float slope = aff ? S.slope : 1.f;
  float bias[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) bias[co] = (c.bias != nullptr && co < c.Cout) ? c.bias[co] : 0.f;
  const size_t in_sample = (size_t)P.Vb * CIN;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * in_sample * XE), 0, (int)(in_sample * XE), 0x00020000);
  float* outp = c.out + (size_t)nb * P.Vb * c.Cout;
  for (long v = (long)blockIdx.x * 256 + tid; v < P.Vb; v += (long)gridDim.x * 256) {
    float x[CIN + 2];
    pw_load_row<CIN, XS>(ra, (int)(v * (CIN * XE)), x);
    float y[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float t = fmaf(x[ci], ssc[ci], ssh[ci]);
      const float a = fmaxf(t, t * slope);                   // LeakyReLU for slope in [0, 1]
#pragma unroll
      for (int co = 0; co < 4; ++co) y[co] = fmaf(a, sw[co][ci], y[co]);
    }
    float* q = outp + v * c.Cout;
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < c.Cout) q[co] = y[co];
  }
}
static bool pw_narrow_ok(const mt_pointwise_t* p, const PwKParams& P) {
  return P.ntaps == 1 && p->soD * p->soH * p->soW == 1 && p->siD == 1 && p->siH == 1 && p->siW == 1 && p->Cout <= 4 && (p->Cin == 30 || p->Cin == 32) && p->src.cs == p->Cin &&
         p->ocs == p->Cout && !p->accumulate && p->stats_part == nullptr && p->Di == p->Db && p->Hi == p->Hb && p->Wi == p->Wb &&
         (p->src.slope >= 0.f && p->src.slope <= 1.f) && ((((uintptr_t)p->out) & 15) == 0);
}
// the dense-output head form (pw_head_kernel): 1x1x1, unit strides, 33..64 output channels written densely, whole waves of voxels
static bool pw_head_ok(const mt_pointwise_t* p, const PwKParams& P) {
  return P.ntaps == 1 && p->soD * p->soH * p->soW == 1 && p->siD == 1 && p->siH == 1 && p->siW == 1 && p->Cout > 32 && p->Cout <= PWH_MAXCO && p->ocs == p->Cout &&
         !p->accumulate && p->stats_part == nullptr && (P.Vb % 32) == 0 && p->Di == p->Db && p->Hi == p->Hb && p->Wi == p->Wb &&
         ((((uintptr_t)p->out) & 15) == 0) && ((P.Vb * p->Cout) % 4 == 0);
}
static constexpr int pw_use_head_env() { return 1; }
// the part of the launch plan the kernel choice depends on (shared by mt_pointwise_fwd and mt_pointwise_pack_layout)
static void pw_plan(const mt_pointwise_t* p, PwKParams& P, bool& narrow, bool& head) {
  P.c = *p;
  P.ntaps = p->scatter ? 1 : p->soD * p->soH * p->soW;
  P.nchunks = mt_cdiv(p->Cin, PW_CK);
  P.Vb = (long)p->Db * p->Hb * p->Wb;
  P.nsb = mt_cdiv(P.Vb, 128);
  P.wide = 0;
  narrow = pw_use_head_env() && p->odtype == MT_F32 && pw_narrow_ok(p, P);
  head = !narrow && pw_use_head_env() && p->odtype == MT_F32 && pw_head_ok(p, P);
}
// Storage types mt_pointwise_fwd takes natively (mt_pointwise_t.src.dtype -> odtype): fp32 -> fp32 always; a 16-bit source needs an even
// channel stride and a dword-aligned base (its 8-channel groups are 16-byte loads on dword boundaries), a 16-bit destination even Cout /
// channel stride and a dword-aligned base (channel-pair dwords).  Combinations: fp16 -> fp16 | fp32 (forward over activations),
// bf16 -> bf16 | fp32 and fp32 -> bf16 (backward-data over gradients).
extern "C" int mt_pointwise_io_supported(const mt_pointwise_t* p) {
  if (p == nullptr) return 0;
  const int xs = p->src.dtype, os = p->odtype;
  if (!mt_dtype_ok(xs) || !mt_dtype_ok(os)) return 0;
  if (xs == MT_F32 && os == MT_F32) return 1;
  if (mt_is16(xs) && ((p->src.cs & 1) || (((uintptr_t)p->src.ptr) & 3))) return 0;
  if (mt_is16(os) && ((p->Cout & 1) || (p->ocs & 1) || (((uintptr_t)p->out) & 3))) return 0;
  if (xs == MT_F16) return os == MT_F16 || os == MT_F32;
  if (xs == MT_BF16) return os == MT_BF16 || os == MT_F32;
  return os == MT_BF16;                       // fp32 source (the loss gradient) into a bf16 gradient
}
// 16-bit products (mt_pointwise_t.mma == 1): fp16 source; pw_head_kernel (fp32 logits) or pw_fast_kernel writing fp16 (transposed convs)
static bool pw_m16(const mt_pointwise_t* p, const PwKParams& P, bool narrow, bool head) {
  constexpr int use = 1;
  if (!use || p->mma != 1 || p->src.dtype != MT_F16 || narrow || p->scatter) return false;
  if ((p->src.cs & 1) || (((uintptr_t)p->src.ptr) & 3)) return false;
  return head ? true : (p->odtype == MT_F16 && P.ntaps >= 2);
}
static void pw_plan(const mt_pointwise_t* p, PwKParams& P, bool& narrow, bool& head);
extern "C" int mt_pointwise_pack_layout(const mt_pointwise_t* p) {
  if (p == nullptr || !mt_pointwise_io_supported(p)) return 1;
  PwKParams P; bool narrow, head;
  pw_plan(p, P, narrow, head);
  return pw_m16(p, P, narrow, head) ? 4 : 1;
}
extern "C" int mt_pointwise_fwd(const mt_pointwise_t* p, mt_stream_t stream) {
  MT_REQUIRE(p != nullptr, "pointwise: null params");
  MT_REQUIRE(mt_pointwise_io_supported(p), "pointwise: storage types (src %d, out %d) not taken (ask mt_pointwise_io_supported, convert with mt_cast)", p->src.dtype, p->odtype);
  const int xs = p->src.dtype, os = p->odtype;
  const bool any16 = xs != MT_F32 || os != MT_F32;
  MT_REQUIRE(p->N > 0 && p->Db > 0 && p->Hb > 0 && p->Wb > 0 && p->Cin > 0 && p->Cout > 0, "pointwise: empty problem");
  MT_REQUIRE(p->siD >= 1 && p->siD <= 2 && p->siH >= 1 && p->siH <= 2 && p->siW >= 1 && p->siW <= 2, "pointwise: input stride must be 1 or 2");
  MT_REQUIRE(p->soD >= 1 && p->soD <= 2 && p->soH >= 1 && p->soH <= 2 && p->soW >= 1 && p->soW <= 2, "pointwise: output stride must be 1 or 2");
  MT_REQUIRE((p->Db - 1) * p->siD < p->Di && (p->Hb - 1) * p->siH < p->Hi && (p->Wb - 1) * p->siW < p->Wi, "pointwise: base grid exceeds stored input");
  MT_REQUIRE(p->src.C == p->Cin, "pointwise: src.C != Cin");
  MT_REQUIRE(p->src.ptr && p->wpack && p->out, "pointwise: null pointers");
  PwKParams P;
  P.c = *p;
  P.ntaps = p->scatter ? 1 : p->soD * p->soH * p->soW;       // scatter: only tap (0,0,0) exists (one packed tap)
  P.nchunks = mt_cdiv(p->Cin, PW_CK);
  P.Vb = (long)p->Db * p->Hb * p->Wb;
  P.nsb = mt_cdiv(P.Vb, 128);
  MT_REQUIRE((double)p->Di * p->Hi * p->Wi * p->src.cs * 4.0 < 2147483648.0 &&
             (double)P.Vb * (p->soD * p->soH * p->soW) * p->ocs * 4.0 < 2147483648.0, "pointwise: sample larger than 2 GiB");
  MT_REQUIRE(P.ntaps == 1 || P.ntaps == 2 || P.ntaps == 4 || P.ntaps == 8, "pointwise: unsupported tap count %d", P.ntaps);
  MT_REQUIRE(P.nchunks * PW_CK <= PW_MAXC, "pointwise: Cin = %d exceeds %d", p->Cin, PW_MAXC);
  {
    constexpr int use_wide = 1;
    const bool shape_ok = use_wide && P.ntaps >= 4 && p->soW == 2 && p->soH == 2 && (p->Wb % 32) == 0 && p->Cout <= 32 && (p->Cout % 2) == 0 &&
                          !p->accumulate && p->stats_part == nullptr && p->siD == 1 && p->siH == 1 && p->siW == 1;
    P.wide = 0;
    if (shape_ok && p->ocs == p->Cout && ((((uintptr_t)p->out) & 7) == 0)) P.wide = 2;                       // dense output: linear 8-byte (fp32) / 4-byte (16-bit) stores
    else if (shape_ok && (p->ocs % 4) == 0 && ((((uintptr_t)p->out) & 15) == 0)) P.wide = 1;                 // concat slot: 16 / 8-byte pieces per voxel
  }
  const mt_src_t& S = p->src;
  // 16-byte loads whatever the alignment: a raw buffer_load_dwordx4 only needs dword alignment and range-checks per dword
  // (tools/ubench/oob128.hip); the 47-channel gradient of the heads (188-byte rows) went through eight scalar loads per chunk before
  constexpr int force_vec = 0;
  int vec = 4;
  if (!any16 && (force_vec == 1 || force_vec == 2 || force_vec == 4)) vec = force_vec;
  if (vec == 2 && !((S.cs % 2) == 0 && (((uintptr_t)S.ptr) & 7) == 0)) vec = 1;
  hipStream_t st = (hipStream_t)stream;
  {
    constexpr int use_head = 1;
    if (use_head && os == MT_F32 && pw_narrow_ok(p, P)) {
      long blocks = (P.Vb + 255) / 256; if (blocks > 4096) blocks = 4096;
      const dim3 g2((unsigned)blocks, (unsigned)p->N);
#define PW_NARROW(CIN_) do { if (xs == MT_F16) hipLaunchKernelGGL((pw_narrow_kernel<CIN_, MT_F16>), g2, dim3(256), 0, st, P);           \
                             else if (xs == MT_BF16) hipLaunchKernelGGL((pw_narrow_kernel<CIN_, MT_BF16>), g2, dim3(256), 0, st, P);    \
                             else hipLaunchKernelGGL((pw_narrow_kernel<CIN_, MT_F32>), g2, dim3(256), 0, st, P); } while (0)
      if (p->Cin == 30) PW_NARROW(30); else PW_NARROW(32);
#undef PW_NARROW
      MT_CHECK_LAUNCH("pointwise_narrow");
      return MT_OK;
    }
    if (use_head && os == MT_F32 && pw_head_ok(p, P)) {
      const dim3 g1((unsigned)(P.nsb * p->N));
      if (pw_m16(p, P, false, true)) hipLaunchKernelGGL((pw_head_kernel<MT_F16, true>), g1, dim3(256), 0, st, P);
      else if (xs == MT_F16) hipLaunchKernelGGL((pw_head_kernel<MT_F16>), g1, dim3(256), 0, st, P);
      else if (xs == MT_BF16) hipLaunchKernelGGL((pw_head_kernel<MT_BF16>), g1, dim3(256), 0, st, P);
      else hipLaunchKernelGGL((pw_head_kernel<MT_F32>), g1, dim3(256), 0, st, P);
      MT_CHECK_LAUNCH("pointwise_head");
      return MT_OK;
    }
  }
  dim3 grid((unsigned)(P.nsb * p->N), (unsigned)mt_cdiv(p->Cout, 32), 1);
#define PW_LAUNCH_T(NT, XS_, OS_) hipLaunchKernelGGL((pw_fast_kernel<NT, 4, XS_, OS_>), grid, dim3(256), 0, st, P)
#define PW_LAUNCH(NT)                                                                              \
  do {                                                                                             \
    if (NT >= 2 && pw_m16(p, P, false, false)) hipLaunchKernelGGL((pw_fast_kernel<(NT >= 2 ? NT : 2), 4, MT_F16, MT_F16, true>), grid, dim3(256), 0, st, P); \
    else if (xs == MT_F16 && os == MT_F16) PW_LAUNCH_T(NT, MT_F16, MT_F16);                        \
    else if (xs == MT_F16) PW_LAUNCH_T(NT, MT_F16, MT_F32);                                        \
    else if (xs == MT_BF16 && os == MT_BF16) PW_LAUNCH_T(NT, MT_BF16, MT_BF16);                    \
    else if (xs == MT_BF16) PW_LAUNCH_T(NT, MT_BF16, MT_F32);                                      \
    else if (os == MT_BF16) PW_LAUNCH_T(NT, MT_F32, MT_BF16);                                      \
    else if (vec == 4) hipLaunchKernelGGL((pw_fast_kernel<NT, 4>), grid, dim3(256), 0, st, P);      \
    else if (vec == 2) hipLaunchKernelGGL((pw_fast_kernel<NT, 2>), grid, dim3(256), 0, st, P);      \
    else hipLaunchKernelGGL((pw_fast_kernel<NT, 1>), grid, dim3(256), 0, st, P);                    \
  } while (0)
  switch (P.ntaps) {
    case 1: PW_LAUNCH(1); break;
    case 2: PW_LAUNCH(2); break;
    case 4: PW_LAUNCH(4); break;
    default: {
      constexpr int split8 = 1;
      if (split8 && p->stats_part == nullptr) { grid.z = 2; PW_LAUNCH(4); }      // two workgroups of four taps (see pw_fast_kernel)
      else PW_LAUNCH(8);
      break;
    }
  }
#undef PW_LAUNCH
#undef PW_LAUNCH_T
  MT_CHECK_LAUNCH("pointwise");
  return MT_OK;
}


// ================================================================================================
// Backward of a 1x1x1 segmentation head in ONE pass over (x, dY) — generic_UNet.py:349-351 / generic_modular_UNet.py:244,251:
//   dX[n,v,ci] (+)= sum_co dY[n,v,co] W[co,ci]                 (gradient w.r.t. the ACTIVATED head input a = lrelu(x*scale+shift))
//   dW[co,ci]  (+)= sum_{n,v} a[n,v,ci] dY[n,v,co],   dbias[co] (+)= sum_{n,v} dY[n,v,co]
// The separate kernels (pointwise backward-data + tiled backward-weight) moved 2 x |x| + 3 x |dY| + |dX| at 1.2-2.5 TB/s: with 47
// output channels at full resolution the heads cost 3.2 ms of a 74 ms Task100 step.  Here a wave walks over 32-voxel tiles:
//   * dX tile = dY tile (A operand: the lane's voxel row, 8 contiguous channels per 16-chunk, 16-byte loads) x W^T (packed B
//     fragments, held in registers for the whole kernel);
//   * dW += a^T dY with the VOXELS as the contraction index: both operands are then "lane = channel" rows of one voxel (coalesced
//     120 / 188-byte reads that hit the lines the dX part just fetched), two voxels per MFMA; row 31 of the last input-channel tile,
//     when free, carries 1.0 so that the same MFMAs produce dbias;
//   * every wave keeps its dW partial (NCI x 2 accumulator tiles) in registers and writes it once; head_bwd_reduce_kernel sums the
//     partials in fp64 in a fixed order (deterministic, no atomics).
struct HeadBwdParams {
  mt_src_t x; const float* dy; int dycs; int N; long V; int Cin, Cout;
  const float* wpack; float* dx; int dxcs; int accumulate_dx;
  float* part; int nwaves; long ntiles;
};
// XS: storage type of the head's input x (fp32 | fp16 | bf16); OS: of dX (fp32 | bf16).  dY (the loss gradient) is fp32.
// ST (round 5, dense tensors: x.cs == dxcs == Cin, dycs == Cout): the tile's dY block [32][Cout] and x block [32][Cin] are ONE
// contiguous run each — staged into a wave-private LDS image as coalesced 16-byte pieces, operands read from there, dX leaves the
// same way.  Without it a tile costs 70 vector-memory instructions (6 row-per-lane b128 loads, 48 two-voxel gathers of 128-376
// bytes, 16 two-row stores) for 56 MFMAs: 10.5 k cycles per tile against 3.6 k of matrix time.
template <int NCI, int XS = MT_F32, int OS = MT_F32, bool ST = false>
__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadBwdParams P) {
  constexpr int XE = mt_ebytes<XS>(), OE = mt_ebytes<OS>();
  constexpr int HB_WF = 2048 + 1024 * NCI;                   // floats of a wave's image: dY [32][<= 64], x / dX [32][<= 32 NCI]
  __shared__ __attribute__((aligned(16))) float hb_img[ST ? 4 * HB_WF : 4];
  static_assert(!ST || 4 * HB_WF * sizeof(float) <= 65536, "head_bwd_kernel: the four wave images must fit the 64 KiB of static LDS (NCI <= 2)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const idy = hb_img + (ST ? wave * HB_WF : 0);
  float* const ixf = idy + (ST ? 2048 : 0);                  // x block (storage type XS), afterwards the dX tile in fp32
  const int li = lane & 31, lhalf = lane >> 5;
  const int gw = blockIdx.x * 4 + wave;
  const mt_src_t& S = P.x;
  const bool aff = S.scale != nullptr;
  const float slope = aff ? S.slope : 1.f;
  const int nchunks = (P.Cout + 15) / 16;                          // K chunks of the dX product (K = Cout <= 64)
  // packed W^T fragments: [ci tile][co chunk][2][64 lanes][4] — constant for the whole kernel
  f32x4 wb[NCI][4][2];
#pragma unroll
  for (int t = 0; t < NCI; ++t)
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float* wq = P.wpack + (size_t)(t * nchunks + ch) * 512 + lane * 4;
      wb[t][ch][0] = ch < nchunks ? *(const f32x4*)(wq) : f32x4{0.f, 0.f, 0.f, 0.f};
      wb[t][ch][1] = ch < nchunks ? *(const f32x4*)(wq + 256) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  f32x16 aw[NCI][2];
#pragma unroll
  for (int t = 0; t < NCI; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) aw[t][n][j] = 0.f;
  // the ones row (dbias): the last row of the last ci tile, when no input channel lives there
  const bool ones_free = (P.Cin % 32) != 0;
  const long tiles_per_sample = (P.V + 31) / 32;
  int cur_nb = -1;
  float xsc[NCI], xsh[NCI];
#pragma unroll
  for (int t = 0; t < NCI; ++t) { xsc[t] = 0.f; xsh[t] = 0.f; }
  for (long tile = gw; tile < P.ntiles; tile += P.nwaves) {
    const int nb = (int)(tile / tiles_per_sample);
    const long m0 = (tile - (long)nb * tiles_per_sample) * 32;
    if (nb != cur_nb) {                                            // (wave-uniform) per-(sample, channel) lazy-activation constants
      cur_nb = nb;
#pragma unroll
      for (int t = 0; t < NCI; ++t) {
        const int ci = t * 32 + li;
        const bool cv = ci < P.Cin;
        xsc[t] = cv ? (aff ? S.scale[(size_t)nb * S.C + ci] : 1.f) : 0.f;
        xsh[t] = (cv && aff) ? S.shift[(size_t)nb * S.C + ci] : 0.f;
      }
    }
    const size_t ysample = (size_t)P.V * P.dycs, xsample = (size_t)P.V * S.cs, dsample = (size_t)P.V * P.dxcs;
    __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dy + (size_t)nb * ysample), 0, (int)(ysample * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * xsample * XE), 0, (int)(xsample * XE), 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)P.dx + (size_t)nb * dsample * OE), 0, (int)(dsample * OE), 0x00020000);
    // ---- dX = dY W^T: A operand = this lane's voxel row of dY, channels 16 ch + 8 lhalf .. +7
    const long bv = m0 + li;
    const bool vok = bv < P.V;
    const int yoff = vok ? (int)((bv * P.dycs + 8 * lhalf) * 4) : (int)0x80000000;
    if constexpr (ST) {
      const int ybytes = 32 * P.Cout * 4, xbytes = 32 * P.Cin * XE;
      const int ybase = (int)(m0 * P.Cout * 4), xbase = (int)(m0 * P.Cin * XE);
      uint4 py[8], px[4 * NCI];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int o = (k * 64 + lane) * 16;
        py[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ry, o < ybytes ? ybase + o : (int)0x80000000, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < 4 * NCI; ++k) {
        const int o = (k * 64 + lane) * 16;
        px[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, o < xbytes ? xbase + o : (int)0x80000000, 0, 0));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();                         // the previous tile's dX pieces have left the image
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int o = (k * 64 + lane) * 16;
        if (o < ybytes) *(uint4*)((char*)idy + o) = py[k];
      }
#pragma unroll
      for (int k = 0; k < 4 * NCI; ++k) {
        const int o = (k * 64 + lane) * 16;
        if (o < xbytes) *(uint4*)((char*)ixf + o) = px[k];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    f32x16 ax[NCI];
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) ax[t][j] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch < nchunks) {
        float xa[8];
        if constexpr (ST) {
#pragma unroll
          for (int e = 0; e < 8; ++e) xa[e] = idy[li * P.Cout + ch * 16 + 8 * lhalf + e];      // (rows past the sample were staged as zeros)
        } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, yoff + (ch * 16 + g * 4) * 4, 0, 0));
          xa[4 * g] = v[0]; xa[4 * g + 1] = v[1]; xa[4 * g + 2] = v[2]; xa[4 * g + 3] = v[3];
        }
        }
        const int cb = ch * 16 + 8 * lhalf;                        // a row's tail runs into the next voxel's first channels: zero them
        if (cb + 8 > P.Cout) {
#pragma unroll
          for (int e = 0; e < 8; ++e) xa[e] = (cb + e < P.Cout) ? xa[e] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NCI; ++t) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ax[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], wb[t][ch][0][e], ax[t], 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) ax[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[4 + e], wb[t][ch][1][e], ax[t], 0, 0, 0);
        }
      }
    }
    // ---- dW += a^T dY, two voxels per MFMA (k = lhalf): operands are channel rows of voxel m0 + 2 s + lhalf
#pragma unroll 4
    for (int s2 = 0; s2 < 16; ++s2) {
      const long v = m0 + 2 * s2 + lhalf;
      const bool in = v < P.V;
      const int vo = in ? (int)(v * 4) : (int)0x80000000;          // (scaled below; bit 31 survives the multiplications as a mask)
      float av[NCI], bvv[2];
#pragma unroll
      for (int t = 0; t < NCI; ++t) {
        const int ci = t * 32 + li;
        const int o = (in && ci < P.Cin) ? (int)((v * S.cs + ci) * XE) : (int)0x80000000;
        float raw;
        if constexpr (ST) {
          const int ei = (2 * s2 + lhalf) * P.Cin + ci;
          if constexpr (XS == MT_F32) raw = (ci < P.Cin) ? ixf[ei] : 0.f;
          else raw = (ci < P.Cin) ? mt_from16<XS>(((const unsigned short*)ixf)[ei]) : 0.f;
        } else if constexpr (XS == MT_F32) raw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        else raw = mt_from16<XS>(__builtin_amdgcn_raw_buffer_load_b16(rx, o, 0, 0));
        const float tt = fmaf(raw, xsc[t], xsh[t]);
        av[t] = in ? fmaxf(tt, tt * slope) : 0.f;
        if (ones_free && t == NCI - 1 && li == 31) av[t] = in ? 1.f : 0.f;
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int co = n * 32 + li;
        const int o = (in && co < P.Cout) ? (int)((v * P.dycs + co) * 4) : (int)0x80000000;
        if constexpr (ST) bvv[n] = (in && co < P.Cout) ? idy[(2 * s2 + lhalf) * P.Cout + co] : 0.f;
        else bvv[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, o, 0, 0));
      }
      (void)vo;
#pragma unroll
      for (int t = 0; t < NCI; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) aw[t][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bvv[n], aw[t][n], 0, 0, 0);
    }
    // ---- store dX (C layout: lane = input channel column, registers = voxel rows)
    if constexpr (ST) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();                         // every lane has read its x operands: the region becomes the fp32 dX tile
#pragma unroll
      for (int t = 0; t < NCI; ++t) {
        const int ci = t * 32 + li;
        if (ci < P.Cin) {
#pragma unroll
          for (int j = 0; j < 16; ++j) ixf[((j & 3) + 8 * (j >> 2) + 4 * lhalf) * P.Cin + ci] = ax[t][j];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const int dbytes = 32 * P.Cin * OE, dbase = (int)(m0 * P.Cin * OE);
#pragma unroll
      for (int k = 0; k < 4 * NCI; ++k) {
        const int o = (k * 64 + lane) * 16;                    // byte offset of this lane's 16-byte piece inside the tile's dX block
        if (o < dbytes) {
          if constexpr (OS == MT_F32) {
            f32x4 v = *(const f32x4*)((const char*)ixf + o);
            if (P.accumulate_dx) {
              const f32x4 old = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, dbase + o, 0, 0));
              v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rd, dbase + o, 0, 0);
          } else {
            const f32x4 v0 = *(const f32x4*)((const char*)ixf + 2 * o), v1 = *(const f32x4*)((const char*)ixf + 2 * o + 16);
            float e[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (P.accumulate_dx) {
              const uint4 old = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rd, dbase + o, 0, 0));
              e[0] += mt_lo16<OS>(old.x); e[1] += mt_hi16<OS>(old.x); e[2] += mt_lo16<OS>(old.y); e[3] += mt_hi16<OS>(old.y);
              e[4] += mt_lo16<OS>(old.z); e[5] += mt_hi16<OS>(old.z); e[6] += mt_lo16<OS>(old.w); e[7] += mt_hi16<OS>(old.w);
            }
            uint4 q; q.x = mt_pk16<OS>(e[0], e[1]); q.y = mt_pk16<OS>(e[2], e[3]); q.z = mt_pk16<OS>(e[4], e[5]); q.w = mt_pk16<OS>(e[6], e[7]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rd, dbase + o, 0, 0);
          }
        }
      }
    } else
#pragma unroll
    for (int t = 0; t < NCI; ++t) {
      const int ci = t * 32 + li;
      if constexpr (OS != MT_F32) {          // channel-pair dwords (pw_pair_exchange): even lanes row j, odd lanes row j + 1; Cin, dxcs even
        const bool odd = li & 1;
        const int cie = ci & ~1;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const long v = m0 + (j & 3) + 8 * (j >> 2) + 4 * lhalf + (odd ? 1 : 0);
          const int o = (cie + 1 < P.Cin && v < P.V) ? (int)((v * P.dxcs + cie) * 2) : (int)0x80000000;
          float a, b;
          pw_pair_exchange(ax[t][j], ax[t][j + 1], odd, a, b);
          if (P.accumulate_dx) { const unsigned pv = __builtin_amdgcn_raw_buffer_load_b32(rd, o, 0, 0); a += mt_lo16<OS>(pv); b += mt_hi16<OS>(pv); }
          __builtin_amdgcn_raw_buffer_store_b32(mt_pk16<OS>(a, b), rd, o, 0, 0);
        }
      } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long v = m0 + (j & 3) + 8 * (j >> 2) + 4 * lhalf;
        const int o = (ci < P.Cin && v < P.V) ? (int)((v * P.dxcs + ci) * 4) : (int)0x80000000;
        float val = ax[t][j];
        if (P.accumulate_dx) val += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, o, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rd, o, 0, 0);
      }
      }
    }
  }
  // ---- this wave's dW partial: [wave][t][n][j 16][lane 64]
  float* pp = P.part + (size_t)gw * (NCI * 2 * 1024);
#pragma unroll
  for (int t = 0; t < NCI; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) pp[((t * 2 + n) * 16 + j) * 64 + lane] = aw[t][n][j];
}

#define HB_SLICES 32
struct HeadBwdReduce { const float* part; double* tmp; int nwaves, nci, Cin, Cout; float* dw; long s_ci, s_co; float* dbias; int accumulate; };
// stage A: tmp[slice][e] = sum over the slice's partials (fp64, fixed order) — 32 x fewer dependent loads per thread than one pass
__global__ __launch_bounds__(256) void head_bwd_reduce_a_kernel(const HeadBwdReduce R) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = R.nci * 2 * 1024;
  if (e >= per) return;
  const int sl = blockIdx.y;
  const int w0 = (int)((long)R.nwaves * sl / HB_SLICES), w1 = (int)((long)R.nwaves * (sl + 1) / HB_SLICES);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;                   // four fixed chains: loads in flight, order independent of timing
  int w = w0;
  for (; w + 4 <= w1; w += 4) {
    s0 += (double)R.part[(size_t)w * per + e];
    s1 += (double)R.part[(size_t)(w + 1) * per + e];
    s2 += (double)R.part[(size_t)(w + 2) * per + e];
    s3 += (double)R.part[(size_t)(w + 3) * per + e];
  }
  for (; w < w1; ++w) s0 += (double)R.part[(size_t)w * per + e];
  R.tmp[(size_t)sl * per + e] = (s0 + s1) + (s2 + s3);
}
// stage B: element e = ((t*2 + n)*16 + j)*64 + lane of the accumulator layout -> dW[co][ci] / dbias[co]
__global__ __launch_bounds__(256) void head_bwd_reduce_b_kernel(const HeadBwdReduce R) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = R.nci * 2 * 1024;
  if (e >= per) return;
  const int lane = e & 63, j = (e >> 6) & 15, tn = e >> 10, n = tn & 1, t = tn >> 1;
  const int row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), col = lane & 31;
  const int ci = t * 32 + row, co = n * 32 + col;
  const bool is_bias = (R.Cin % 32) != 0 && t == R.nci - 1 && row == 31;
  if (co >= R.Cout || (ci >= R.Cin && !is_bias)) return;
  double a = 0.0;
  for (int sl = 0; sl < HB_SLICES; ++sl) a += R.tmp[(size_t)sl * per + e];
  const float s = (float)a;
  if (is_bias) { if (R.dbias != nullptr) R.dbias[co] = R.accumulate ? R.dbias[co] + s : s; return; }
  float* o = R.dw + (long)ci * R.s_ci + (long)co * R.s_co;
  *o = R.accumulate ? *o + s : s;
}

// ---- narrow heads (Cout <= 4, 30 / 32 dense input channels; see pw_narrow_kernel): the same three results from one streaming pass
// with a thread per voxel — dX[v][ci] (+)= sum_co dY[v][co] W[co][ci]; per-thread partial sums of dW[co][ci] = sum_v act(x)[v][ci]
// dY[v][co] and dbias[co] = sum_v dY[v][co] in registers over the thread's voxels, reduced over the wave by DPP shuffles and over
// the workgroup through LDS in a fixed order; one partial row per workgroup, summed in fp64 by head_narrow_reduce_kernel.
#define HN_BLOCKS 1024
// a lane's row of CIN channels in the wave's LDS image (rows are only 8- / 4-byte aligned: 120 / 60 bytes apart)
template <int CIN, int ST>
__device__ __forceinline__ void hn_row_from_lds(const char* row, float (&x)[CIN + 2]) {
  if constexpr (ST == MT_F32) {
#pragma unroll
    for (int q = 0; q < CIN / 2; ++q) { const float2 t = *(const float2*)(row + q * 8); x[2 * q] = t.x; x[2 * q + 1] = t.y; }
  } else {
#pragma unroll
    for (int q = 0; q < CIN / 2; ++q) { const unsigned d = *(const unsigned*)(row + q * 4); x[2 * q] = mt_lo16<ST>(d); x[2 * q + 1] = mt_hi16<ST>(d); }
  }
}
template <int CIN, int ST>
__device__ __forceinline__ void hn_row_to_lds(char* row, const float (&x)[CIN + 2]) {
  if constexpr (ST == MT_F32) {
#pragma unroll
    for (int q = 0; q < CIN / 2; ++q) { float2 t; t.x = x[2 * q]; t.y = x[2 * q + 1]; *(float2*)(row + q * 8) = t; }
  } else {
#pragma unroll
    for (int q = 0; q < CIN / 2; ++q) *(unsigned*)(row + q * 4) = mt_pk16<ST>(x[2 * q], x[2 * q + 1]);
  }
}
template <int CIN, int NCO, int XS = MT_F32, int OS = MT_F32>
__global__ __launch_bounds__(256) void head_bwd_narrow_kernel(const HeadBwdParams P) {
  static_assert(CIN % 2 == 0, "channel pairs");
  __shared__ __attribute__((aligned(16))) float hn_img[4 * 64 * CIN];          // one [64 voxels][CIN] image per wave (fp32-sized)
  constexpr int XE = mt_ebytes<XS>(), OE = mt_ebytes<OS>();
  constexpr int NP = NCO * CIN + NCO;                       // partial sums per thread: dW rows, then dbias
  __shared__ __attribute__((aligned(16))) float sw[NCO][CIN + 2], ssc[CIN + 2], ssh[CIN + 2];
  __shared__ float red[4][NP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const mt_src_t& S = P.x;
  const bool aff = S.scale != nullptr;
  const float slope = aff ? S.slope : 1.f;
  for (int i = tid; i < NCO * CIN; i += 256) {
    const int co = i / CIN, ci = i - co * CIN;
    sw[co][ci] = co < P.Cout ? P.wpack[ci * 4 + co] : 0.f;   // packed W^T (K = Cout in one chunk, co < 4: [lane = ci][e = co])
  }
  float part[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) part[k] = 0.f;
  const long per_sample_blocks = HN_BLOCKS / P.N > 0 ? HN_BLOCKS / P.N : 1;
  const int nb = (int)(blockIdx.x / per_sample_blocks);      // a workgroup stays inside one sample (its scale / shift)
  if (nb < P.N) {
    const long b = blockIdx.x - (long)nb * per_sample_blocks;
    for (int i = tid; i < CIN; i += 256) {
      ssc[i] = aff ? S.scale[(size_t)nb * S.C + i] : 1.f;
      ssh[i] = aff ? S.shift[(size_t)nb * S.C + i] : 0.f;
    }
    __syncthreads();
    const size_t xs = (size_t)P.V * CIN;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (size_t)nb * xs * XE), 0, (int)(xs * XE), 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)P.dx + (size_t)nb * xs * OE), 0, (int)(xs * OE), 0x00020000);
    const float* dyp = P.dy + (size_t)nb * P.V * P.dycs;
    // Rows through LDS (round 5).  A thread owns a voxel and needs its CIN channels as registers, but a row-per-lane access is 64
    // pieces at 120-byte (60-byte) strides per instruction: 3.3 TB/s.  The rows of a wave's 64 consecutive voxels are ONE contiguous
    // block of 64 * CIN elements, so the wave moves it as 16-byte pieces (lane l: pieces l, l + 64, ...) through a wave-private LDS
    // image of the same linear layout, and every lane reads / writes its own row there.
    constexpr int ROWX = CIN * XE, ROWO = CIN * OE;            // bytes per row
    constexpr int NPX = 64 * ROWX / 16, NPO = 64 * ROWO / 16;  // 16-byte pieces of a wave's block
    char* const img = (char*)hn_img + wave * (64 * CIN * 4);
    for (long v0 = b * 256 + wave * 64; v0 < P.V; v0 += per_sample_blocks * 256) {
      const long v = v0 + lane;
      const bool vok = v < P.V;
      float x[CIN + 2], old[CIN + 2], dy[NCO];
      {
        uint4 pc[(NPX + 63) / 64];
#pragma unroll
        for (int k = 0; k < (NPX + 63) / 64; ++k) {
          const int p = k * 64 + lane;
          pc[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, p < NPX ? (int)(v0 * ROWX) + p * 16 : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < (NPX + 63) / 64; ++k) {
          const int p = k * 64 + lane;
          if (p < NPX) *(uint4*)(img + p * 16) = pc[k];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        hn_row_from_lds<CIN, XS>(img + lane * ROWX, x);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
      if (P.accumulate_dx) {
        uint4 pc[(NPO + 63) / 64];
#pragma unroll
        for (int k = 0; k < (NPO + 63) / 64; ++k) {
          const int p = k * 64 + lane;
          pc[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, p < NPO ? (int)(v0 * ROWO) + p * 16 : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < (NPO + 63) / 64; ++k) {
          const int p = k * 64 + lane;
          if (p < NPO) *(uint4*)(img + p * 16) = pc[k];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        hn_row_from_lds<CIN, OS>(img + lane * ROWO, old);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int co = 0; co < NCO; ++co) dy[co] = (vok && co < P.Cout) ? dyp[v * P.dycs + co] : 0.f;      // (a lane past the sample adds nothing)
      float dx[CIN + 2];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float t = fmaf(x[ci], ssc[ci], ssh[ci]);
        const float a = fmaxf(t, t * slope);
        float g = P.accumulate_dx ? old[ci] : 0.f;
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          g = fmaf(dy[co], sw[co][ci], g);
          part[co * CIN + ci] = fmaf(a, dy[co], part[co * CIN + ci]);
        }
        dx[ci] = g;
      }
#pragma unroll
      for (int co = 0; co < NCO; ++co) part[NCO * CIN + co] += dy[co];
      hn_row_to_lds<CIN, OS>(img + lane * ROWO, dx);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < (NPO + 63) / 64; ++k) {
        const int p = k * 64 + lane;
        if (p < NPO) {                                         // (pieces past the sample's last row: beyond num_records, dropped)
          const uint4 q = *(const uint4*)(img + p * 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rx, (int)(v0 * ROWO) + p * 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
  // wave reduction (fixed butterfly), then the four waves through LDS in wave order
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    float s = part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  for (int k = tid; k < NP; k += 256) P.part[(size_t)blockIdx.x * NP + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}
// dW[co][ci] / dbias[co] (+)= sum over the workgroups' partial rows, in block order, fp64
__global__ __launch_bounds__(64) void head_narrow_reduce_kernel(const float* part, int nblocks, int np, int Cin, int Cout, int nco, float* dw, long s_ci,
                                                               long s_co, float* dbias, int accumulate) {
  const int k = blockIdx.x;                                 // one partial column per workgroup, 64 lanes over the rows
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += (double)part[(size_t)b * np + k];
  s = mt_wave_sum_d(s);
  if (threadIdx.x != 0) return;
  const int co = k < nco * Cin ? k / Cin : k - nco * Cin, ci = k < nco * Cin ? k - co * Cin : -1;
  if (co >= Cout) return;
  if (ci >= 0) { float* o = dw + (long)ci * s_ci + (long)co * s_co; *o = accumulate ? *o + (float)s : (float)s; }
  else if (dbias != nullptr) dbias[co] = accumulate ? dbias[co] + (float)s : (float)s;
}
static bool head_bwd_narrow_ok(const mt_src_t* x, int dycs, int dxcs, int Cin, int Cout, long V, int N, const float* dx) {
  constexpr int use = 1;
  return use && Cout <= 4 && (Cin == 30 || Cin == 32) && x->cs == Cin && dxcs == Cin && N <= HN_BLOCKS &&
         (x->scale == nullptr || (x->slope >= 0.f && x->slope <= 1.f)) && (double)V * Cin * 4.0 < 2147483648.0;
}

static inline int head_bwd_waves(int N, long V) {
  const long ntiles = (long)N * ((V + 31) / 32);
  long w = ntiles / 32;                                            // >= 32 tiles per wave: the 8 - 16 KiB partial of a wave is written once
  if (w > 256 * 4 * 4) w = 256 * 4 * 4;                            // at most 4 workgroups of 4 waves per CU
  if (w < 4) w = 4;
  return (int)((w + 3) / 4 * 4);
}
// Cin <= 32 only: the two-input-tile instantiation (Cin <= 64) needs 182 VGPRs (one wave per SIMD) and measured 0.70 ms on the
// 24x96x96 level — slower than the generic kernels there; the instantiation is not dispatched (constexpr wide = 0)
extern "C" int mt_head_bwd_supported(int Cin, int Cout) {
  constexpr int wide = 0;
  return Cin >= 1 && Cin <= (wide ? 64 : 32) && Cout >= 1 && Cout <= 64;
}
extern "C" size_t mt_head_bwd_workspace(int N, long V, int Cin, int Cout) {
  if (!(Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64)) return 0;
  const size_t per = (size_t)((Cin + 31) / 32) * 2 * 1024;
  const size_t wide = (size_t)head_bwd_waves(N, V) * per * sizeof(float) + HB_SLICES * per * sizeof(double) + 64;
  const size_t narrow = (Cout <= 4) ? (size_t)HN_BLOCKS * (4 * 32 + 4) * sizeof(float) : 0;      // head_bwd_narrow_kernel: one partial row per workgroup
  return wide > narrow ? wide : narrow;
}
// storage types mt_head_bwd takes natively: x fp32 with dX fp32; x fp16 or bf16 with dX bf16 (even Cin and channel strides, dword-aligned
// bases); dY is the fp32 loss gradient
extern "C" int mt_head_bwd_io_supported(int xdtype, int xcs, int dxdtype, int dxcs, int Cin, int Cout) {
  if (xdtype == MT_F32 && dxdtype == MT_F32) return 1;
  if (!(mt_is16(xdtype) && dxdtype == MT_BF16)) return 0;
  return ((Cin & 1) || (xcs & 1) || (dxcs & 1)) ? 0 : 1;
}
extern "C" int mt_head_bwd(const mt_src_t* x, const float* dy, int dycs, int N, long V, int Cin, int Cout, const float* wpack_bwd,
                           float* dx, int dxcs, int dxdtype, int accumulate_dx, float* dw, long s_ci, long s_co, float* dbias, int accumulate_dw,
                           int* dbias_done, void* ws, size_t ws_bytes, mt_stream_t stream) {
  MT_REQUIRE(x && x->ptr && dy && wpack_bwd && dx && dw && N > 0 && V > 0, "head_bwd: null / empty argument");
  MT_REQUIRE(mt_head_bwd_io_supported(x->dtype, x->cs, dxdtype, dxcs, Cin, Cout) && !(((uintptr_t)x->ptr) & 3) && !(((uintptr_t)dx) & 3),
             "head_bwd: storage types (x %d, dX %d) not taken (ask mt_head_bwd_io_supported, convert with mt_cast)", x->dtype, dxdtype);
  const int xs = x->dtype;
  MT_REQUIRE(Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64, "head_bwd: Cin (%d) and Cout (%d) must be <= 64", Cin, Cout);
  MT_REQUIRE(x->C == Cin, "head_bwd: x->C != Cin");
  MT_REQUIRE((double)V * x->cs * 4.0 < 2147483648.0 && (double)V * dycs * 4.0 < 2147483648.0 && (double)V * dxcs * 4.0 < 2147483648.0, "head_bwd: sample larger than 2 GiB");
  if (ws == nullptr || ws_bytes < mt_head_bwd_workspace(N, V, Cin, Cout)) { mt_set_error("head_bwd: workspace too small"); return MT_EWORKSPACE; }
  HeadBwdParams P;
  P.x = *x; P.dy = dy; P.dycs = dycs; P.N = N; P.V = V; P.Cin = Cin; P.Cout = Cout; P.wpack = wpack_bwd;
  P.dx = dx; P.dxcs = dxcs; P.accumulate_dx = accumulate_dx; P.part = (float*)ws;
  if (head_bwd_narrow_ok(x, dycs, dxcs, Cin, Cout, V, N, dx)) {
    const int nco = Cout <= 2 ? 2 : 4, np = nco * Cin + nco;
    const int per_sample = HN_BLOCKS / N > 0 ? HN_BLOCKS / N : 1, nblocks = per_sample * N;
    MT_REQUIRE((size_t)nblocks * np * sizeof(float) <= ws_bytes, "head_bwd: workspace too small for the narrow form");
    hipStream_t st = (hipStream_t)stream;
#define HBN(CIN_, NCO_) do { if (xs == MT_F16) hipLaunchKernelGGL((head_bwd_narrow_kernel<CIN_, NCO_, MT_F16, MT_BF16>), dim3(nblocks), dim3(256), 0, st, P);          \
                             else if (xs == MT_BF16) hipLaunchKernelGGL((head_bwd_narrow_kernel<CIN_, NCO_, MT_BF16, MT_BF16>), dim3(nblocks), dim3(256), 0, st, P);   \
                             else hipLaunchKernelGGL((head_bwd_narrow_kernel<CIN_, NCO_>), dim3(nblocks), dim3(256), 0, st, P); } while (0)
    if (Cin == 30 && nco == 2) HBN(30, 2);
    else if (Cin == 30) HBN(30, 4);
    else if (nco == 2) HBN(32, 2);
    else HBN(32, 4);
#undef HBN
    hipLaunchKernelGGL(head_narrow_reduce_kernel, dim3(np), dim3(64), 0, st, (const float*)ws, nblocks, np, Cin, Cout, nco, dw, s_ci, s_co, dbias, accumulate_dw);
    MT_CHECK_LAUNCH("head_bwd_narrow");
    if (dbias_done != nullptr) *dbias_done = 1;
    return MT_OK;
  }
  P.nwaves = head_bwd_waves(N, V); P.ntiles = (long)N * ((V + 31) / 32);
  const int nci = (Cin + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  constexpr int staged = 1;
  const bool dense = staged && x->cs == Cin && dxcs == Cin && dycs == Cout && (mt_is16(xs) ? (Cin % 2) == 0 : true);
#define HB(NCI_, ST_) do { if (xs == MT_F16) hipLaunchKernelGGL((head_bwd_kernel<NCI_, MT_F16, MT_BF16, ST_>), dim3(P.nwaves / 4), dim3(256), 0, st, P);          \
                      else if (xs == MT_BF16) hipLaunchKernelGGL((head_bwd_kernel<NCI_, MT_BF16, MT_BF16, ST_>), dim3(P.nwaves / 4), dim3(256), 0, st, P);   \
                      else hipLaunchKernelGGL((head_bwd_kernel<NCI_, MT_F32, MT_F32, ST_>), dim3(P.nwaves / 4), dim3(256), 0, st, P); } while (0)
  if (dense) { if (nci == 1) HB(1, true); else HB(2, true); }
  else { if (nci == 1) HB(1, false); else HB(2, false); }
#undef HB
  MT_CHECK_LAUNCH("head_bwd");
  HeadBwdReduce R;
  R.part = (const float*)ws; R.nwaves = P.nwaves; R.nci = nci; R.Cin = Cin; R.Cout = Cout; R.dw = dw; R.s_ci = s_ci; R.s_co = s_co;
  R.dbias = dbias; R.accumulate = accumulate_dw;
  const size_t per = (size_t)nci * 2 * 1024;
  R.tmp = (double*)(((uintptr_t)((float*)ws + (size_t)P.nwaves * per) + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(head_bwd_reduce_a_kernel, dim3(mt_cdiv((long)per, 256), HB_SLICES), dim3(256), 0, st, R);
  hipLaunchKernelGGL(head_bwd_reduce_b_kernel, dim3(mt_cdiv((long)per, 256)), dim3(256), 0, st, R);
  MT_CHECK_LAUNCH("head_bwd_reduce");
  if (dbias_done != nullptr) *dbias_done = ((Cin % 32) != 0) ? 1 : 0;      // 0: the caller sums dY itself (mt_channel_sum)
  return MT_OK;
}


// ---- device probe --------------------------------------------------------------------------------------------------------------
// pw_fast_kernel, conv_gather_kernel, head_bwd_kernel and the Winograd stagers issue buffer_load_dwordx4 on addresses that are
// only dword-aligned (188-byte rows at 47 channels) and rely on raw buffers range-checking every dword of a load on its own (a
// 16-byte load that straddles num_records returns its in-range dwords and zeros for the rest).  Both are properties of gfx950 in
// the unaligned-access mode the ROCm driver configures; the probe below verifies them ON THE DEVICE IN USE so that a differently
// configured system fails loudly at library load instead of computing garbage.
__global__ void probe_straddle_kernel(const float* p, int nrec_bytes, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec_bytes, 0x00020000);
  const int off = threadIdx.x * 8;      // lane i reads floats 2i .. 2i+3
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
__global__ void probe_unaligned_kernel(const float* p, int nrec_bytes, float* out) {     // rows of 47 floats: 4-byte aligned only
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec_bytes, 0x00020000);
  const int off = (threadIdx.x * 47 + 1) * 4;
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}

extern "C" int mt_probe_device(void* scratch, size_t scratch_bytes, int* vector_loads_ok, char* arch, size_t arch_len, mt_stream_t stream) {
  MT_REQUIRE(scratch != nullptr && scratch_bytes >= 32768 && vector_loads_ok != nullptr, "probe_device: needs 32 KiB of device scratch");
  hipStream_t st = (hipStream_t)stream;
  *vector_loads_ok = 0;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { mt_set_error("probe_device: no device"); return MT_EHIP; }
  if (arch != nullptr && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { mt_set_error("probe_device: built for gfx950, the device is %s", prop.gcnArchName); return MT_EUNSUPPORTED; }
  float* in = (float*)scratch;                 // 64 * 47 + 8 floats of input, then 256 floats of output
  const int n = 64 * 47 + 8;
  float* out = in + 4096;
  float* h = (float*)malloc((size_t)n * sizeof(float));
  float r[256];
  if (h == nullptr) { mt_set_error("probe_device: out of host memory"); return MT_EHIP; }
  for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
  bool ok = hipMemcpyAsync(in, h, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess;
  // (1) 30 floats in range: lanes 13 / 14 straddle the end of the buffer, lanes >= 15 are entirely outside
  hipLaunchKernelGGL(probe_straddle_kernel, dim3(1), dim3(64), 0, st, in, 30 * 4, out);
  ok = ok && hipMemcpyAsync(r, out, sizeof(r), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (ok)
    for (int i = 0; i < 64; ++i)
      for (int e = 0; e < 4; ++e) {
        const int idx = 2 * i + e;
        if (r[4 * i + e] != (idx < 30 ? (float)(idx + 1) : 0.f)) ok = false;
      }
  // (2) dword-aligned 16-byte loads return the right four values
  if (ok) {
    hipLaunchKernelGGL(probe_unaligned_kernel, dim3(1), dim3(64), 0, st, in, n * 4, out);
    ok = hipMemcpyAsync(r, out, sizeof(r), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (ok)
      for (int i = 0; i < 64; ++i)
        for (int e = 0; e < 4; ++e)
          if (r[4 * i + e] != (float)(i * 47 + 1 + e + 1)) ok = false;
  }
  free(h);
  if (hipGetLastError() != hipSuccess) ok = false;
  *vector_loads_ok = ok ? 1 : 0;
  return MT_OK;
}
