// conv16_common.h — pieces shared by the 16-bit matrix convolution kernels (conv_bf16.inc inside conv_lds.hip, conv_x16.hip):
// operand types, the MFMA wrapper, the channel-pair exchange of the 16-bit epilogue and the per-tile dword store path.
#pragma once
#include "mt_common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mt_pack_bf16(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 v; v[0] = a; v[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// ---- 16-bit OUTPUT of a 32x32 accumulator tile (mt_conv3d_t.odtype == MT_BF16 | MT_F16) ------------------------------------------------------
// Lane (li, lhalf) of a 32x32 MFMA accumulator holds column (output channel) li of rows (voxels) (j&3) + 8*(j>>2) + 4*lhalf.  Storing
// 2-byte elements one lane at a time would halve the bytes per store instruction; instead the lanes of a channel pair (li even, li odd)
// trade one value per two accumulator rows, so that the EVEN lane holds both channels of voxel j and the ODD lane both channels of
// voxel j+1: every lane stores one dword, a store instruction writes two full 64-byte channel runs.  Returns a = channel (co & ~1),
// b = channel (co | 1) of the lane's voxel.
__device__ __forceinline__ void mt_pair_exchange(float vj, float vj1, bool odd, float& a, float& b) {
  const float send = odd ? vj : vj1;
  const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  a = odd ? recv : vj;
  b = odd ? vj1 : recv;
}
// statistics of the stored (rounded) values: the lane accumulated s[0] for channel (co & ~1) and s[1] for (co | 1) over ITS voxels;
// the channel's total over both lanes of the pair, returned in the lane that owns the channel
__device__ __forceinline__ float mt_pair_combine(float s0, float s1, bool odd) {
  const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
  const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
  return odd ? s1 + t1 : s0 + t0;
}

// One 16-channel chunk: 27 taps, one MFMA per (tap, M tile, N tile).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MTY> __device__ __forceinline__ f32x16 mt_mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (MTY == MT_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Epilogue of one workgroup tile: bias, optional accumulation into the destination, stores (fp32 dwords or 16-bit channel pairs), and
// this lane's statistics partials: a1[n] / a2[n] = sum / sum of squares of channel (ntile0 + n) * 32 + li over the wave's voxels
// (already combined over the two lane halves; 0 for channels beyond Cout).  `wave` indexes the wave's M tiles (0 .. NW-1).
template <int MW, int RH, int TD, int NT, int NW, int OS>
__device__ __forceinline__ void bf16_store_tile(const mt_conv3d_t& c, f32x16 (&acc)[NT][TD * RH / NW], int wave, int lane, int nb,
                                                int od0, int oh0, int ow0, int ntile0, float (&a1)[NT], float (&a2)[NT]) {
  constexpr bool OB = OS != MT_F32;
  constexpr int MH = 32 / MW, TH = MH * RH, TW = MW, MT = TD * RH / NW;
  const int li = lane & 31, lhalf = lane >> 5;
  const bool interior = (od0 + TD <= c.Do) && (oh0 + TH <= c.Ho) && (ow0 + TW <= c.Wo);   // block-uniform
  const size_t out_sample = (size_t)c.Do * c.Ho * c.Wo;
  constexpr int OEB = mt_ebytes<OS>();     // bytes per stored output element
  __amdgpu_buffer_rsrc_t r0d = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)c.out0 + (size_t)nb * out_sample * c.ocs0 * OEB), 0,
                                                                 (int)(out_sample * c.ocs0 * OEB), 0x00020000);
  const bool split = c.csplit < c.Cout;
  __amdgpu_buffer_rsrc_t r1d = r0d;
  if (split) r1d = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)c.out1 + (size_t)nb * out_sample * c.ocs1 * OEB), 0,
                                                     (int)(out_sample * c.ocs1 * OEB), 0x00020000);
  int loff[NT]; float bv[NT]; bool use1[NT]; float s1[NT], s2[NT];
  if constexpr (OB) {
    // the three block-uniform conditions (tile inside the volume, two destinations, accumulate) as COMPILE-TIME constants of the common
    // cases: as run-time values they turn every store offset and every statistics update into a select (480 v_cndmask in the epilogue)
    auto run = [&](auto IC, auto SC, auto AC) {
    const bool interior_ = decltype(IC)::value == 2 ? interior : (decltype(IC)::value == 1);
    const bool split_ = decltype(SC)::value == 2 ? split : (decltype(SC)::value == 1);
    const bool acc_ = decltype(AC)::value == 2 ? (c.accumulate != 0) : (decltype(AC)::value == 1);
    // ---- 16-bit destination: one dword (two channels of one voxel) per lane and pair of accumulator rows (mt_pair_exchange)
    const bool odd = li & 1;
    const int lane_col = 4 * lhalf + (odd ? 1 : 0);
    float q1[NT][2], q2[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = (ntile0 + n) * 32 + li, coe = co & ~1;
      const bool covalid = coe + 1 < c.Cout;                 // Cout, csplit, ocs0, ocs1 are even (mt_conv3d_io_supported)
      bv[n] = (c.bias != nullptr && co < c.Cout) ? c.bias[co] : 0.f;
      use1[n] = split_ && !(coe < c.csplit);
      const int ocs = use1[n] ? c.ocs1 : c.ocs0, cofs = use1[n] ? coe - c.csplit : coe;
      loff[n] = covalid ? (lane_col * ocs + cofs) * 2 : (int)0x80000000;
      q1[n][0] = q1[n][1] = q2[n][0] = q2[n][1] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mt = wave * MT + m;
      const int dm = mt / RH, rh = mt % RH;
      const int od = od0 + dm;
      const int vox0 = ((od * c.Ho) + (oh0 + rh * MH)) * c.Wo + ow0;
      auto jgeom = [&](int j, bool& ok, int& so0, int& so1) {          // j even: rows j (even lanes) and j + 1 (odd lanes)
        const int ivj = (j & 3) + 8 * (j >> 2);
        const int r = ivj / MW, colj = ivj % MW;
        ok = true;
        if (!interior_) ok = (od < c.Do) && (oh0 + rh * MH + r < c.Ho) && (ow0 + colj + lane_col < c.Wo);
        so0 = (vox0 + r * c.Wo + colj) * c.ocs0 * 2;
        so1 = (vox0 + r * c.Wo + colj) * c.ocs1 * 2;
      };
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += 4) {
        unsigned prev[2][NT];
        if (acc_) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            bool ok; int so0, so1; jgeom(j0 + 2 * jj, ok, so0, so1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const int off = ok ? loff[n] : (int)0x80000000;
              if (!split_) prev[jj][n] = __builtin_amdgcn_raw_buffer_load_b32(r0d, off, so0, 0);
              else {
                const int offa = use1[n] ? (int)0x80000000 : off, offb = use1[n] ? off : (int)0x80000000;
                prev[jj][n] = __builtin_amdgcn_raw_buffer_load_b32(r0d, offa, so0, 0) | __builtin_amdgcn_raw_buffer_load_b32(r1d, offb, so1, 0);
              }
            }
          }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = j0 + 2 * jj;
          bool ok; int so0, so1; jgeom(j, ok, so0, so1);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int off = ok ? loff[n] : (int)0x80000000;
            float a, b;
            mt_pair_exchange(acc[n][m][j] + bv[n], acc[n][m][j + 1] + bv[n], odd, a, b);
            if (acc_) { a += mt_lo16<OS>(prev[jj][n]); b += mt_hi16<OS>(prev[jj][n]); }
            const unsigned pk = mt_pk16<OS>(a, b);
            if (!split_) {
              __builtin_amdgcn_raw_buffer_store_b32(pk, r0d, off, so0, 0);
            } else {
              const int offa = use1[n] ? (int)0x80000000 : off, offb = use1[n] ? off : (int)0x80000000;
              __builtin_amdgcn_raw_buffer_store_b32(pk, r0d, offa, so0, 0);
              __builtin_amdgcn_raw_buffer_store_b32(pk, r1d, offb, so1, 0);
            }
            if (interior_ || off >= 0) {           // statistics of the values as stored
              const float ar = mt_lo16<OS>(pk), br = mt_hi16<OS>(pk);
              q1[n][0] += ar; q2[n][0] = fmaf(ar, ar, q2[n][0]);
              q1[n][1] += br; q2[n][1] = fmaf(br, br, q2[n][1]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      s1[n] = mt_pair_combine(q1[n][0], q1[n][1], odd);
      s2[n] = mt_pair_combine(q2[n][0], q2[n][1], odd);
      const int co = (ntile0 + n) * 32 + li;
      loff[n] = co < c.Cout ? 0 : (int)0x80000000;           // (only its sign is used below: channel validity)
    }
    };
    typedef std::integral_constant<int, 0> C0; typedef std::integral_constant<int, 1> C1; typedef std::integral_constant<int, 2> CR;
    if (interior && !split && !c.accumulate) run(C1(), C0(), C0());
    else if (interior && !split) run(C1(), C0(), C1());
    else run(CR(), CR(), CR());
  } else {
  const int lane_col = 4 * lhalf;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = (ntile0 + n) * 32 + li;
    const bool covalid = co < c.Cout;
    bv[n] = (c.bias != nullptr && covalid) ? c.bias[co] : 0.f;
    use1[n] = split && !(co < c.csplit);
    const int ocs = use1[n] ? c.ocs1 : c.ocs0, cofs = use1[n] ? co - c.csplit : co;
    loff[n] = covalid ? (lane_col * ocs + cofs) * 4 : (int)0x80000000;
    s1[n] = 0.f; s2[n] = 0.f;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wave * MT + m;
    const int dm = mt / RH, rh = mt % RH;
    const int od = od0 + dm;
    const int vox0 = ((od * c.Ho) + (oh0 + rh * MH)) * c.Wo + ow0;
    auto jgeom = [&](int j, bool& ok, int& so0, int& so1) {
      const int ivj = (j & 3) + 8 * (j >> 2);
      const int r = ivj / MW, colj = ivj % MW;
      ok = true;
      if (!interior) ok = (od < c.Do) && (oh0 + rh * MH + r < c.Ho) && (ow0 + colj + lane_col < c.Wo);
      so0 = (vox0 + r * c.Wo + colj) * c.ocs0 * 4;
      so1 = (vox0 + r * c.Wo + colj) * c.ocs1 * 4;
    };
    // accumulate (residual blocks): the old values of four accumulator rows are requested together, before their stores — one
    // load -> add -> store round trip per element otherwise
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
      float prev[4][NT];
      if (c.accumulate) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          bool ok; int so0, so1; jgeom(j0 + jj, ok, so0, so1);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int off = ok ? loff[n] : (int)0x80000000;
            if (!split) prev[jj][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0d, off, so0, 0));
            else {
              const int offa = use1[n] ? (int)0x80000000 : off, offb = use1[n] ? off : (int)0x80000000;
              prev[jj][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0d, offa, so0, 0)) +
                            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1d, offb, so1, 0));
            }
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        bool ok; int so0, so1; jgeom(j, ok, so0, so1);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int off = ok ? loff[n] : (int)0x80000000;       // out-of-volume lanes and channels are dropped by the bounds check
          float v = acc[n][m][j] + bv[n];
          if (c.accumulate) v += prev[jj][n];
          if (!split) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r0d, off, so0, 0);
          } else {
            const int offa = use1[n] ? (int)0x80000000 : off, offb = use1[n] ? off : (int)0x80000000;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r0d, offa, so0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r1d, offb, so1, 0);
          }
          if (interior) { s1[n] += v; s2[n] = fmaf(v, v, s2[n]); }
          else if (off >= 0) { s1[n] += v; s2[n] = fmaf(v, v, s2[n]); }
        }
      }
    }
  }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float t1 = loff[n] >= 0 ? s1[n] : 0.f, t2 = loff[n] >= 0 ? s2[n] : 0.f;
    t1 += __shfl_xor(t1, 32, 64);
    t2 += __shfl_xor(t2, 32, 64);
    a1[n] = t1; a2[n] = t2;
  }
}
