// pointwise.hip — 1x1x1 convolutions and kernel==stride transposed convolutions on fp32 MFMA.
//
// Replaces: seg heads nn.Conv3d(C, num_classes, 1) (generic_UNet.py:349-351; generic_modular_UNet.py:244,251),
// strided 1x1x1 skip projections (conv_blocks.py:192-197), nn.ConvTranspose3d(k == stride, bias=False)
// (generic_UNet.py:335-336; generic_modular_UNet.py:236-237) and, with transposed packed weights,
// the backward-data of the 1x1x1 convs.
//
// One wave = 32 base voxels x 32 output channels; A (32 voxels x 2 channels) is gathered straight
// from global memory with the lazy InstanceNorm+LeakyReLU applied on load, B comes from the packed
// weights; for a transposed conv every tap is an independent GEMM whose rows are scattered to
// out[base*so + tap] — written directly into the first half of the skip-concat buffer (ocs).
#include "mt_common.h"

struct PwKParams {
  mt_pointwise_t c;
  int ntaps, nkp, nsb;
  long Vb;
};

__global__ __launch_bounds__(256) void pointwise_kernel(const PwKParams P) {
  const mt_pointwise_t& c = P.c;
  __shared__ float red[4 * 32 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lhalf = lane >> 5;
  const int nb = blockIdx.x / P.nsb, sb = blockIdx.x % P.nsb;
  const int ntile = blockIdx.y;
  const long m0 = (long)sb * 128 + wave * 32;
  const mt_src_t& S = c.src;

  // A operand addressing: this lane's base voxel
  const long bv = m0 + li;
  const bool vok = bv < P.Vb;
  const int wb = (int)(bv % c.Wb), hb = (int)((bv / c.Wb) % c.Hb), db = (int)(bv / ((long)c.Wb * c.Hb));
  const float* ap = S.ptr + ((size_t)((size_t)((size_t)nb * c.Di + db * c.siD) * c.Hi + hb * c.siH) * c.Wi + wb * c.siW) * S.cs;
  const bool aff = S.scale != nullptr;
  const float* scp = aff ? S.scale + (size_t)nb * S.C : nullptr;
  const float* shp = aff ? S.shift + (size_t)nb * S.C : nullptr;

  const int co = ntile * 32 + li;
  const bool covalid = co < c.Cout;
  const float bias = (c.bias != nullptr && covalid) ? c.bias[co] : 0.f;
  const int Ho = c.Hb * c.soH, Wo = c.Wb * c.soW, Do = c.Db * c.soD;
  float s1 = 0.f, s2 = 0.f;

  for (int tap = 0; tap < P.ntaps; ++tap) {
    const int tw = tap % c.soW, th = (tap / c.soW) % c.soH, tdd = tap / (c.soW * c.soH);
    const float* wq = c.wpack + ((size_t)ntile * P.ntaps + tap) * ((size_t)P.nkp * 64) + lane;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int kp0 = 0; kp0 < P.nkp; kp0 += 4) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kp = kp0 + u;
        const int ci = 2 * kp + lhalf;
        float x = 0.f;
        if (vok && kp < P.nkp && ci < c.Cin) {
          x = ap[ci];
          if (aff) x = mt_lrelu(fmaf(x, scp[ci], shp[ci]), S.slope);
        }
        a[u] = x;
        b[u] = (kp < P.nkp) ? wq[(size_t)kp * 64] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int iv = (j & 3) + 8 * (j >> 2) + 4 * lhalf;
      const long v = m0 + iv;
      if (covalid && v < P.Vb) {
        const int w2 = (int)(v % c.Wb), h2 = (int)((v / c.Wb) % c.Hb), d2 = (int)(v / ((long)c.Wb * c.Hb));
        const size_t idx = ((size_t)((size_t)((size_t)nb * Do + d2 * c.soD + tdd) * Ho + h2 * c.soH + th) * Wo + w2 * c.soW + tw) * c.ocs + co;
        float val = acc[j] + bias;
        if (c.accumulate) val += c.out[idx];
        c.out[idx] = val;
        s1 += val; s2 += val * val;
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

extern "C" int mt_pointwise_stats_blocks(const mt_pointwise_t* p) {
  if (p == nullptr) return -1;
  return mt_cdiv((long)p->Db * p->Hb * p->Wb, 128);
}

extern "C" int mt_pointwise_fwd(const mt_pointwise_t* p, mt_stream_t stream) {
  MT_REQUIRE(p != nullptr, "pointwise: null params");
  MT_REQUIRE(p->N > 0 && p->Db > 0 && p->Hb > 0 && p->Wb > 0 && p->Cin > 0 && p->Cout > 0, "pointwise: empty problem");
  MT_REQUIRE(p->siD >= 1 && p->siD <= 2 && p->siH >= 1 && p->siH <= 2 && p->siW >= 1 && p->siW <= 2, "pointwise: input stride must be 1 or 2");
  MT_REQUIRE(p->soD >= 1 && p->soD <= 2 && p->soH >= 1 && p->soH <= 2 && p->soW >= 1 && p->soW <= 2, "pointwise: output stride must be 1 or 2");
  MT_REQUIRE((p->Db - 1) * p->siD < p->Di && (p->Hb - 1) * p->siH < p->Hi && (p->Wb - 1) * p->siW < p->Wi, "pointwise: base grid exceeds stored input");
  MT_REQUIRE(p->src.C == p->Cin, "pointwise: src.C != Cin");
  MT_REQUIRE(p->src.ptr && p->wpack && p->out, "pointwise: null pointers");
  PwKParams P;
  P.c = *p;
  P.ntaps = p->soD * p->soH * p->soW;
  P.nkp = (p->Cin + 1) / 2;
  P.Vb = (long)p->Db * p->Hb * p->Wb;
  P.nsb = mt_cdiv(P.Vb, 128);
  dim3 grid((unsigned)(P.nsb * p->N), (unsigned)mt_cdiv(p->Cout, 32), 1);
  hipLaunchKernelGGL(pointwise_kernel, grid, dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("pointwise");
  return MT_OK;
}
