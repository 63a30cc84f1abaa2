// infer.hip — device side of sliding-window inference (neural_network.py:287-428,502-591).
// The reference flips tensors with torch.flip, accumulates the 8 mirror predictions, multiplies by the
// Gaussian, copies each 333 MB tile to the HOST and adds it into numpy aggregates.  Here the flip is
// index arithmetic, and the aggregate lives in HBM (288 GB) for the whole volume.
#include "mt_common.h"

// acc[c][d][h][w] (+)= weight * nonlin(logits[fd][fh][fw][c]);  logits are of the FLIPPED input, so the
// un-flip (neural_network.py:531-586 `torch.flip(pred, axes)`) is the same index reflection.
// One workgroup per (d, h) row: the source row (W voxels x cs channels, contiguous) is read coalesced into LDS with an odd
// channel pitch, each thread then owns one voxel (all C channels from LDS: softmax needs them together) and the C output
// planes are written coalesced along w.  NDHWC logits -> NCDHW accumulator is a transpose; doing it through LDS keeps both
// sides of it at full line width.
__global__ __launch_bounds__(256) void flip_accumulate_kernel(const float* __restrict__ logits, int cs, int D, int H, int W, int C,
                                                              int fD, int fH, int fW, int nonlin, float weight,
                                                              float* __restrict__ acc, int first) {
  extern __shared__ float row[];                 // [256 voxels][C | 1]
  const int CP = C | 1;
  const long V = (long)D * H * W;
  const int d = blockIdx.x / H, h = blockIdx.x % H;
  const int sd = fD ? D - 1 - d : d, shh = fH ? H - 1 - h : h;
  const float* srow = logits + ((size_t)sd * H + shh) * (size_t)W * cs;
  for (int w0 = 0; w0 < W; w0 += 256) {
    const int nw = (W - w0 < 256) ? (W - w0) : 256;
    // source voxels of output columns [w0, w0+nw): sw = fW ? W-1-w : w  -> a contiguous run either way
    const int s0 = fW ? W - w0 - nw : w0;
    __syncthreads();
    for (int e = threadIdx.x; e < nw * C; e += 256) {
      const int sv = e / C, c = e - sv * C;
      row[sv * CP + c] = srow[(size_t)(s0 + sv) * cs + c];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nw) {
      const int sv = fW ? nw - 1 - t : t;
      const float* src = row + sv * CP;
      const long v = ((long)d * H + h) * W + w0 + t;
      if (nonlin == 2) {
        float mx = -3.0e38f;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, src[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(src[c] - mx);
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) {
          const float p = expf(src[c] - mx) * inv * weight;
          float* a = acc + (size_t)c * V + v;
          *a = first ? p : *a + p;
        }
      } else {
        for (int c = 0; c < C; ++c) {
          float x = src[c];
          if (nonlin == 1) x = 1.f / (1.f + expf(-x));
          x *= weight;
          float* a = acc + (size_t)c * V + v;
          *a = first ? x : *a + x;
        }
      }
    }
  }
}
extern "C" int mt_flip_accumulate(const float* logits, int cs, int D, int H, int W, int C, int flipD, int flipH, int flipW,
                                  int nonlin, float weight, float* acc, int first, mt_stream_t stream) {
  MT_REQUIRE(logits && acc && D > 0 && H > 0 && W > 0 && C > 0, "flip_accumulate: bad args");
  const size_t ldsb = (size_t)256 * (C | 1) * sizeof(float);
  MT_REQUIRE(ldsb <= 160 * 1024, "flip_accumulate: too many classes (%d)", C);
  if (ldsb > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)flip_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) { mt_set_error("flip_accumulate: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MT_EHIP; }
  }
  hipLaunchKernelGGL(flip_accumulate_kernel, dim3((unsigned)(D * H)), dim3(256), ldsb, (hipStream_t)stream, logits, cs, D, H, W, C,
                     flipD, flipH, flipW, nonlin, weight, acc, first);
  MT_CHECK_LAUNCH("flip_accumulate");
  return MT_OK;
}

// agg[c, x0+d, y0+h, z0+w] += acc[c,d,h,w] * gauss[d,h,w];  nb[x0+d, y0+h, z0+w] += gauss  (neural_network.py:388-394, 588-589)
__global__ __launch_bounds__(256) void tile_accumulate_kernel(const float* __restrict__ acc, const float* __restrict__ gauss, int C,
                                                              int D, int H, int W, float* __restrict__ agg, float* __restrict__ nb,
                                                              long aX, long aY, long aZ, int x0, int y0, int z0) {
  const long V = (long)D * H * W;
  const long aV = aX * aY * aZ;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
    const size_t o = ((size_t)(x0 + d) * aY + (y0 + h)) * aZ + (z0 + w);
    const float g = gauss ? gauss[v] : 1.f;
    for (int c = 0; c < C; ++c) agg[(size_t)c * aV + o] += acc[(size_t)c * V + v] * g;
    if (nb) nb[o] += g;
  }
}
extern "C" int mt_tile_accumulate(const float* acc, const float* gauss, int C, int D, int H, int W, float* agg, float* nb,
                                  long aX, long aY, long aZ, int x0, int y0, int z0, mt_stream_t stream) {
  MT_REQUIRE(acc && agg && C > 0 && D > 0 && H > 0 && W > 0, "tile_accumulate: bad args");
  MT_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + D <= aX && y0 + H <= aY && z0 + W <= aZ, "tile_accumulate: tile outside aggregate");
  const long V = (long)D * H * W;
  int blocks = mt_cdiv(V, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(tile_accumulate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, gauss, C, D, H, W, agg, nb, aX, aY,
                     aZ, x0, y0, z0);
  MT_CHECK_LAUNCH("tile_accumulate");
  return MT_OK;
}

// probs = agg / nb (in place); seg: regions -> for i,c in enumerate(order): seg[probs[i] > 0.5] = c ; else argmax (first max)
__global__ __launch_bounds__(256) void normalize_threshold_kernel(float* __restrict__ agg, const float* __restrict__ nb, int C, long V,
                                                                  const int32_t* __restrict__ order, int use_regions,
                                                                  int32_t* __restrict__ seg) {
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const float d = nb ? nb[v] : 1.f;
    int32_t s = 0;
    float best = -3.0e38f;
    for (int c = 0; c < C; ++c) {
      const float p = agg[(size_t)c * V + v] / d;
      agg[(size_t)c * V + v] = p;
      if (use_regions) { if (p > 0.5f) s = order[c]; }
      else if (p > best) { best = p; s = c; }
    }
    if (seg) seg[v] = s;
  }
}
extern "C" int mt_normalize_threshold(float* agg, const float* nb, int C, long V, const int32_t* class_order, int use_regions,
                                      int32_t* seg, mt_stream_t stream) {
  MT_REQUIRE(agg && C > 0 && V > 0 && (!use_regions || class_order), "normalize_threshold: bad args");
  int blocks = mt_cdiv(V, 256); if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(normalize_threshold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, agg, nb, C, V, class_order, use_regions, seg);
  MT_CHECK_LAUNCH("normalize_threshold");
  return MT_OK;
}
