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

// ------------------------------------------------------------------------------------------------
// Export post-processing on the device: the probabilities of a volume are resampled back to the original
// voxel grid and turned into a label map in ONE pass — save_segmentation_nifti_from_softmax (segmentation_export.py:27-160) =
// resample_data_or_seg(is_seg=False, order 1[, separate z with order_z 0]) (preprocessing.py:109-197) -> per-region threshold
// in regions_class_order or argmax -> re-insertion into the uncropped volume.  The 47-channel resampled volume (25 GB at 512^3)
// never exists: each output voxel interpolates every channel from the low-resolution probabilities and keeps only its label.
// Coordinate rule of skimage.transform.resize(order=1, mode='edge', anti_aliasing=False) = scipy.ndimage.zoom(order=1,
// mode='nearest', grid_mode=True): x = (o + 0.5) * in/out - 0.5, clamped to [0, in-1]; the separate-z branch samples the
// anisotropic axis at floor(x + 0.5) (map_coordinates order 0, mode 'nearest', preprocessing.py:166-174).
struct ResampleParams {
  const float* probs; const int32_t* order; uint8_t* out;
  int C, D, H, W, OD, OH, OW, sep_axis, use_regions;
  long FD, FH, FW;          // dims of the uncropped output volume
  int bD, bH, bW;           // insertion offset (crop_bbox lower corner)
  int cD, cH, cW;           // voxels actually written per dim (clipped to the volume)
};
__device__ __forceinline__ void mt_axis_coord(int o, int I, int O, bool nearest, int& i0, int& i1, float& f) {
  const double x = ((double)o + 0.5) * ((double)I / (double)O) - 0.5;
  if (nearest) {
    int i = (int)floor(x + 0.5); i = i < 0 ? 0 : (i > I - 1 ? I - 1 : i);
    i0 = i1 = i; f = 0.f;
    return;
  }
  const double xc = x < 0.0 ? 0.0 : (x > (double)(I - 1) ? (double)(I - 1) : x);
  const int a = (int)floor(xc);
  i0 = a; i1 = a + 1 > I - 1 ? I - 1 : a + 1; f = (float)(xc - (double)a);
}
__global__ __launch_bounds__(256) void resample_classify_kernel(const ResampleParams P) {
  const long total = (long)P.cD * P.cH * P.cW;
  const size_t V = (size_t)P.D * P.H * P.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ow = (int)(i % P.cW), oh = (int)((i / P.cW) % P.cH), od = (int)(i / ((long)P.cW * P.cH));
    int d0, d1, h0, h1, w0, w1; float fd, fh, fw;
    mt_axis_coord(od, P.D, P.OD, P.sep_axis == 0, d0, d1, fd);
    mt_axis_coord(oh, P.H, P.OH, P.sep_axis == 1, h0, h1, fh);
    mt_axis_coord(ow, P.W, P.OW, P.sep_axis == 2, w0, w1, fw);
    const size_t o00 = ((size_t)d0 * P.H + h0) * P.W, o01 = ((size_t)d0 * P.H + h1) * P.W;
    const size_t o10 = ((size_t)d1 * P.H + h0) * P.W, o11 = ((size_t)d1 * P.H + h1) * P.W;
    int s = 0; float best = -3.0e38f;
    for (int c = 0; c < P.C; ++c) {
      const float* p = P.probs + (size_t)c * V;
      const float a00 = p[o00 + w0] + fw * (p[o00 + w1] - p[o00 + w0]);
      const float a01 = p[o01 + w0] + fw * (p[o01 + w1] - p[o01 + w0]);
      const float a10 = p[o10 + w0] + fw * (p[o10 + w1] - p[o10 + w0]);
      const float a11 = p[o11 + w0] + fw * (p[o11 + w1] - p[o11 + w0]);
      const float a0 = a00 + fh * (a01 - a00), a1 = a10 + fh * (a11 - a10);
      const float v = a0 + fd * (a1 - a0);
      if (P.use_regions) { if (v > 0.5f) s = P.order[c]; }
      else if (v > best) { best = v; s = c; }
    }
    P.out[((size_t)(P.bD + od) * P.FH + (P.bH + oh)) * P.FW + (P.bW + ow)] = (uint8_t)s;
  }
}
extern "C" int mt_resample_classify(const float* probs, int C, int D, int H, int W, int OD, int OH, int OW, int sep_axis,
                                    const int32_t* class_order, int use_regions, uint8_t* out, long FD, long FH, long FW,
                                    int bD, int bH, int bW, mt_stream_t stream) {
  MT_REQUIRE(probs && out && C > 0 && D > 0 && H > 0 && W > 0 && OD > 0 && OH > 0 && OW > 0, "resample_classify: bad sizes");
  MT_REQUIRE(sep_axis >= -1 && sep_axis <= 2 && (!use_regions || class_order), "resample_classify: bad mode");
  MT_REQUIRE(bD >= 0 && bH >= 0 && bW >= 0 && bD < FD && bH < FH && bW < FW, "resample_classify: insertion offset outside the volume");
  ResampleParams P;
  P.probs = probs; P.order = class_order; P.out = out; P.C = C; P.D = D; P.H = H; P.W = W; P.OD = OD; P.OH = OH; P.OW = OW;
  P.sep_axis = sep_axis; P.use_regions = use_regions; P.FD = FD; P.FH = FH; P.FW = FW; P.bD = bD; P.bH = bH; P.bW = bW;
  P.cD = (int)((bD + (long)OD <= FD) ? OD : FD - bD); P.cH = (int)((bH + (long)OH <= FH) ? OH : FH - bH);
  P.cW = (int)((bW + (long)OW <= FW) ? OW : FW - bW);
  const long total = (long)P.cD * P.cH * P.cW;
  int blocks = mt_cdiv(total, 256); if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(resample_classify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("resample_classify");
  return MT_OK;
}

// ---- tile extraction with mirror flips (neural_network.py:531-586: torch.flip(x, axes) per mirror combination) ---------------
// out[k][c][d][h][w] = vol[c][x0_k + fd(d)][y0_k + fh(h)][z0_k + fw(w)],  f(i) = flip ? size-1-i : i.  One launch builds the whole
// network batch (all mirror combinations of up to 8 tiles) straight from the volume: the flips are index arithmetic, the
// torch.flip / torch.cat / .contiguous() copies of the reference's loop never exist.
struct ExtractDesc { int n; int x0[64], y0[64], z0[64], fl[64]; };
__global__ __launch_bounds__(256) void extract_tiles_kernel(const float* __restrict__ vol, int C, long X, long Y, long Z,
                                                            float* __restrict__ out, int D, int H, int W, ExtractDesc ds) {
  const int k = blockIdx.z;
  const int c = blockIdx.y / D, d = blockIdx.y % D;
  const int fl = ds.fl[k];
  const int sd = (fl & 1) ? D - 1 - d : d;
  const float* src = vol + ((size_t)c * X + ds.x0[k] + sd) * (size_t)Y * Z;
  float* dst = out + (((size_t)k * C + c) * D + d) * (size_t)H * W;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
    const int h = e / W, w = e - h * W;
    const int sh = (fl & 2) ? H - 1 - h : h, sw = (fl & 4) ? W - 1 - w : w;
    dst[e] = src[(size_t)(ds.y0[k] + sh) * Z + ds.z0[k] + sw];
  }
}
extern "C" int mt_extract_tiles(const float* vol, int C, long X, long Y, long Z, float* out, int ntiles, int D, int H, int W,
                                const int32_t* desc, mt_stream_t stream) {
  MT_REQUIRE(vol && out && desc && C > 0 && ntiles > 0 && ntiles <= 64 && D > 0 && H > 0 && W > 0, "extract_tiles: bad args (ntiles <= 64)");
  ExtractDesc ds;
  ds.n = ntiles;
  for (int k = 0; k < ntiles; ++k) {
    ds.x0[k] = desc[4 * k]; ds.y0[k] = desc[4 * k + 1]; ds.z0[k] = desc[4 * k + 2]; ds.fl[k] = desc[4 * k + 3];
    MT_REQUIRE(ds.x0[k] >= 0 && ds.y0[k] >= 0 && ds.z0[k] >= 0 && ds.x0[k] + D <= X && ds.y0[k] + H <= Y && ds.z0[k] + W <= Z,
               "extract_tiles: tile %d leaves the volume", k);
  }
  int bx = mt_cdiv((long)H * W, 256 * 4); if (bx < 1) bx = 1;
  hipLaunchKernelGGL(extract_tiles_kernel, dim3(bx, C * D, ntiles), dim3(256), 0, (hipStream_t)stream, vol, C, X, Y, Z, out, D, H, W, ds);
  MT_CHECK_LAUNCH("extract_tiles");
  return MT_OK;
}
