// prep.hip — device-side target preparation: the deep-supervision label pyramid.
//
// Replaces: DownsampleSegForDSTransform2 / downsample_seg_for_ds_transform2 (downsampling.py:70-104, order 0) and
// RemoveLabelTransform(-1, 0) (data_augmentation_moreDA.py:117), which the reference runs in CPU augmentation workers.
// resize_segmentation(order 0) is batchgenerators -> skimage.transform.resize(order=0, mode="edge", anti_aliasing=False),
// which delegates to scipy.ndimage.zoom(order=0, grid_mode=True): output voxel o samples input coordinate
// (o + 0.5) * in/out - 0.5 and order 0 takes floor(coord + 0.5) = floor((o + 0.5) * in/out), clamped to the volume.
// HBM-bound: every level reads |out| scattered labels and writes |out| floats.
#include "mt_common.h"

struct DsParams {
  const float* src; float* dst;
  int NC, Di, Hi, Wi, Do, Ho, Wo;
  double fd, fh, fw;       // in/out per axis
  int remove_minus_one;
};
__global__ __launch_bounds__(256) void downsample_seg_kernel(const DsParams P) {
  const long total = (long)P.NC * P.Do * P.Ho * P.Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int ow = (int)(r % P.Wo); r /= P.Wo;
    const int oh = (int)(r % P.Ho); r /= P.Ho;
    const int od = (int)(r % P.Do);
    const long nc = r / P.Do;
    int sd = (int)floor((od + 0.5) * P.fd), sh = (int)floor((oh + 0.5) * P.fh), sw = (int)floor((ow + 0.5) * P.fw);
    sd = sd < 0 ? 0 : (sd >= P.Di ? P.Di - 1 : sd);
    sh = sh < 0 ? 0 : (sh >= P.Hi ? P.Hi - 1 : sh);
    sw = sw < 0 ? 0 : (sw >= P.Wi ? P.Wi - 1 : sw);
    float v = P.src[((size_t)((size_t)nc * P.Di + sd) * P.Hi + sh) * P.Wi + sw];
    if (P.remove_minus_one && v == -1.f) v = 0.f;
    P.dst[i] = v;
  }
}

extern "C" int mt_downsample_seg_nearest(const float* src, int NC, int Di, int Hi, int Wi, float* dst, int Do, int Ho, int Wo,
                                         int remove_minus_one, mt_stream_t stream) {
  MT_REQUIRE(src != nullptr && dst != nullptr, "downsample_seg: null pointers");
  MT_REQUIRE(NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "downsample_seg: empty problem");
  DsParams P;
  P.src = src; P.dst = dst; P.NC = NC; P.Di = Di; P.Hi = Hi; P.Wi = Wi; P.Do = Do; P.Ho = Ho; P.Wo = Wo;
  P.fd = (double)Di / Do; P.fh = (double)Hi / Ho; P.fw = (double)Wi / Wo;
  P.remove_minus_one = remove_minus_one;
  const long total = (long)NC * Do * Ho * Wo;
  int blocks = mt_cdiv(total, 256); if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(downsample_seg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("downsample_seg");
  return MT_OK;
}

// ================================================================================================
// Spatial augmentation on the device (second half): batchgenerators' SpatialTransform as nnU-Net configures it
// (data_augmentation_moreDA.py:66-80: rotation + scaling, no elastic deformation, order_data 3 / order_seg 1, constant borders)
// is, per sample, scipy.ndimage.map_coordinates over an AFFINE coordinate field x = M (o - (O-1)/2) + centre.  Two kernels:
//  * mt_spline_prefilter3: the cubic B-spline prefilter map_coordinates(order=3) applies first — per axis the recursive filter with
//    pole z = sqrt(3) - 2, gain 6 and scipy's exact mirror boundary initialisation (mode 'constant' filters with mirror
//    boundaries); one thread per line, double-precision recursion, in place, three passes (W, H, D);
//  * mt_affine_sample: output voxel -> coordinate -> 64-tap cubic B-spline (order 3), 8-tap linear (order 1), nearest (order 0) or
//    the per-label rule batchgenerators uses for segmentations with order 1 (label c wins where the order-1 interpolation of the
//    indicator (seg == c) is >= 0.5, later = larger labels override).  scipy's 'constant' mode: a coordinate outside [0, n-1]
//    on any axis yields cval; inside, stencil taps that fall off the array read the MIRRORED coefficient.
struct PrefilterParams { float* vol; long nlines; int len; long line_stride_outer, line_stride_inner; int inner; long elem_stride; };
// line index l -> base offset: (l / inner) * line_stride_outer + (l % inner) * line_stride_inner; elements at + k * elem_stride
__global__ __launch_bounds__(256) void spline_prefilter_kernel(const PrefilterParams P) {
  const double z = -0.26794919243112270647;       // sqrt(3) - 2
  const int n = P.len;
  for (long l = (long)blockIdx.x * 256 + threadIdx.x; l < P.nlines; l += (long)gridDim.x * 256) {
    float* p = P.vol + (l / P.inner) * P.line_stride_outer + (l % P.inner) * P.line_stride_inner;
    const long es = P.elem_stride;
    if (n == 1) continue;
    // causal initialisation, exact for mirror boundaries: sum_{k=0}^{2n-3} z^k s_mirror[k] / (1 - z^(2n-2)), gain 6 folded in
    const double zn = pow(z, (double)(n - 1));
    double sum = 6.0 * ((double)p[0] + zn * (double)p[(long)(n - 1) * es]);
    double z1 = z, z2 = zn * zn / z;
    for (int k = 1; k < n - 1; ++k) { sum += 6.0 * (z1 + z2) * (double)p[(long)k * es]; z1 *= z; z2 /= z; }
    double c = sum / (1.0 - zn * zn);
    p[0] = (float)c;                                // NB: float storage between the passes, double recursion within a pass
    double prev = c;
    // to keep the recursion in double the causal pass is redone from the float inputs: c[k] = 6 s[k] + z c[k-1]
    for (int k = 1; k < n; ++k) { prev = 6.0 * (double)p[(long)k * es] + z * prev; p[(long)k * es] = (float)prev; }
    // anticausal: c[n-1] = z/(z^2-1) (z c[n-2] + c[n-1]);  c[k] = z (c[k+1] - c[k])
    double last = (z / (z * z - 1.0)) * (z * (double)p[(long)(n - 2) * es] + prev);
    p[(long)(n - 1) * es] = (float)last;
    for (int k = n - 2; k >= 0; --k) { last = z * (last - (double)p[(long)k * es]); p[(long)k * es] = (float)last; }
  }
}
extern "C" int mt_spline_prefilter3(float* vol, int NC, int D, int H, int W, int axes, mt_stream_t stream) {
  MT_REQUIRE(vol != nullptr && NC > 0 && D > 0 && H > 0 && W > 0 && axes > 0 && axes < 8, "spline_prefilter3: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  PrefilterParams P; P.vol = vol;
  const long HW = (long)H * W, V = (long)D * HW;
  // along W: lines (nc, d, h)
  P.nlines = (long)NC * D * H; P.len = W; P.inner = 1; P.line_stride_outer = W; P.line_stride_inner = 0; P.elem_stride = 1;
  if (axes & 1) hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)mt_cdiv(P.nlines, 256)), dim3(256), 0, st, P);
  // along H: lines (nc*d, w)
  P.nlines = (long)NC * D * W; P.len = H; P.inner = W; P.line_stride_outer = HW; P.line_stride_inner = 1; P.elem_stride = W;
  if (axes & 2) hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)mt_cdiv(P.nlines, 256)), dim3(256), 0, st, P);
  // along D: lines (nc, h*w)
  P.nlines = (long)NC * HW; P.len = D; P.inner = (int)HW; P.line_stride_outer = V; P.line_stride_inner = 1; P.elem_stride = HW;
  if (axes & 4) hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)mt_cdiv(P.nlines, 256)), dim3(256), 0, st, P);
  MT_CHECK_LAUNCH("spline_prefilter3");
  return MT_OK;
}

struct AffineParams {
  const float* src; float* dst; const float* mats;     // mats: per sample 12 floats = 3x3 matrix M (row-major) + centre (3)
  int planar;                                           // 1: "dummy 2D" — D is a stack of independent slices (xd = od), M acts on (h, w) only
  int N, C, D, H, W, OD, OH, OW, mode;                  // mode 0 nearest, 1 linear, 3 cubic (src = prefiltered coefficients), 11 per-label linear
  float cval;
};
__device__ __forceinline__ int mt_mirror(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i < 0 ? -i : i;
  i %= period;
  return i > n - 1 ? period - i : i;
}
__global__ __launch_bounds__(256) void affine_sample_kernel(const AffineParams P) {
  const long ovol = (long)P.OD * P.OH * P.OW, total = (long)P.N * ovol;
  const size_t V = (size_t)P.D * P.H * P.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / ovol);
    long r = i % ovol;
    const int ow = (int)(r % P.OW), oh = (int)((r / P.OW) % P.OH), od = (int)(r / ((long)P.OW * P.OH));
    const float* M = P.mats + n * 12;
    const double gd = od - 0.5 * (P.OD - 1), gh = oh - 0.5 * (P.OH - 1), gw = ow - 0.5 * (P.OW - 1);
    const double xd = P.planar ? (double)od : (double)M[0] * gd + (double)M[1] * gh + (double)M[2] * gw + (double)M[9];
    const double xh = (P.planar ? 0.0 : (double)M[3] * gd) + (double)M[4] * gh + (double)M[5] * gw + (double)M[10];
    const double xw = (P.planar ? 0.0 : (double)M[6] * gd) + (double)M[7] * gh + (double)M[8] * gw + (double)M[11];
    const bool inside = xd >= 0.0 && xd <= (double)(P.D - 1) && xh >= 0.0 && xh <= (double)(P.H - 1) && xw >= 0.0 && xw <= (double)(P.W - 1);
    const float* sn = P.src + (size_t)n * P.C * V;
    float* dn = P.dst + (size_t)n * P.C * ovol + r;
    if (!inside) {
      for (int c = 0; c < P.C; ++c) dn[(size_t)c * ovol] = (P.mode == 11) ? 0.f : P.cval;      // per-label rule: nothing reaches 0.5
      continue;
    }
    if (P.mode == 0) {
      const int a = (int)floor(xd + 0.5), b = (int)floor(xh + 0.5), e = (int)floor(xw + 0.5);
      const size_t o = ((size_t)(a > P.D - 1 ? P.D - 1 : a) * P.H + (b > P.H - 1 ? P.H - 1 : b)) * P.W + (e > P.W - 1 ? P.W - 1 : e);
      for (int c = 0; c < P.C; ++c) dn[(size_t)c * ovol] = sn[(size_t)c * V + o];
      continue;
    }
    if (P.mode == 1 || P.mode == 11) {
      const int d0 = (int)floor(xd), h0 = (int)floor(xh), w0 = (int)floor(xw);
      const float fd = (float)(xd - d0), fh = (float)(xh - h0), fw = (float)(xw - w0);
      const int d1 = mt_mirror(d0 + 1, P.D), h1 = mt_mirror(h0 + 1, P.H), w1 = mt_mirror(w0 + 1, P.W);
      size_t off[8]; float wt[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int a = (k & 4) ? d1 : d0, b = (k & 2) ? h1 : h0, e = (k & 1) ? w1 : w0;
        off[k] = ((size_t)a * P.H + b) * P.W + e;
        wt[k] = ((k & 4) ? fd : 1.f - fd) * ((k & 2) ? fh : 1.f - fh) * ((k & 1) ? fw : 1.f - fw);
      }
      for (int c = 0; c < P.C; ++c) {
        const float* s = sn + (size_t)c * V;
        if (P.mode == 1) {
          float v = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) v += wt[k] * s[off[k]];
          dn[(size_t)c * ovol] = v;
        } else {                       // per-label: the largest label whose indicator interpolates to >= 0.5 (ascending override)
          float lab[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) lab[k] = s[off[k]];
          float best = 0.f; bool any = false;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (lab[j] == lab[k]) ? wt[j] : 0.f;
            if (sum >= 0.5f && (!any || lab[k] > best)) { best = lab[k]; any = true; }
          }
          dn[(size_t)c * ovol] = any ? best : 0.f;
        }
      }
      continue;
    }
    // cubic B-spline on prefiltered coefficients
    int id[4], ih[4], iw[4]; float wd[4], wh[4], ww[4];
    {
      const double x[3] = {xd, xh, xw};
      const int dims[3] = {P.D, P.H, P.W};
      int* idx[3] = {id, ih, iw}; float* wgt[3] = {wd, wh, ww};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int i0 = (int)floor(x[a]) - 1;
        const double t = x[a] - floor(x[a]);
        wgt[a][0] = (float)((1.0 - t) * (1.0 - t) * (1.0 - t) / 6.0);
        wgt[a][1] = (float)((4.0 - 6.0 * t * t + 3.0 * t * t * t) / 6.0);
        wgt[a][2] = (float)((1.0 + 3.0 * t + 3.0 * t * t - 3.0 * t * t * t) / 6.0);
        wgt[a][3] = (float)(t * t * t / 6.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) idx[a][k] = mt_mirror(i0 + k, dims[a]);
      }
    }
    if (P.planar) { wd[0] = 0.f; wd[1] = 1.f; wd[2] = 0.f; wd[3] = 0.f; id[1] = od; }      // slices are independent images: no D stencil
    for (int c = 0; c < P.C; ++c) {
      const float* s = sn + (size_t)c * V;
      float v = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float va = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float* row = s + ((size_t)id[a] * P.H + ih[b]) * P.W;
          va += wh[b] * (ww[0] * row[iw[0]] + ww[1] * row[iw[1]] + ww[2] * row[iw[2]] + ww[3] * row[iw[3]]);
        }
        v += wd[a] * va;
      }
      dn[(size_t)c * ovol] = v;
    }
  }
}
extern "C" int mt_affine_sample(const float* src, int N, int C, int D, int H, int W, float* dst, int OD, int OH, int OW,
                                const float* mats, int mode, float cval, int planar, mt_stream_t stream) {
  MT_REQUIRE(!planar || OD == D, "affine_sample: planar sampling keeps the slice axis (OD == D)");
  MT_REQUIRE(src && dst && mats && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && OD > 0 && OH > 0 && OW > 0, "affine_sample: bad arguments");
  MT_REQUIRE(mode == 0 || mode == 1 || mode == 3 || mode == 11, "affine_sample: mode must be 0, 1, 3 or 11 (got %d)", mode);
  AffineParams P;
  P.src = src; P.dst = dst; P.mats = mats; P.N = N; P.C = C; P.D = D; P.H = H; P.W = W; P.OD = OD; P.OH = OH; P.OW = OW;
  P.mode = mode; P.cval = cval; P.planar = planar ? 1 : 0;
  const long total = (long)N * OD * OH * OW;
  int blocks = mt_cdiv(total, 256); if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(affine_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("affine_sample");
  return MT_OK;
}

// ================================================================================================
// GaussianBlurTransform of nnU-Net's moreDA chain (data_augmentation_moreDA.py:87-88; batchgenerators' augment_gaussian_blur =
// scipy.ndimage.gaussian_filter(channel, sigma, order=0) per chosen channel): separable, per axis a correlation with
// w[k] = exp(-k^2 / (2 sigma^2)) / sum, k = -r..r, r = int(4 sigma + 0.5) (truncate = 4) and scipy's default 'reflect' boundary
// (half-sample symmetric: d c b a | a b c d | d c b a).  One launch per axis; sigma per (sample, channel) — sigma <= 0 copies the
// channel (the transform blurs only some channels of some samples).  HBM-bound: one read (the 2r+1 taps of neighbouring threads
// overlap in L1/L2) and one write per element and pass.
struct BlurParams { const float* src; float* dst; const float* sigma; int NC, D, H, W, axis; };
#define MT_BLUR_MAXR 32
__global__ __launch_bounds__(256) void gaussian_blur_axis_kernel(const BlurParams P) {
  __shared__ float wsh[2 * MT_BLUR_MAXR + 1];
  const int nc = blockIdx.y;
  const float sg = P.sigma[nc];
  const long V = (long)P.D * P.H * P.W;
  const float* src = P.src + (size_t)nc * V;
  float* dst = P.dst + (size_t)nc * V;
  int r = 0;
  if (sg > 0.f) {
    r = (int)(4.0 * (double)sg + 0.5);
    if (r > MT_BLUR_MAXR) r = MT_BLUR_MAXR;
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int k = -r; k <= r; ++k) s += exp(-0.5 * (double)k * k / ((double)sg * sg));
      for (int k = -r; k <= r; ++k) wsh[k + r] = (float)(exp(-0.5 * (double)k * k / ((double)sg * sg)) / s);
    }
  }
  __syncthreads();
  const int len = P.axis == 0 ? P.D : (P.axis == 1 ? P.H : P.W);
  const long stride = P.axis == 0 ? (long)P.H * P.W : (P.axis == 1 ? P.W : 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
    if (sg <= 0.f) { dst[i] = src[i]; continue; }
    const int pos = (int)((i / stride) % len);
    const long base = i - (long)pos * stride;
    double acc = 0.0;                                     // scipy's correlate1d accumulates in double
    for (int k = -r; k <= r; ++k) {
      int q = pos + k;
      // 'reflect': period 2 len, q -> q mod 2 len, mirrored in the upper half
      if (q < 0 || q >= len) {
        const int p2 = 2 * len;
        q %= p2; if (q < 0) q += p2;
        if (q >= len) q = p2 - 1 - q;
      }
      acc += (double)wsh[k + r] * (double)src[base + (long)q * stride];
    }
    dst[i] = (float)acc;
  }
}

extern "C" int mt_gaussian_blur_axis(const float* src, float* dst, int NC, int D, int H, int W, int axis, const float* sigma, mt_stream_t stream) {
  MT_REQUIRE(src != nullptr && dst != nullptr && sigma != nullptr && src != dst, "gaussian_blur_axis: null or aliased pointers");
  MT_REQUIRE(NC > 0 && NC <= 65535 && D > 0 && H > 0 && W > 0 && axis >= 0 && axis <= 2, "gaussian_blur_axis: bad geometry");
  BlurParams P;
  P.src = src; P.dst = dst; P.sigma = sigma; P.NC = NC; P.D = D; P.H = H; P.W = W; P.axis = axis;
  const long V = (long)D * H * W;
  int blocks = mt_cdiv(V, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gaussian_blur_axis_kernel, dim3(blocks, NC), dim3(256), 0, (hipStream_t)stream, P);
  MT_CHECK_LAUNCH("gaussian_blur_axis");
  return MT_OK;
}
