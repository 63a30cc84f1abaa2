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
