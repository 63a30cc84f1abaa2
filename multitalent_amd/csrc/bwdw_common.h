// Parameter blocks shared by the convolution translation units (conv_lds.hip plans, dispatches and reduces; bwdw_tr16.hip holds one
// kernel family of the mixed-precision mode).
#pragma once
#include "mt_common.h"

struct ConvChunk { short src, c0, ck, cglob; };

struct ConvKParams {
  mt_conv3d_t c;
  int tilesD, tilesH, tilesW, nsb;
  int nchunks, ntaps;
  int dbg;       // 0 (timing ablations of conv_fwd_kernel when set by hand: 1 skip staging, 2 skip weight loads, 4 skip epilogue, 8 skip MFMA, 16 stamps)
  int stagger;   // 0 (one-time start delay per residency slot of the first block wave: measured without effect, rounds 3 and 6)
  ConvChunk chunk[MT_MAX_CHUNKS];
};

#define BW_CK 16
struct BwdWParams {
  mt_conv3d_t c;      // X geometry (src), conv geometry; Do/Ho/Wo = Y dims
  mt_src_t y;         // Y source (C = Cout)
  int TD, TH, TW;     // spatial tile (TW % 4 == 0)
  int tilesD, tilesH, tilesW, ntiles_total;
  int nchunks, ntaps, ncot, nsg;
  int nsg_cap, nunits, nseg, dseg;   // marching kernel: units = (sample, h-tile, w-tile, D segment of dseg planes)
  int cw;             // conv_bwdw_fast_kernel: cout tiles per workgroup (1 | 2 | 4; grid.y = ceil(ncot / cw))
  float* part;        // [chunk][cot][sg][tap][16][32]
  ConvChunk chunk[MT_MAX_CHUNKS];
};

typedef __bf16 bwb_bf16x8 __attribute__((ext_vector_type(8)));

// bwdw_tr16.hip: direct bf16 backward-weight of 3x3x3 (KD = 3) / 1x3x3 (KD = 1) stride-1 convolutions fed by LDS transpose reads
int mt_launch_bwdw_tr16(const BwdWParams& P, int KD, int xdt, hipStream_t st);

// conv_x16.hip: persistent stride-1 3x3x3 (KD = 3) / 1x3x3 (KD = 1) convolution with ONE 16-bit storage type on all operands (fp16 forward,
// bf16 backward-data); weights in pack layout 4 / 3; items = (spatial 4x4x32 tile, 32-channel cout tile), nwg persistent workgroups
struct X16Params {
  mt_conv3d_t c;
  int tilesD, tilesH, tilesW, nsb;
  int nchunks, ncot, nitems, nwg;
  int npairs;                         // chunk pairs: two consecutive 16-channel chunks of one source (64 bytes of a voxel), or a single chunk (-1)
  short pair[MT_MAX_CHUNKS][2];
  ConvChunk chunk[MT_MAX_CHUNKS];
};
int mt_conv_x16_workgroups(int nitems);
int mt_launch_conv_x16(const X16Params& P, int KD, int dt, hipStream_t st);
