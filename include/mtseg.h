/*
 * mtseg.h — C ABI of libmtseg_hip.so, the MI355X (gfx950) hot path of the 3D patch segmentation engine.
 *
 * The reference (MIC-DKFZ/MultiTalent, nnU-Net v1) has no FFI: its plugin surface is Python
 * (SURVEY.md §8b).  This ABI sits UNDER that surface: the Python modules that mirror
 * `Generic_UNet` / `FabiansUNet` / the trainers call these entry points through ctypes with raw
 * device pointers owned by the caller (torch), a hipStream_t and plain sizes.  No torch types,
 * no allocation, no synchronisation inside (mt_probe_device excepted, see there); every function returns 0
 * on success or a negative MT_E* code (text via mt_last_error(), thread-local).  Threading: launches are
 * re-entrant per stream and per device; since ABI 4 there is no mutable process-wide state (kernel selection is a field of the
 * problem struct, see "State" below) - only per-device caches of immutable properties and one-time kernel attributes.
 *
 * Layout: activations are NDHWC ("channels last"), fp32 or — in the mixed-precision mode, for the tensors the caller chooses —
 * bf16 (`dtype` = MT_BF16: the pointer then addresses 2-byte elements although it is typed `float*`), possibly a channel slice of a
 * wider buffer (channel stride `cs` in ELEMENTS, first channel folded into the pointer).  A "lazy activation" is a
 * raw conv output y plus per-(n,c) scale/shift and a LeakyReLU slope:
 *     a = lrelu_slope(y*scale + shift)         (InstanceNorm + LeakyReLU applied on load)
 * which is how `ConvDropoutNormNonlin.forward` (generic_UNet.py:66-70) is fused away.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef MTSEG_H
#define MTSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MT_OK 0
#define MT_EINVAL (-1)  /* bad argument / unsupported shape */
#define MT_EWORKSPACE (-2) /* workspace too small */
#define MT_EHIP (-3)    /* HIP runtime error on launch */
#define MT_EUNSUPPORTED (-4) /* the device is not the one this library is built for (gfx950) */

#define MT_ABI_VERSION 4   /* 2: storage dtypes (mt_src_t.dtype, odtype fields, dtype arguments of the streaming kernels, mt_cast); 3: mt_pointwise_t.mma + mt_pointwise_pack_layout;
                            * 4: kernel selection is a FIELD of the problem (mt_conv3d_t.select / max_workgroups): mt_set_option and every MT_* environment
                            *    switch of the library are gone — no process-wide state besides per-device caches of immutable properties */
#define MT_MAX_CHUNKS 64

/* Storage type of an activation / gradient tensor in HBM.  fp32 is the parity path.  The mixed-precision mode (the reference's
 * autocast keeps conv inputs / outputs in half precision: MultiTalent_Trainer_DDP.py:340-354, network_trainer.py:400-402) stores
 * ACTIVATIONS as fp16 — the reference's own type: 11 significand bits, and the forward convolutions then multiply fp16 operands
 * (v_mfma_f32_32x32x16_f16) — and GRADIENTS as bf16 (fp32's exponent range: no loss scaling; the backward convolutions multiply
 * bf16 operands).  Values are rounded to nearest-even when stored and widened exactly when loaded; all arithmetic in between
 * (normalisation, statistics, accumulation, loss, optimizer) is fp32.  Not every kernel takes every combination: ask the
 * *_io_supported queries and convert with mt_cast where the answer is 0. */
#define MT_F32 0
#define MT_BF16 1
#define MT_F16 2

typedef void* mt_stream_t; /* hipStream_t */

/* One input source of a convolution: channel slice [0,C) of an NDHWC buffer with channel stride cs.
 * scale/shift: [N][C] per-sample affine applied on load (NULL = none); slope: LeakyReLU slope
 * applied after the affine (1.0f = identity). */
typedef struct {
  const float* ptr;
  int32_t cs;
  int32_t C;
  const float* scale;
  const float* shift;
  float slope;
  int32_t dtype;  /* MT_F32 | MT_BF16 | MT_F16: element type behind `ptr` (cs counts elements) */
} mt_src_t;

/* Fused statistics for the InstanceNorm + LeakyReLU backward that runs next (generic_UNet.py:63-64 in reverse).  A convolution that
 * is the LAST writer of a gradient tensor g = dL/d lrelu(IN(y)) (the backward-data of the layer's consumer) can emit, per block,
 * the two sums the normalisation backward needs over g,
 *     A = sum dz,   B = sum dz * zhat,      dz = g * lrelu'(z),  zhat = (y - mean) * rstd,  z = zhat * gamma + beta,
 * for its output channels [c0, c0 + C) (one of its two destinations) — the separate reduction pass over (g, y) disappears
 * (mt_inorm_lrelu_bwd with `part`).  The partials go to stats_part ([N][nsb][Cout][2], slot (.., c0 + c, 0 / 1) = A / B; channels
 * outside the range hold 0).  Supported by the kernels mt_conv3d_bwd_stats_supported() says yes to. */
typedef struct {
  const float* y;        /* raw forward output of the normalised layer [N, Do, Ho, Wo, ycs]; NULL = off */
  const float* mean;     /* [N][C] */
  const float* rstd;     /* [N][C] */
  const float* gamma;    /* [C] or NULL (1) */
  const float* beta;     /* [C] or NULL (0) */
  int32_t ycs, c0, C;
  float slope;
} mt_bwd_stats_t;

/* Convolution problem.  Forward conv (nn.Conv3d, generic_UNet.py:57,67; conv_blocks.py:49-85,116-213):
 *   out[n,o,co] = bias[co] + sum_{t,ci} in[n, o*S + t - P, ci] * W[t][ci][co]
 * with the input being the channel concatenation of src[0..nsrc) (torch.cat((x, skip), 1),
 * generic_UNet.py:392).  `dil` inserts zeros between stored input samples (virtual input coordinate
 * u maps to stored u/dil when divisible) — this is how the backward-data of a strided conv and
 * nothing else is expressed.  Weights are pre-packed by mt_pack_conv_weights. */
typedef struct {
  mt_src_t src[2];
  int32_t nsrc;
  int32_t N, Di, Hi, Wi;      /* stored input dims */
  int32_t dilD, dilH, dilW;   /* zero-insertion factor of the virtual input (1 or 2) */
  int32_t Do, Ho, Wo;         /* output dims */
  int32_t KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int32_t Cin, Cout;
  const float* wpack;         /* packed weights (mt_pack_conv_weights, ck = MT_CONV_CK) */
  const float* bias;          /* [Cout] or NULL */
  float* out0; int32_t ocs0;  /* channels [0,csplit) -> out0[... * ocs0 + co] */
  float* out1; int32_t ocs1;  /* channels [csplit,Cout) -> out1[... * ocs1 + (co - csplit)] (may be NULL if csplit>=Cout) */
  int32_t csplit;
  int32_t accumulate;         /* 1: out += result (gradient accumulation on skip connections) */
  float* stats_part;          /* NULL or [N][nsb][Cout][2] per-block (sum, sumsq) partials, nsb = mt_conv3d_stats_blocks() */
  /* Strided output placement (0 = dense): logical output position o is written at o*os + oo of a tensor with spatial dims
   * (OD,OH,OW).  Used by the backward-data of a strided conv, which is computed as one exact stride-1 convolution per
   * parity class of the input position instead of a convolution over a zero-inserted gradient. */
  int32_t OD, OH, OW, osD, osH, osW, ooD, ooH, ooW;
  int32_t mma;                /* matrix input type of the convolution: 0 fp32 (exact), 1 bf16 inputs with fp32 accumulation
                                 (mixed precision, the reference's autocast mode); packed weights must match (mt_conv3d_pack_layout) */
  int32_t odtype;             /* MT_F32 | MT_BF16 | MT_F16: element type of out0 / out1 (ocs0 / ocs1 count elements); statistics are taken from
                                 the values as stored */
  mt_bwd_stats_t bstats;      /* bstats.y != NULL: fused first pass of the NEXT InstanceNorm backward (see mt_bwd_stats_t) */
  uint32_t select;            /* ABI 4: kernel selection of THIS problem, 2-bit fields MT_SEL_* (0 = the library's policy everywhere).  The queries
                                 (mt_conv3d_ck, _pack_layout, _stats_blocks, _kernel_name, *_io_supported, *_workspace) and the launch must see the same value. */
  int32_t max_workgroups;     /* ABI 4: > 0 caps the workers of the persistent kernels (conv_wino8p: per output-channel tile; conv_x16, conv_bwdw_tr16: total)
                                 — tests make a worker walk several tiles of a small problem; 0 = fill the chip */
} mt_conv3d_t;

/* mt_conv3d_t.select: which kernel FAMILIES may serve the problem.  All families compute the same convolution (fp32: to rounding of the
 * summation order; mma == 1: the 16-bit matrix kernels round the operands); the policy (MT_SEL_DEFAULT) takes a specialised kernel where
 * its grid fills the chip.  Tests force a family on small shapes (MT_SEL_FORCE) or exclude it (MT_SEL_OFF) to compare two kernels on the
 * same problem; the A/B tools do the same over whole steps. */
#define MT_SEL_DEFAULT 0u
#define MT_SEL_OFF 1u
#define MT_SEL_FORCE 2u
#define MT_SEL_WINO 0        /* shift: Winograd F(2x2x2, 3x3x3) forward / backward-data (conv_wino8p_kernel) */
#define MT_SEL_M16 2         /* 16-bit matrix kernels of mma == 1 problems (conv_bf16_kernel, conv_x16_kernel; OFF = the fp32 kernels) */
#define MT_SEL_X16 4         /* conv_x16_kernel among them (DEFAULT: where it measured faster; FORCE: wherever eligible) */
#define MT_SEL_TAPSPLIT 6    /* conv_tapsplit_kernel on under-filled grids (FORCE: the strided form also on well-filled grids) */
#define MT_SEL_BWDW_WINO 8   /* F(3x3, 2x2) backward-weight kernels (3x3x3 and 1x3x3) */
#define MT_SEL_BWDW_TR16 10  /* conv_bwdw_tr16_kernel (16-bit X, bf16 dY) */
#define MT_SEL_BWDW_CW 12    /* cout tiles per workgroup of the tiled backward-weight kernels: 0 = up to 4 where a workgroup walks enough tiles,
                                1 = one, 2 = at most two, 3 = up to 4 also on small problems */
#define MT_SEL_GET(sel, shift) (((sel) >> (shift)) & 3u)

const char* mt_last_error(void);
int mt_abi_version(void);

/* ---- weight packing ------------------------------------------------------------------------
 * Packs W_eff[tap][ci][co] = w[ci*s_ci + co*s_co + kd'*s_kd + kh'*s_kh + kw'*s_kw] (k' = K-1-k when
 * flip) into the MFMA B-fragment order [ntile][chunk][tap][ck/2][64] used by the conv kernels, where
 * the input channels are split into chunks of at most `ck` channels that never straddle the two
 * sources (C0 | C1).  Returns number of floats via *packed_floats (query with dst == NULL). */
int mt_pack_conv_weights(const float* w, float* dst, size_t* packed_floats,
                         int C0, int C1, int Cout, int KD, int KH, int KW,
                         long s_ci, long s_co, long s_kd, long s_kh, long s_kw, int flip, int ck,
                         int layout /* 1: all MFMA kernels ([kp/4][lane][4], pair (kp, ck/2+kp)); mt_pointwise_fwd takes ck = 16;
                                       0: legacy [kp][lane], pair (2kp,2kp+1); 2: Winograd F(2x2x2,3x3x3) U = G g G^T (ck = 8);
                                       3: bf16 B fragments of v_mfma_f32_32x32x16_bf16, [tap][lane][8 bf16] (ck = 16), RNE */,
                         const int32_t* tapmap /* NULL, or {tbD,tsD,tbH,tsH,tbW,tsW}: packed tap j of dim d takes source
                                                  tap tb + ts*j (overrides flip) — sub-kernels of the parity classes */,
                         mt_stream_t stream);

/* ---- convolution (N1/N2/N5 forward, and backward-data through flipped weights) --------------- */
int mt_conv3d_fwd(const mt_conv3d_t* p, mt_stream_t stream);
int mt_conv3d_stats_blocks(const mt_conv3d_t* p); /* spatial blocks per sample (size of stats_part dim 1) */
int mt_conv3d_ck(const mt_conv3d_t* p);           /* channel chunk the kernel will use (pack weights with it) */
/* Batched packing: all layers of one optimizer step in one launch.  mt_pack_desc_fill writes one opaque descriptor
 * (mt_pack_desc_size() bytes, same arguments as mt_pack_conv_weights) into HOST memory; the caller uploads the table once and
 * calls mt_pack_batched(table_on_device, n) every step — the descriptors stay valid while the weight and destination buffers
 * do. */
size_t mt_pack_desc_size(void);
int mt_pack_desc_fill(void* desc, const float* w, float* dst, int C0, int C1, int Cout, int KD, int KH, int KW, long s_ci,
                      long s_co, long s_kd, long s_kh, long s_kw, int flip, int ck, int layout, const int32_t* tapmap);
int mt_pack_batched(const void* descs_device, int n, mt_stream_t stream);

/* Backward-data of a strided 3x3x3 convolution (pad 1, stride (2,2,2) or (1,2,2)) in one launch — replaces autograd's
 * conv_transpose for the strided stage convs (generic_UNet.py:263-278 `first_stride`, generic_modular_UNet.py:69-77).
 * p carries the FORWARD geometry: Di/Hi/Wi = X dims, Do/Ho/Wo = Y dims, K, S, P, Cin, Cout; src[0] = dY (C = Cout, may be lazy);
 * out0/ocs0 = dX (Cin channels), accumulate adds to it.  wpack = mt_pack_conv_weights(w, C0 = Cout, C1 = 0, Cout_arg = Cin,
 * kernel 3x3x3, strides with the ci/co roles swapped, flip = 0, ck = 16, layout 1).  Each of the 27 taps is one MFMA block
 * into the accumulator of the parity class (x mod stride) it reaches: no work on inserted zeros. */
int mt_conv3d_bwd_data_strided(const mt_conv3d_t* p, mt_stream_t stream);
int mt_conv3d_bwd_data_strided_supported(const mt_conv3d_t* p);   /* 1 when the geometry is handled, else 0 */
int mt_conv3d_bwd_data_strided_pack_layout(const mt_conv3d_t* p); /* `layout` of its packed weights: 1, or 3 (bf16) when p->mma == 1 */
/* Storage types: 1 when the kernel that serves p takes p->src[*].dtype / p->odtype natively (all-fp32: always).  0: convert the
 * operands with mt_cast (the launch itself refuses with MT_EINVAL).  Same question for the other convolution entry points. */
int mt_conv3d_io_supported(const mt_conv3d_t* p);
int mt_conv3d_bwd_data_strided_io_supported(const mt_conv3d_t* p);
int mt_conv3d_bwd_weight_io_supported(const mt_conv3d_t* p, const mt_src_t* ysrc);

/* State.  The library keeps NO mutable process-wide state (ABI 4): which kernel serves a problem is a function of the problem struct
 * alone (geometry, storage types, mma, select, max_workgroups), launches are re-entrant across host threads and streams, and memory
 * (outputs, workspaces, statistics partials) is caller-owned.  What it caches is immutable per DEVICE: a kernel's raised dynamic-LDS
 * limit and the CU count that sizes persistent grids, keyed by the current HIP device (one process may drive several GPUs).
 * Rounds 1 - 5 had 39 MT_* environment switches and mt_set_option in here; the A/Bs they served are closed, the losing
 * kernels are deleted or unreachable, and the families tests still compare are the MT_SEL_* fields above.
 * The HOST side above this ABI (multitalent_amd/engine.py, ops.py, inference/, bench.py) has its own knobs: MT_BF16_STORAGE (0: fp32
 * storage in mixed precision), MT_ACT_STORAGE (fp16 | bf16), MT_BF16_MIN_VOXELS, MT_BWDW_STREAMS, MT_FUSED_LOSS, MT_STEP_GRAPH, MT_IO_DEBUG
 * (1: print every launch that needed an mt_cast), MT_SELECT ("x16=off,wino=force,...": default mt_conv3d_t.select of the process, for A/B
 * runs of whole steps), MT_FORCE_REDUCER, MT_BENCH_ONE_GPU (bench.py: all ranks on cuda:0 over gloo), MT_LIB_VARIANT.
 * Which workgroup computes which tile (block id -> XCD -> tile order) is a compile-time choice (-DMT_TILE_ORDER=0 builds
 * the round-2 order for A/B measurements); results do not depend on it. */

/* Device probe (SYNCHRONOUS, call once per device before the first launch; the Python binding does so when it loads the
 * library): checks that the current device is gfx950 and that raw buffer loads behave the way the vector-load kernels assume —
 * a dword-aligned buffer_load_dwordx4 returns its four dwords, and a load that straddles num_records returns the in-range dwords
 * and zeros for the rest.  `scratch`: >= 32 KiB of device memory owned by the caller.  *vector_loads_ok = 1 / 0; `arch` receives
 * the gcnArchName.  Returns MT_EUNSUPPORTED on a non-gfx950 device. */
int mt_probe_device(void* scratch, size_t scratch_bytes, int* vector_loads_ok, char* arch, size_t arch_len, mt_stream_t stream);
int mt_conv3d_pack_layout(const mt_conv3d_t* p);   /* `layout` for mt_pack_conv_weights: 1; 2 when the Winograd kernel serves p; 3 (bf16) when p->mma == 1 and the bf16 kernel does */
int mt_conv3d_kernel_name(const mt_conv3d_t* p, char* buf, size_t n); /* device kernel that will run (profiler name) */
int mt_conv3d_bwd_stats_supported(const mt_conv3d_t* p); /* 1 when the kernel that serves p (geometry, ignoring p->bstats) honours p->bstats */
/* the same for mt_conv3d_bwd_weight(p, ysrc, ...) and mt_conv3d_bwd_data_strided(p): which kernel family the dispatcher picks
 * (tests assert that full-size problems run on the Winograd / strided / stem kernels; bench.py groups its timings by it) */
int mt_conv3d_bwd_weight_kernel_name(const mt_conv3d_t* p, const mt_src_t* ysrc, char* buf, size_t n);
int mt_conv3d_bwd_data_strided_kernel_name(const mt_conv3d_t* p, char* buf, size_t n);

/* ---- backward-weight -------------------------------------------------------------------------
 * dW[tap][ci][co] = sum_{n,o} X[n, o*S + t - P, ci] * Y[n, o, co]   (autograd of nn.Conv3d /
 * nn.ConvTranspose3d weights).  X is described by p->src (lazy activations allowed), Y by ysrc
 * (N,Do,Ho,Wo from p).  Result is written (or accumulated) into dw with the given element strides,
 * i.e. directly in the torch parameter layout.  Workspace: query mt_conv3d_bwd_weight_workspace. */
size_t mt_conv3d_bwd_weight_workspace(const mt_conv3d_t* p);
int mt_conv3d_bwd_weight(const mt_conv3d_t* p, const mt_src_t* ysrc, float* dw,
                         long s_ci, long s_co, long s_kd, long s_kh, long s_kw, int accumulate,
                         void* workspace, size_t workspace_bytes, mt_stream_t stream);

/* ---- pointwise / transposed convolution -----------------------------------------------------
 * Base grid [N][Db][Hb][Wb].  in voxel = base*si (gather), out voxel = base*so + tap (scatter, taps
 * = prod(so)).  si=so=1: 1x1x1 conv (seg heads generic_UNet.py:349-351; backward-data with packed
 * transposed weights).  si=2: strided 1x1x1 skip projection (conv_blocks.py:192-197).
 * so=pool kernel: nn.ConvTranspose3d with kernel==stride, bias=False (generic_UNet.py:335-336),
 * written straight into the concat buffer through ocs. */
typedef struct {
  mt_src_t src;
  int32_t N, Db, Hb, Wb;
  int32_t Di, Hi, Wi;          /* stored input dims (base*si must be inside) */
  int32_t siD, siH, siW;
  int32_t soD, soH, soW;
  int32_t Cin, Cout;
  const float* wpack;  /* mt_pack_conv_weights(C0=Cin, C1=0, taps = so, ck = Cin rounded up to even) */
  const float* bias;   /* [Cout] or NULL */
  float* out; int32_t ocs;
  int32_t accumulate;
  float* stats_part;   /* NULL or [N][nsb][Cout][2] */
  int32_t odtype;      /* MT_F32 | MT_BF16 | MT_F16: element type of `out` */
  int32_t scatter;     /* 0: all prod(so) taps (transposed convolution).  1: ONLY tap (0,0,0) — out[base*so] (+)= W x in[base], one packed
                          tap: the backward-data of a strided 1x1x1 convolution (conv_blocks.py:159-165); the other output positions are
                          not touched (the caller zero-fills or accumulates) */
  int32_t mma;         /* ABI v3.  0: fp32 products.  1 (mixed precision): fp16 products where the source is stored in fp16 and the kernel
                          that serves the problem has the 16-bit matrix form — the packed weights must then be in the layout
                          mt_pointwise_pack_layout(p) returns (4 instead of 1) */
} mt_pointwise_t;
int mt_pointwise_fwd(const mt_pointwise_t* p, mt_stream_t stream);
int mt_pointwise_stats_blocks(const mt_pointwise_t* p);
int mt_pointwise_io_supported(const mt_pointwise_t* p);
/* pack layout (mt_pack_conv_weights, ck = 16) of the weights this problem's launch reads: 1, or 4 (fp16 B fragments) when p->mma == 1
 * and the launch multiplies in fp16 (transposed convolutions and 33..64-channel heads over fp16 activations). */
int mt_pointwise_pack_layout(const mt_pointwise_t* p);   /* 1 when p->src.dtype / p->odtype are read / written natively (see MT_F16) */
/* Backward of a 1x1x1 segmentation head (generic_UNet.py:349-351, generic_modular_UNet.py:244,251: seg_outputs / deep_supervision_outputs)
 * in ONE pass over (x, dY):  dX[n,v,ci] (+)= sum_co dY[n,v,co] W[co,ci] (gradient w.r.t. the lazily ACTIVATED head input),
 * dW[co*s_co + ci*s_ci] (+)= sum_{n,v} act(x)[n,v,ci] dY[n,v,co], dbias[co] (+)= sum dY.  Cin, Cout <= 64; mt_head_bwd_supported says
 * where it beats the separate kernels (Cin <= 32: the full-resolution heads).
 * x: the head's input (lazy activation allowed); dy [N][V][dycs]; wpack_bwd = mt_pack_conv_weights(w, C0 = Cout, C1 = 0, Cout' = Cin,
 * 1x1x1, strides with ci/co swapped, flip 0, ck 16, layout 1) — the packing mt_pointwise_fwd takes for the same backward-data.
 * *dbias_done = 1 when dbias was produced (Cin not a multiple of 32: a spare MFMA row carries 1.0), else the caller sums dY
 * (mt_channel_sum).  ws: mt_head_bwd_workspace bytes (per-wave partials, summed in fp64 in a fixed order). */
int mt_head_bwd_supported(int Cin, int Cout);
size_t mt_head_bwd_workspace(int N, long V, int Cin, int Cout);
int mt_head_bwd(const mt_src_t* x, const float* dy, int dycs, int N, long V, int Cin, int Cout, const float* wpack_bwd,
                float* dx, int dxcs, int dxdtype /* storage type of dx: fp32, or bf16 with a 16-bit x */, int accumulate_dx, float* dw,
                long s_ci, long s_co, float* dbias, int accumulate_dw, int* dbias_done, void* ws, size_t ws_bytes, mt_stream_t stream);
int mt_head_bwd_io_supported(int xdtype, int xcs, int dxdtype, int dxcs, int Cin, int Cout);


/* ---- InstanceNorm3d(eps, affine) + LeakyReLU (generic_UNet.py:63-64,69-70) ------------------- */
/* partials [N][nsb][C][2] -> mean,rstd,scale,shift each [N][C]; scale = gamma*rstd, shift = beta - mean*scale */
int mt_inorm_finalize(const float* part, int N, int nsb, int C, double count, const float* gamma,
                      const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                      mt_stream_t stream);
/* materialise a = lrelu(y*scale+shift) (+ optional residual add BEFORE the lrelu: conv_blocks.py:201-213) */
int mt_inorm_lrelu_apply(const float* y, int ycs, const float* scale, const float* shift, float slope,
                         const float* res, int rcs, const float* rscale, const float* rshift, float rslope,
                         float* out, int ocs, int N, long V, int C, int dtype /* storage type of y, res and out */, mt_stream_t stream);
/* Backward of out = lrelu(IN(y)): given g = dL/dout (in place), produce dy in place, plus
 * dgamma[C] += , dbeta[C] +=, dbias[C] (= sum dy, may be NULL).  ws: mt_inorm_bwd_workspace bytes.
 * part != NULL: the first pass (sum dz, sum dz zhat per block) was fused into the convolution that produced g
 * (mt_conv3d_t.bstats): part = its stats_part [N][part_nblk][part_cs][2], this layer's channels at columns part_c0 ..; the
 * reduction pass over (g, y) is skipped.  part == NULL: computed here. */
size_t mt_inorm_bwd_workspace(int N, long V, int C);
int mt_inorm_lrelu_bwd(float* g, int gcs, const float* y, int ycs, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, float slope, int N, long V, int C,
                       float* dgamma, float* dbeta, float* dbias, const float* part, int part_nblk, int part_cs, int part_c0,
                       void* ws, size_t ws_bytes, int gdtype, int ydtype /* storage types of g and of y */, mt_stream_t stream);
/* g *= lrelu'(y*scale+shift) in place, optionally also writes a copy (residual branch gradient) */
int mt_lrelu_bwd(float* g, int gcs, const float* y, int ycs, const float* scale, const float* shift,
                 float slope, const float* y2, int y2cs, const float* scale2, const float* shift2,
                 float slope2, float* gcopy, int gcopycs, int N, long V, int C, int gdtype /* of g, gcopy */, int ydtype /* of y, y2 */,
                 mt_stream_t stream);
/* mt_lrelu_bwd on dense tensors (channel stride == C) that also emits the first pass of the NEXT InstanceNorm backward: in a residual
 * block out = lrelu(IN(y) + residual) (conv_blocks.py:201-213) the masked gradient g' is the gradient of IN(y), so
 * part[n][blk][c] = (sum g', sum g' * (y - mean) * rstd) over the block's voxels — mt_inorm_lrelu_bwd(part, part_nblk =
 * mt_lrelu_bwd_stats_blocks(V, C), part_cs = C, part_c0 = 0) then skips its own reduction over (g', y).  mean / rstd: [N][C] of that
 * norm.  mt_lrelu_bwd_stats_blocks returns 0 for shapes this path does not take (C % 4 != 0, C > 1024). */
int mt_lrelu_bwd_stats_blocks(long V, int C);
int mt_lrelu_bwd_stats(float* g, const float* y, const float* scale, const float* shift, float slope,
                       const float* y2, const float* scale2, const float* shift2, float slope2, float* gcopy,
                       const float* mean, const float* rstd, float* part, int N, long V, int C, int gdtype /* of g, gcopy */,
                       int ydtype /* of y, y2 */, mt_stream_t stream);
/* per-channel sum over all voxels: out[C] (+)= sum_{n,v} x[n,v,c]  (bias gradients of heads) */
size_t mt_channel_sum_workspace(int N, long V, int C);
int mt_channel_sum(const float* x, int xcs, int N, long V, int C, float* out, int accumulate,
                   void* ws, size_t ws_bytes, int dtype /* storage type of x */, mt_stream_t stream);
/* Storage-type conversion, the boundary op of the mixed-precision mode: dst[r][c] (+)= src[r][c] for `rows` voxels (all samples) of
 * C channels, each side with its own storage type (MT_F32 | MT_BF16 | MT_F16) and channel stride in elements.  accumulate adds in fp32 and
 * rounds once.  A kernel that does not take a tensor's storage type (the *_io_supported queries) works on such a copy. */
int mt_cast(const void* src, int scs, int sdtype, void* dst, int dcs, int ddtype, long rows, int C, int accumulate,
            mt_stream_t stream);

/* ---- losses --------------------------------------------------------------------------------- */
/* MultiTalent loss for one deep-supervision level (MultiTalent_Trainer_DDP.py:544-623):
 * logits [B][V][C=47 (cs)], target [B][V] float label map, valid[B] = 64-bit mask of valid output
 * channels, lut[C] = 64-bit mask over label values belonging to each channel's region
 * (Task100_MultiTalent.py:118-166).  Forward: stats[B][C][4] = (bce_sum, tp, fp, fn).
 * Backward = exact vector-Jacobian product of the forward: given gstats[B][C][4] = dLoss/dstats,
 * dlogits[b,v,c] = valid * ( g0*(sigmoid-y) + sigmoid*(1-sigmoid)*( y*(g1-g3) + (1-y)*g2 ) ), 0 for invalid
 * channels — written once (the reference's autograd zero-fills a full tensor per region). */
int mt_multitalent_loss_fwd(const float* logits, int cs, const float* target, int B, long V, int C,
                            const uint64_t* valid, const uint64_t* lut, float* stats, void* ws,
                            size_t ws_bytes, mt_stream_t stream);
size_t mt_loss_workspace(int B, long V, int C);
int mt_multitalent_loss_bwd(const float* logits, int cs, const float* target, int B, long V, int C,
                            const uint64_t* valid, const uint64_t* lut, const float* gstats,
                            float* dlogits, int dcs, mt_stream_t stream);
/* Online evaluation of the MultiTalent trainers (MultiTalent_Trainer_DDP.py:372-398): hard predictions sigmoid(x) > 0.5 of the
 * full-resolution logits against the region masks; stats[B][C][3] = exact counts (tp, fp, fn) of the channels valid for the
 * sample's dataset, 0 elsewhere.  ws: mt_hard_stats_workspace(B, C) bytes (64-bit counters, integer atomics: order-free). */
int mt_multitalent_hard_stats(const float* logits, int cs, const float* target, int B, long V, int C,
                              const uint64_t* valid, const uint64_t* lut, float* stats, void* ws, size_t ws_bytes,
                              mt_stream_t stream);
size_t mt_hard_stats_workspace(int B, int C);
/* Softmax Dice+CE for one level (dice_loss.py:100-195,488-545; crossentropy.py:4-11):
 * stats[B][C][4] = (ce_sum (only c=0 slot used), tp, fp, fn). */
int mt_softmax_dice_ce_fwd(const float* logits, int cs, const float* target, int B, long V, int C,
                           float* stats, void* ws, size_t ws_bytes, mt_stream_t stream);
int mt_softmax_dice_ce_bwd(const float* logits, int cs, const float* target, int B, long V, int C,
                           const float* gstats /* [B][C][4], slot (b,0,0) = dLoss/dce_sum[b] */,
                           float* dlogits, int dcs, mt_stream_t stream);
/* The loss combination on top of the per-level statistics, value and gradient in one launch (the reference forms it with a few
 * dozen autograd operations on [B, C] tensors: MultiTalent_Trainer_DDP.py:598-623; dice_loss.py:150-183 + deep_supervision.py:37-42;
 * nnUNetTrainerV2_DDP.py:262-282).  stats [L][B][C][4] as written by the *_fwd calls of the L levels; dice [L][B][C][dice_stride] =
 * (tp, fp, fn, ...) the Dice ratios are formed from: stats + 1 with stride 4, or the sum of the statistics over ranks with stride 3.
 *   loss = sum_l ce_coef[l] * sum_{b, c in CE set} stats[l,b,c,0]  -  sum_l dice_coef[l] * sum_entries r,
 *   r = (2 tp + smooth_num) / max(2 tp + fp + fn + smooth_den + den_eps, clamp_min)   over channels c >= c0,
 * flags: MT_LOSS_CE_ALL_CHANNELS (else channel 0 carries the level's CE sum), MT_LOSS_DICE_OVER_BATCH (tp / fp / fn summed over b
 * before the ratio).  ce_coef, dice_coef: device vectors [L].  out3 = (loss, CE part, Dice part); gstats [L][B][C][4] =
 * dLoss/d(local stats), the argument of the *_bwd calls; its Dice part is multiplied by dice_grad_scale (the world size when `dice`
 * is the sum over ranks: the backward of the reference's all-gather sums the ranks' identical gradients, distributed.py:63-73). */
#define MT_LOSS_CE_ALL_CHANNELS 1
#define MT_LOSS_DICE_OVER_BATCH 2
int mt_loss_combine(const float* stats, const float* dice, int dice_stride, int L, int B, int C, const float* ce_coef,
                    const float* dice_coef, int flags, int c0, float smooth_num, float smooth_den, float den_eps, float clamp_min,
                    float dice_grad_scale, float* out3, float* gstats, mt_stream_t stream);

/* ---- optimizer (nnUNetTrainerV2.py:166-170; clip MultiTalent_Trainer_DDP.py:352,362) --------- */
int mt_sumsq(const float* x, long n, float* out /* [1], overwritten */, void* ws, size_t ws_bytes,
             mt_stream_t stream);
size_t mt_sumsq_workspace(long n);
/* torch.optim.SGD(nesterov=True, dampening=0): g = clip*g + wd*p; buf = mom*buf + g (buf=g on first
 * step); p -= lr*(g + mom*buf).  clip_coef_dev: device scalar total_norm^2 -> coef = min(1, max_norm/(sqrt+1e-6)) */
int mt_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float wd, float mom,
                    int first_step, const float* sumsq_dev, float max_norm, mt_stream_t stream);

/* ---- sliding-window inference (neural_network.py:287-428,502-591) ---------------------------- */
/* acc[c][x] (+)= w * nonlin(logits[flip(x)][c])  NCDHW accumulator over one tile: mirror-TTA mean */
int mt_flip_accumulate(const float* logits, int cs, int D, int H, int W, int C, int flipD, int flipH,
                       int flipW, int nonlin /*0 none,1 sigmoid,2 softmax*/, float weight, float* acc,
                       int first, mt_stream_t stream);
/* Fused form of (1x1x1 head -> mt_flip_accumulate) for inference: p describes the head (src = its lazily activated input of N
 * samples, Cin, Cout <= 64, wpack layout 1 ck 16, bias; Db,Hb,Wb = tile size); sample `sample` is evaluated, passed through the
 * nonlinearity, un-flipped and accumulated into acc[Cout][D][H][W] like mt_flip_accumulate — the logits are never stored. */
int mt_head_flip_accumulate(const mt_pointwise_t* p, int sample, int flipD, int flipH, int flipW, int nonlin, float weight,
                            float* acc, int first, mt_stream_t stream);
/* All mirror combinations of one tile AND the overlap-add in one kernel: samples sample0 .. sample0+nsamples-1 of p's source are
 * the flipped versions of the tile (flips[k]: bit 0 D, bit 1 H, bit 2 W); per output voxel
 *   agg[c][x0+d][y0+h][z0+w] += gauss[d][h][w] * weight * sum_k nonlin(head(sample_k[flip_k(d,h,w)]))_c ;  nb[...] += gauss
 * (neural_network.py:531-586 and :384-394).  gauss NULL = 1, nb NULL = not updated.  flips is a HOST array of nsamples ints. */
int mt_head_mirror_accumulate(const mt_pointwise_t* p, int sample0, int nsamples, const int32_t* flips, int nonlin, float weight,
                              const float* gauss, float* agg, float* nb, long aX, long aY, long aZ, int x0, int y0, int z0,
                              mt_stream_t stream);
/* The network batch of a group of tiles, mirror flips included, straight from the (padded) volume vol[C][X][Y][Z]:
 * out[k][C][D][H][W] = tile k at origin (x0,y0,z0) read through the reflections in `flips` (bit 0 D, bit 1 H, bit 2 W) —
 * `x = torch.flip(x, axes)` of neural_network.py:531-586 as index arithmetic.  desc: HOST array of ntiles x (x0, y0, z0, flips). */
int mt_extract_tiles(const float* vol, int C, long X, long Y, long Z, float* out, int ntiles, int D, int H, int W,
                     const int32_t* desc, mt_stream_t stream);
/* agg[c, tile] += acc * gauss ; nb[tile] += gauss   (neural_network.py:388-394) */
int mt_tile_accumulate(const float* acc, const float* gauss, int C, int D, int H, int W, float* agg,
                       float* nb, long aX, long aY, long aZ, int x0, int y0, int z0, mt_stream_t stream);
/* probs = agg/nb; seg = argmax_c or per-channel >0.5 in regions_class_order (neural_network.py:405-417) */
int mt_normalize_threshold(float* agg, const float* nb, int C, long V, const int32_t* class_order,
                           int use_regions, int32_t* seg, mt_stream_t stream);
/* Spatial augmentation (second half): batchgenerators' SpatialTransform as configured by
 * data_augmentation_moreDA.py:66-80 (rotation + scaling, order_data 3, order_seg 1, constant borders) = per sample
 * scipy.ndimage.map_coordinates over the affine field  x = M (o - (O-1)/2) + centre.
 * mt_spline_prefilter3: in-place cubic B-spline prefilter of vol[NC, D, H, W] (pole sqrt(3)-2, gain 6, scipy's exact mirror
 * boundary initialisation = what map_coordinates(order=3, mode='constant') filters with).
 * mt_affine_sample: dst[N, C, OD, OH, OW] from src[N, C, D, H, W]; mats = N x 12 floats (row-major 3x3 M, then the centre);
 * mode 0 nearest, 1 linear, 3 cubic (src must be prefiltered), 11 = segmentation rule for order 1 (label c where the linear
 * interpolation of (seg == c) is >= 0.5, larger labels override, else 0).  A coordinate outside [0, n-1] on any axis gives cval
 * (mode 11: 0); stencil taps off the array read the mirrored element (scipy 'constant' mode). */
int mt_spline_prefilter3(float* vol, int NC, int D, int H, int W, int axes /* bit 0 W, 1 H, 2 D; 7 = 3D, 3 = per slice */,
                         mt_stream_t stream);
int mt_affine_sample(const float* src, int N, int C, int D, int H, int W, float* dst, int OD, int OH, int OW,
                     const float* mats, int mode, float cval,
                     int planar /* 1: nnU-Net's "dummy 2D" augmentation — every D slice is an independent 2D image, OD == D */,
                     mt_stream_t stream);
/* Export post-processing: save_segmentation_nifti_from_softmax (segmentation_export.py:27-160) without the
 * resampled 47-channel intermediate — probabilities [C, D, H, W] are interpolated at the OD x OH x OW grid of the original
 * spacing (order 1, skimage/scipy half-pixel rule, edge clamp; sep_axis in 0..2 = the anisotropic axis sampled nearest as in
 * resample_data_or_seg's separate-z branch, preprocessing.py:134-187; -1 = trilinear), classified per voxel (use_regions: last
 * channel i in order with p_i > 0.5 gives class_order[i], else 0; otherwise argmax) and written as uint8 into the uncropped
 * volume out[FD, FH, FW] at offset (bD, bH, bW), clipped to the volume (crop_bbox re-insertion).  The caller zero-fills out. */
int mt_resample_classify(const float* probs, int C, int D, int H, int W, int OD, int OH, int OW, int sep_axis,
                         const int32_t* class_order, int use_regions, uint8_t* out, long FD, long FH, long FW,
                         int bD, int bH, int bW, mt_stream_t stream);
/* ---- device-side target preparation ----------------------------------------
 * Deep-supervision label pyramid: DownsampleSegForDSTransform2 / downsample_seg_for_ds_transform2 (downsampling.py:70-104,
 * order 0 = nearest through batchgenerators' resize_segmentation -> skimage.transform.resize(order 0, mode "edge") ->
 * scipy.ndimage.zoom(order 0, grid_mode=True): source index = floor((o + 0.5) * in / out), clamped) and, when
 * remove_minus_one != 0, RemoveLabelTransform(-1, 0) (data_augmentation_moreDA.py:117).  src/dst: [NC, D, H, W] float32
 * label maps (the reference's target layout [B,1,D,H,W] with NC = B). */
int mt_downsample_seg_nearest(const float* src, int NC, int Di, int Hi, int Wi, float* dst, int Do, int Ho, int Wo,
                              int remove_minus_one, mt_stream_t stream);

/* GaussianBlurTransform (data_augmentation_moreDA.py:87-88 -> scipy.ndimage.gaussian_filter per channel): ONE axis pass of the
 * separable filter, dst = correlate1d(src, w) along axis (0 D, 1 H, 2 W) with w[k] ~ exp(-k^2 / 2 sigma^2), radius int(4 sigma + 0.5),
 * 'reflect' boundaries; sigma[NC] per (sample, channel), sigma <= 0 copies that channel.  src != dst. */
int mt_gaussian_blur_axis(const float* src, float* dst, int NC, int D, int H, int W, int axis, const float* sigma, mt_stream_t stream);

/* NCDHW <-> NDHWC transposes used at the module boundary */
int mt_ncdhw_to_ndhwc(const float* in, float* out, int N, int C, long V, int ocs, mt_stream_t stream);
int mt_ndhwc_to_ncdhw(const float* in, int ics, float* out, int N, int C, long V, mt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MTSEG_H */
