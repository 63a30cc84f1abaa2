"""ctypes binding of libmtseg_hip.so (C ABI declared in include/mtseg.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (the engine must never silently run on a PyTorch/CPU path).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MT_LIB_VARIANT selects an instrumented build of the same ABI (tools/build_variant.sh) for kernel timing experiments
LIB_PATH = os.path.join(_HERE, os.environ.get('MT_LIB_VARIANT', 'libmtseg_hip.so'))

MT_MAX_CHUNKS = 64
c_float_p = C.POINTER(C.c_float)


class mt_src_t(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('cs', C.c_int32), ('C', C.c_int32), ('scale', C.c_void_p),
                ('shift', C.c_void_p), ('slope', C.c_float), ('dtype', C.c_int32)]


class mt_bwd_stats_t(C.Structure):
    _fields_ = [('y', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('ycs', C.c_int32), ('c0', C.c_int32), ('C', C.c_int32), ('slope', C.c_float)]


class mt_conv3d_t(C.Structure):
    _fields_ = [('src', mt_src_t * 2), ('nsrc', C.c_int32),
                ('N', C.c_int32), ('Di', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32),
                ('dilD', C.c_int32), ('dilH', C.c_int32), ('dilW', C.c_int32),
                ('Do', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
                ('KD', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32),
                ('SD', C.c_int32), ('SH', C.c_int32), ('SW', C.c_int32),
                ('PD', C.c_int32), ('PH', C.c_int32), ('PW', C.c_int32),
                ('Cin', C.c_int32), ('Cout', C.c_int32),
                ('wpack', C.c_void_p), ('bias', C.c_void_p),
                ('out0', C.c_void_p), ('ocs0', C.c_int32),
                ('out1', C.c_void_p), ('ocs1', C.c_int32),
                ('csplit', C.c_int32), ('accumulate', C.c_int32),
                ('stats_part', C.c_void_p),
                ('OD', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32),
                ('osD', C.c_int32), ('osH', C.c_int32), ('osW', C.c_int32),
                ('ooD', C.c_int32), ('ooH', C.c_int32), ('ooW', C.c_int32), ('mma', C.c_int32), ('odtype', C.c_int32),
                ('bstats', mt_bwd_stats_t), ('select', C.c_uint32), ('max_workgroups', C.c_int32)]


class mt_pointwise_t(C.Structure):
    _fields_ = [('src', mt_src_t),
                ('N', C.c_int32), ('Db', C.c_int32), ('Hb', C.c_int32), ('Wb', C.c_int32),
                ('Di', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32),
                ('siD', C.c_int32), ('siH', C.c_int32), ('siW', C.c_int32),
                ('soD', C.c_int32), ('soH', C.c_int32), ('soW', C.c_int32),
                ('Cin', C.c_int32), ('Cout', C.c_int32),
                ('wpack', C.c_void_p), ('bias', C.c_void_p),
                ('out', C.c_void_p), ('ocs', C.c_int32), ('accumulate', C.c_int32),
                ('stats_part', C.c_void_p), ('odtype', C.c_int32), ('scatter', C.c_int32), ('mma', C.c_int32)]


MT_F32, MT_BF16, MT_F16 = 0, 1, 2
MT_ABI_VERSION = 4

_vp, _i, _l, _f, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); must list EVERY symbol of include/mtseg.h (checked by tests/test_abi.py)
SIGNATURES = {
    'mt_last_error': (C.c_char_p, []),
    'mt_abi_version': (_i, []),
    'mt_probe_device': (_i, [_vp, _sz, _P(_i), C.c_char_p, _sz, _vp]),
    'mt_pack_conv_weights': (_i, [_vp, _vp, _P(_sz), _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, _i, _vp, _vp]),
    'mt_conv3d_fwd': (_i, [_P(mt_conv3d_t), _vp]),
    'mt_conv3d_stats_blocks': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_ck': (_i, [_P(mt_conv3d_t)]),
    'mt_pack_desc_size': (_sz, []),
    'mt_pack_desc_fill': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, _i, _vp]),
    'mt_pack_batched': (_i, [_vp, _i, _vp]),
    'mt_conv3d_bwd_data_strided': (_i, [_P(mt_conv3d_t), _vp]),
    'mt_conv3d_bwd_data_strided_supported': (_i, [_P(mt_conv3d_t)]),
    'mt_downsample_seg_nearest': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    'mt_gaussian_blur_axis': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'mt_conv3d_pack_layout': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_bwd_data_strided_pack_layout': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_kernel_name': (_i, [_P(mt_conv3d_t), C.c_char_p, _sz]),
    'mt_conv3d_bwd_stats_supported': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_bwd_weight_kernel_name': (_i, [_P(mt_conv3d_t), _P(mt_src_t), C.c_char_p, _sz]),
    'mt_conv3d_bwd_data_strided_kernel_name': (_i, [_P(mt_conv3d_t), C.c_char_p, _sz]),
    'mt_conv3d_bwd_weight_workspace': (_sz, [_P(mt_conv3d_t)]),
    'mt_conv3d_bwd_weight': (_i, [_P(mt_conv3d_t), _P(mt_src_t), _vp, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp]),
    'mt_pointwise_fwd': (_i, [_P(mt_pointwise_t), _vp]),
    'mt_pointwise_stats_blocks': (_i, [_P(mt_pointwise_t)]),
    'mt_head_bwd_supported': (_i, [_i, _i]),
    'mt_head_bwd_workspace': (_sz, [_i, _l, _i, _i]),
    'mt_head_bwd': (_i, [_P(mt_src_t), _vp, _i, _i, _l, _i, _i, _vp, _vp, _i, _i, _i, _vp, _l, _l, _vp, _i, _P(C.c_int), _vp, _sz, _vp]),
    'mt_head_bwd_io_supported': (_i, [_i, _i, _i, _i, _i, _i]),
    'mt_pointwise_io_supported': (_i, [_P(mt_pointwise_t)]),
    'mt_pointwise_pack_layout': (_i, [_P(mt_pointwise_t)]),
    'mt_inorm_finalize': (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    'mt_inorm_lrelu_apply': (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _f, _vp, _i, _i, _l, _i, _i, _vp]),
    'mt_inorm_bwd_workspace': (_sz, [_i, _l, _i]),
    'mt_inorm_lrelu_bwd': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _i, _l, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _i, _i, _vp]),
    'mt_lrelu_bwd': (_i, [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _f, _vp, _i, _i, _l, _i, _i, _i, _vp]),
    'mt_lrelu_bwd_stats_blocks': (_i, [_l, _i]),
    'mt_lrelu_bwd_stats': (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _l, _i, _i, _i, _vp]),
    'mt_channel_sum_workspace': (_sz, [_i, _l, _i]),
    'mt_channel_sum': (_i, [_vp, _i, _i, _l, _i, _vp, _i, _vp, _sz, _i, _vp]),
    'mt_cast': (_i, [_vp, _i, _i, _vp, _i, _i, _l, _i, _i, _vp]),
    'mt_conv3d_io_supported': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_bwd_data_strided_io_supported': (_i, [_P(mt_conv3d_t)]),
    'mt_conv3d_bwd_weight_io_supported': (_i, [_P(mt_conv3d_t), _P(mt_src_t)]),
    'mt_multitalent_loss_fwd': (_i, [_vp, _i, _vp, _i, _l, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'mt_loss_workspace': (_sz, [_i, _l, _i]),
    'mt_multitalent_loss_bwd': (_i, [_vp, _i, _vp, _i, _l, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    'mt_multitalent_hard_stats': (_i, [_vp, _i, _vp, _i, _l, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'mt_hard_stats_workspace': (_sz, [_i, _i]),
    'mt_softmax_dice_ce_fwd': (_i, [_vp, _i, _vp, _i, _l, _i, _vp, _vp, _sz, _vp]),
    'mt_softmax_dice_ce_bwd': (_i, [_vp, _i, _vp, _i, _l, _i, _vp, _vp, _i, _vp]),
    'mt_loss_combine': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    'mt_sumsq': (_i, [_vp, _l, _vp, _vp, _sz, _vp]),
    'mt_sumsq_workspace': (_sz, [_l]),
    'mt_sgd_nesterov': (_i, [_vp, _vp, _vp, _l, _f, _f, _f, _i, _vp, _f, _vp]),
    'mt_flip_accumulate': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    'mt_spline_prefilter3': (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    'mt_affine_sample': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _f, _i, _vp]),
    'mt_resample_classify': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    'mt_head_flip_accumulate': (_i, [_P(mt_pointwise_t), _i, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    'mt_head_mirror_accumulate': (_i, [_P(mt_pointwise_t), _i, _i, _vp, _i, _f, _vp, _vp, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    'mt_extract_tiles': (_i, [_vp, _i, _l, _l, _l, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mt_tile_accumulate': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    'mt_normalize_threshold': (_i, [_vp, _vp, _i, _l, _vp, _i, _vp, _vp]),
    'mt_ncdhw_to_ndhwc': (_i, [_vp, _vp, _i, _i, _l, _i, _vp]),
    'mt_ndhwc_to_ncdhw': (_i, [_vp, _i, _vp, _i, _i, _l, _vp]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError when it is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64: import it FIRST so this library binds to the same HIP runtime instance
    # (two runtimes in one process cannot share streams / device pointers).
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "libmtseg_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C multitalent_amd/csrc`). The engine has no CPU/PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mt_abi_version() != MT_ABI_VERSION:
        raise RuntimeError("libmtseg_hip.so ABI version %d, this binding needs %d: rebuild it (make -C multitalent_amd/csrc)"
                           % (lib.mt_abi_version(), MT_ABI_VERSION))
    _lib = lib
    probe_device()          # fail loudly on a device the vector-load kernels are not valid for (no-op without a GPU)
    return lib


_probed = {}


def probe_device(index=None):
    """Once per device: the library is built for gfx950 only and its vector-load kernels assume per-dword range checks and
    dword-aligned 16-byte buffer loads (mt_probe_device).  A device that behaves differently must fail HERE, loudly, not compute
    garbage later.  Called by Engine.attach and ops.Act construction paths through `ensure_device`."""
    import torch
    if not torch.cuda.is_available():
        return None
    index = torch.cuda.current_device() if index is None else int(index)
    if index in _probed:
        return _probed[index]
    lib = load()
    with torch.cuda.device(index):
        scratch = torch.zeros(8192, dtype=torch.float32, device='cuda')
        ok = C.c_int(0)
        arch = C.create_string_buffer(64)
        rc = lib.mt_probe_device(scratch.data_ptr(), scratch.numel() * 4, C.byref(ok), arch, 64, torch.cuda.current_stream().cuda_stream)
        check(rc, 'probe_device')
    if not ok.value:
        raise RuntimeError("libmtseg_hip: device %d (%s) does not return per-dword range-checked, dword-aligned 16-byte buffer loads; "
                           "the vector-load kernels of this library would compute wrong results on it (see include/mtseg.h, "
                           "mt_probe_device)" % (index, arch.value.decode()))
    _probed[index] = arch.value.decode()
    return _probed[index]


def check(rc, what=''):
    if rc != 0:
        msg = load().mt_last_error()
        raise RuntimeError("libmtseg_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ''))
