"""Thin torch-tensor wrappers over the C ABI (include/mtseg.h).

Everything here is plumbing: torch owns device memory and streams, the arithmetic happens in
libmtseg_hip.so.  Activations are NDHWC tensors [N, D, H, W, C] (possibly channel slices of a wider buffer), float32 or — the
storage of the mixed-precision mode — float16 (activations) / bfloat16 (gradients) (mt_src_t.dtype / odtype).  There is NO CPU fallback: tensors must live on a HIP device.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import mt_conv3d_t, mt_pointwise_t, mt_src_t


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("multitalent_amd ops require HIP device tensors (no CPU fallback)")


_DT_CODES = {torch.float32: _lib.MT_F32, torch.bfloat16: _lib.MT_BF16, torch.float16: _lib.MT_F16}


class Act:
    """A (lazy) activation: channel slice [c0, c0+C) of an NDHWC buffer `buf` [N,D,H,W,cs], read as
    lrelu_slope(buf*scale + shift) when scale is not None (InstanceNorm+LeakyReLU applied on load)."""

    __slots__ = ('buf', 'c0', 'C', 'scale', 'shift', 'slope', 'mean', 'rstd')

    def __init__(self, buf, c0=0, C=None, scale=None, shift=None, slope=1.0, mean=None, rstd=None):
        assert buf.dim() == 5 and buf.is_contiguous() and buf.dtype in _DT_CODES
        self.buf, self.c0 = buf, c0
        self.C = buf.shape[4] - c0 if C is None else C
        self.scale, self.shift, self.slope = scale, shift, float(slope)
        self.mean, self.rstd = mean, rstd

    @property
    def N(self): return self.buf.shape[0]

    @property
    def spatial(self): return tuple(self.buf.shape[1:4])

    @property
    def cs(self): return self.buf.shape[4]

    @property
    def V(self): return self.buf.shape[1] * self.buf.shape[2] * self.buf.shape[3]

    @property
    def dtype(self): return self.buf.dtype

    @property
    def dt(self):
        """storage type code of the C ABI (MT_F32 | MT_BF16 | MT_F16)"""
        return _DT_CODES[self.buf.dtype]

    def data_ptr(self):
        return self.buf.data_ptr() + self.buf.element_size() * self.c0

    def with_buf(self, buf, c0=0):
        """the same lazy activation over another buffer (a storage-type copy of the raw values)"""
        return Act(buf, c0=c0, C=self.C, scale=self.scale, shift=self.shift, slope=self.slope, mean=self.mean, rstd=self.rstd)

    def src(self):
        s = mt_src_t()
        s.ptr = self.data_ptr()
        s.cs = self.cs
        s.C = self.C
        s.dtype = self.dt
        s.scale = self.scale.data_ptr() if self.scale is not None else None
        s.shift = self.shift.data_ptr() if self.shift is not None else None
        s.slope = self.slope
        return s

    def dense(self):
        """Materialise to a plain [N,D,H,W,C] tensor with torch ops (test/debug helper only)."""
        x = self.buf[..., self.c0:self.c0 + self.C].float()
        if self.scale is not None:
            x = x * self.scale[:, None, None, None, :] + self.shift[:, None, None, None, :]
            x = torch.where(x > 0, x, x * self.slope)
        return x.contiguous()


def _triple(v):
    return tuple(int(i) for i in v) if isinstance(v, (list, tuple)) or hasattr(v, '__len__') else (int(v),) * 3


class ConvGeom:
    """Geometry of one convolution problem as the kernels see it."""

    def __init__(self, in_spatial, kernel, stride=(1, 1, 1), pad=None, dil=(1, 1, 1), out_spatial=None):
        self.inp = _triple(in_spatial)
        self.k = _triple(kernel)
        self.s = _triple(stride)
        self.p = tuple((k - 1) // 2 for k in self.k) if pad is None else _triple(pad)
        self.dil = _triple(dil)
        if out_spatial is None:
            out_spatial = tuple(((i - 1) * d + 1 + 2 * p - k) // s + 1
                                for i, d, p, k, s in zip(self.inp, self.dil, self.p, self.k, self.s))
        self.out = _triple(out_spatial)


# matrix input type given to every mt_conv3d_t built by fill_conv: 0 = fp32, 1 = bf16 inputs / fp32 accumulation.  The engine
# sets it on entry of pack / forward / backward (Engine.mma), so several engines with different modes can coexist.
_MMA = 0


def set_mma(mode):
    """DEFAULT matrix input type of fill_conv / fill_pointwise when their caller passes none (kernel-level tests and tools; the engine
    passes its own): 0 = fp32 (exact), 1 = bf16 inputs with fp32 accumulation.  Host-side convenience — the C ABI takes mt_conv3d_t.mma."""
    global _MMA
    _MMA = int(mode)


# ---- kernel selection (mt_conv3d_t.select / max_workgroups, ABI 4) ------------------------------------------------------------------
# The library keeps no process-wide switches: which kernel family serves a problem is a field of the problem.  What lives HERE is the
# default this process writes into the structs it builds — 0 (the library's policy) unless a test, an A/B tool (MT_SELECT) or
# set_option changes it.  The legacy option names of rounds 1-5 map onto the fields.
_SEL_SHIFT = {'conv_wino': 0, 'conv_bf16': 2, 'conv_x16': 4, 'conv_tapsplit': 6, 'bwdw_wino': 8, 'bwdw_tr16': 10, 'bwdw_cw': 12}
_SEL_ALIASES = {'wino': 'conv_wino', 'm16': 'conv_bf16', 'x16': 'conv_x16', 'tapsplit': 'conv_tapsplit'}
_select = 0
_caps = {}


def _sel_set(name, code):
    global _select
    sh = _SEL_SHIFT[name]
    _select = (_select & ~(3 << sh)) | ((code & 3) << sh)


def set_option(name, value):
    """Default kernel selection of the problems this process builds (tests, A/B tools).  Legacy names and values:
    conv_wino | conv_bf16 | conv_tapsplit: 0 never, 1 the library's policy, 2 wherever eligible; bwdw_wino: 0 | 1;
    conv_x16 | bwdw_tr16: 0 | 1 | n > 1 = wherever eligible with at most n workgroups (4096: no cap); wino_persist: n > 1 = at most n workers
    per output-channel tile; bwdw_cw: 4 (policy) | 2 | 1 | 104 (four tiles also on small problems)."""
    name = _SEL_ALIASES.get(name, name)
    value = int(value)
    if name == 'wino_persist':
        _caps.pop(name, None) if value <= 1 else _caps.__setitem__(name, value)
    elif name in ('conv_x16', 'bwdw_tr16'):
        _sel_set(name, 1 if value == 0 else 0 if value == 1 else (2 if name == 'conv_x16' else 0))
        _caps.pop(name, None) if value <= 1 or value >= 4096 else _caps.__setitem__(name, value)
    elif name == 'bwdw_cw':
        _sel_set(name, {4: 0, 1: 1, 2: 2, 104: 3}[value])
    elif name in _SEL_SHIFT:
        _sel_set(name, {0: 1, 1: 0, 2: 2}[value])
    else:
        raise ValueError("set_option: unknown option %r" % (name,))


def apply_selection(p):
    """Write the process default selection into an mt_conv3d_t that was built earlier (tests that keep ONE struct while they switch
    between kernel families)."""
    p.select = _select
    p.max_workgroups = min(_caps.values()) if _caps else 0
    return p


def options_are_default():
    return _select == _select_env and not _caps and _MMA == 0


def _parse_select_env():
    """MT_SELECT="x16=off,wino=force,tapsplit=off": the default selection of a whole process (A/B runs of bench.py: tools/ab_env.sh)."""
    for item in os.environ.get('MT_SELECT', '').replace(' ', '').split(','):
        if item:
            k, v = item.split('=')
            k = _SEL_ALIASES.get(k, k)
            if k == 'bwdw_cw':
                set_option(k, int(v))
            else:
                _sel_set(k, {'default': 0, 'off': 1, 'force': 2}[v])


def fill_conv(srcs, geom, Cout, wpack=None, bias=None, out0=None, out1=None, csplit=None, accumulate=False,
              stats_part=None, place=None, mma=None):
    """Build an mt_conv3d_t.  srcs: list of 1-2 Act; out0/out1: Act-like destination slices.
    place = (stored_spatial, out_stride, out_offset): logical output o is written at o*stride + offset."""
    p = mt_conv3d_t()
    p.mma = _MMA if mma is None else int(mma)
    p.select = _select
    p.max_workgroups = min(_caps.values()) if _caps else 0
    p.nsrc = len(srcs)
    for i, a in enumerate(srcs):
        p.src[i] = a.src()
    p.N = srcs[0].N
    p.Di, p.Hi, p.Wi = geom.inp
    p.dilD, p.dilH, p.dilW = geom.dil
    p.Do, p.Ho, p.Wo = geom.out
    p.KD, p.KH, p.KW = geom.k
    p.SD, p.SH, p.SW = geom.s
    p.PD, p.PH, p.PW = geom.p
    p.Cin = sum(a.C for a in srcs)
    p.Cout = Cout
    p.wpack = wpack.data_ptr() if wpack is not None else None
    p.bias = bias.data_ptr() if bias is not None else None
    if out0 is not None:
        p.out0 = out0.data_ptr()
        p.ocs0 = out0.cs
        p.odtype = out0.dt
    if out1 is not None:
        assert out0 is not None and out1.dt == out0.dt, "the two destinations of a convolution share one storage type"
        p.out1 = out1.data_ptr()
        p.ocs1 = out1.cs
    p.csplit = Cout if csplit is None else csplit
    p.accumulate = 1 if accumulate else 0
    p.stats_part = stats_part.data_ptr() if stats_part is not None else None
    if place is not None:
        (p.OD, p.OH, p.OW), (p.osD, p.osH, p.osW), (p.ooD, p.ooH, p.ooW) = place
    return p


def bwd_data_parity_classes(geom):
    """Backward-data of a strided conv (geometry `geom`) as one exact stride-1 convolution per parity class of the input
    position:  dX[S*m + par] = sum_j dY[m - pad' + j] * W[tmax - S*j]  over the taps t = tmax - S*j congruent to
    par + P (mod S).  Returns [(ConvGeom on dY, placement, tapmap)] for fill_conv / pack_conv_weights; classes without taps
    are omitted (their input positions receive no gradient from this conv)."""
    dims = []
    for d in range(3):
        K, S, P, Di = geom.k[d], geom.s[d], geom.p[d], geom.inp[d]
        opts = []
        for par in range(S):
            taps = [t for t in range(K) if (t - par - P) % S == 0]
            cnt = (Di - par + S - 1) // S
            if not taps or cnt <= 0:
                continue
            tmax = max(taps)
            opts.append(dict(par=par, k=len(taps), pad=(tmax - par - P) // S, tb=tmax, ts=-S, cnt=cnt, S=S))
        dims.append(opts)
    out = []
    for a in dims[0]:
        for b in dims[1]:
            for c in dims[2]:
                sel = (a, b, c)
                geomc = ConvGeom(geom.out, tuple(x['k'] for x in sel), (1, 1, 1), tuple(x['pad'] for x in sel),
                                 out_spatial=tuple(x['cnt'] for x in sel))
                place = (geom.inp, tuple(x['S'] for x in sel), tuple(x['par'] for x in sel))
                tapmap = [v for x in sel for v in (x['tb'], x['ts'])]
                out.append((geomc, place, tapmap))
    return out


def conv_ck(p):
    ck = _lib.load().mt_conv3d_ck(C.byref(p))
    if ck <= 0:
        raise RuntimeError("conv3d: no kernel configuration for this shape")
    return ck


def conv_pack_layout(p):
    return _lib.load().mt_conv3d_pack_layout(C.byref(p))


def conv_bwd_data_strided_pack_layout(p):
    return _lib.load().mt_conv3d_bwd_data_strided_pack_layout(C.byref(p))


def conv_kernel_name(p):
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().mt_conv3d_kernel_name(C.byref(p), buf, 128), 'conv3d_kernel_name')
    return buf.value.decode()


def conv_bwd_weight_kernel_name(p, y):
    buf = C.create_string_buffer(128)
    ys = y.src()
    _lib.check(_lib.load().mt_conv3d_bwd_weight_kernel_name(C.byref(p), C.byref(ys), buf, 128), 'conv3d_bwd_weight_kernel_name')
    return buf.value.decode()


def conv_bwd_data_strided_kernel_name(p):
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().mt_conv3d_bwd_data_strided_kernel_name(C.byref(p), buf, 128), 'conv3d_bwd_data_strided_kernel_name')
    return buf.value.decode()


def conv_stats_blocks(p):
    return _lib.load().mt_conv3d_stats_blocks(C.byref(p))


_pack_recorder = None      # list while an Engine records its per-step packing program (see Engine._pack)


class PackProgram:
    """All weight packings of one optimizer step as ONE launch (mt_pack_batched): the descriptors are recorded once from the
    ordinary pack_conv_weights calls and live in a device table; valid while the weight and destination buffers are."""

    def __init__(self, records, device):
        lib = _lib.load()
        sz = lib.mt_pack_desc_size()
        self.n = len(records)
        host = (C.c_uint8 * (sz * self.n))()
        self.keep = []
        for i, (w, out, args, tm) in enumerate(records):
            tmc = (C.c_int32 * 6)(*[int(t) for t in tm]) if tm is not None else None
            _lib.check(lib.mt_pack_desc_fill(C.cast(C.byref(host, i * sz), C.c_void_p), _ptr(w), _ptr(out), *args,
                                             C.cast(tmc, C.c_void_p) if tmc is not None else None), 'pack_desc_fill')
            self.keep.append((w, out))
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)

    def run(self):
        _lib.check(_lib.load().mt_pack_batched(_ptr(self.table), self.n, _stream()), 'pack_batched')


def pack_conv_weights(w, C0, C1, Cout, kernel, strides, flip, ck, out=None, layout=1, tapmap=None):
    """strides = (s_ci, s_co, s_kd, s_kh, s_kw) element strides of `w` for W_eff[tap][ci][co].
    layout 1 = every MFMA kernel (mt_conv3d_fwd with ck = mt_conv3d_ck, mt_pointwise_fwd with ck = POINTWISE_CK)."""
    lib = _lib.load()
    _check_dev(w)
    n = C.c_size_t(0)
    kd, kh, kw = kernel
    tm = (C.c_int32 * 6)(*[int(i) for i in tapmap]) if tapmap is not None else None
    _lib.check(lib.mt_pack_conv_weights(None, None, C.byref(n), C0, C1, Cout, kd, kh, kw, *strides, int(flip), ck, layout, None, None), 'pack(query)')
    if out is None:
        out = torch.empty(n.value, dtype=torch.float32, device=w.device)
    assert out.numel() >= n.value
    if _pack_recorder is not None:
        _pack_recorder.append((w, out, (C0, C1, Cout, kd, kh, kw) + tuple(int(s) for s in strides) + (int(flip), ck, layout), tapmap))
        return out
    _lib.check(lib.mt_pack_conv_weights(_ptr(w), _ptr(out), C.byref(n), C0, C1, Cout, kd, kh, kw, *strides, int(flip), ck, layout,
                                        C.cast(tm, C.c_void_p) if tm is not None else None, _stream()), 'pack')
    return out


def conv_weight_strides(w, transposed_layout=False, as_bwd_data=False):
    """Element strides (s_ci, s_co, s_kd, s_kh, s_kw) for a contiguous nn.Conv3d weight [Cout,Cin,kd,kh,kw]
    (or nn.ConvTranspose3d weight [Cin,Cout,kd,kh,kw] when transposed_layout).  as_bwd_data swaps the
    roles of ci/co (the backward-data conv maps Cout channels back to Cin channels)."""
    k = w.shape[2] * w.shape[3] * w.shape[4]
    s_first, s_second = w.shape[1] * k, k  # strides of dim0 / dim1
    if not transposed_layout:
        s_co, s_ci = s_first, s_second
    else:
        s_ci, s_co = s_first, s_second
    if as_bwd_data:
        s_ci, s_co = s_co, s_ci
    return (s_ci, s_co, w.shape[3] * w.shape[4], w.shape[4], 1)


def conv3d_fwd(p):
    _lib.check(_lib.load().mt_conv3d_fwd(C.byref(p), _stream()), 'conv3d_fwd')


def conv_io_supported(p):
    return bool(_lib.load().mt_conv3d_io_supported(C.byref(p)))


def conv_bwd_data_strided_io_supported(p):
    return bool(_lib.load().mt_conv3d_bwd_data_strided_io_supported(C.byref(p)))


def conv_bwd_weight_io_supported(p, y):
    ys = y.src()
    return bool(_lib.load().mt_conv3d_bwd_weight_io_supported(C.byref(p), C.byref(ys)))


def pointwise_io_supported(p):
    return bool(_lib.load().mt_pointwise_io_supported(C.byref(p)))


def cast(src, dst, accumulate=False):
    """dst (+)= src between storage types (mt_cast); src / dst: Act over the raw values (channel slices allowed)."""
    assert src.N == dst.N and src.V == dst.V and src.C == dst.C
    _lib.check(_lib.load().mt_cast(C.c_void_p(src.data_ptr()), src.cs, src.dt, C.c_void_p(dst.data_ptr()), dst.cs, dst.dt,
                                   src.N * src.V, src.C, int(accumulate), _stream()), 'cast')


def conv3d_bwd_data_strided_supported(p):
    return bool(_lib.load().mt_conv3d_bwd_data_strided_supported(C.byref(p)))


def conv3d_bwd_data_strided(p):
    """dX of a strided 3x3x3 conv in one launch; p = FORWARD geometry with src[0] = dY, out0 = dX (see include/mtseg.h)."""
    _lib.check(_lib.load().mt_conv3d_bwd_data_strided(C.byref(p), _stream()), 'conv3d_bwd_data_strided')


def conv3d_bwd_weight_workspace(p):
    return _lib.load().mt_conv3d_bwd_weight_workspace(C.byref(p))


def conv3d_bwd_weight(p, y, dw, strides, accumulate, ws):
    ys = y.src()
    _lib.check(_lib.load().mt_conv3d_bwd_weight(C.byref(p), C.byref(ys), _ptr(dw), *strides, int(accumulate), _ptr(ws),
                                                ws.numel() * ws.element_size(), _stream()), 'conv3d_bwd_weight')


POINTWISE_CK = 16   # mt_pointwise_fwd takes weights packed with layout 1, ck 16


def fill_pointwise(src, base, in_spatial, si, so, Cout, wpack, bias, out, accumulate=False, stats_part=None, mma=0):
    p = mt_pointwise_t()
    p.src = src.src()
    p.N = src.N
    p.Db, p.Hb, p.Wb = base
    p.Di, p.Hi, p.Wi = in_spatial
    p.siD, p.siH, p.siW = si
    p.soD, p.soH, p.soW = so
    p.Cin = src.C
    p.Cout = Cout
    p.wpack = wpack.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.out = out.data_ptr()
    p.ocs = out.cs
    p.odtype = out.dt
    p.accumulate = 1 if accumulate else 0
    p.stats_part = stats_part.data_ptr() if stats_part is not None else None
    p.mma = int(mma)
    return p


def pointwise_pack_layout(p):
    return _lib.load().mt_pointwise_pack_layout(C.byref(p))


def pointwise_fwd(p):
    _lib.check(_lib.load().mt_pointwise_fwd(C.byref(p), _stream()), 'pointwise_fwd')


def head_bwd_supported(Cin, Cout):
    return bool(_lib.load().mt_head_bwd_supported(int(Cin), int(Cout)))


def head_bwd(x, dy, wpack_bwd, dx, accumulate_dx, dw, s_ci, s_co, dbias, accumulate_dw, ws):
    """Fused backward of a 1x1x1 head: x = Act (head input, lazy), dy = Act over the dense gradient of the logits, dx = Act over the
    gradient buffer of the head's input.  Returns True when dbias was produced (see mt_head_bwd)."""
    lib = _lib.load()
    assert dy.dt == _lib.MT_F32, "mt_head_bwd reads the fp32 loss gradient"
    xs = x.src()
    done = C.c_int(0)
    _lib.check(lib.mt_head_bwd(C.byref(xs), C.c_void_p(dy.data_ptr()), dy.cs, x.N, x.V, x.C, dy.C, _ptr(wpack_bwd),
                               C.c_void_p(dx.data_ptr()), dx.cs, dx.dt, int(accumulate_dx), _ptr(dw), int(s_ci), int(s_co), _ptr(dbias),
                               int(accumulate_dw), C.byref(done), _ptr(ws), ws.numel() * ws.element_size(), _stream()), 'head_bwd')
    return bool(done.value)


def head_bwd_io_supported(x, dx, Cout):
    return bool(_lib.load().mt_head_bwd_io_supported(x.dt, x.cs, dx.dt, dx.cs, x.C, int(Cout)))


def head_bwd_workspace(N, V, Cin, Cout):
    return _lib.load().mt_head_bwd_workspace(int(N), int(V), int(Cin), int(Cout))


def pointwise_stats_blocks(p):
    return _lib.load().mt_pointwise_stats_blocks(C.byref(p))


def inorm_finalize(part, N, nsb, Cn, count, gamma, beta, eps, mean, rstd, scale, shift):
    _lib.check(_lib.load().mt_inorm_finalize(_ptr(part), N, nsb, Cn, float(count), _ptr(gamma), _ptr(beta), float(eps),
                                             _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(shift), _stream()), 'inorm_finalize')


def inorm_lrelu_apply(y, out, res=None):
    """out = lrelu_slope(y*scale+shift [+ res as lazy act]) materialised; y/res/out are Act."""
    _lib.check(_lib.load().mt_inorm_lrelu_apply(
        C.c_void_p(y.data_ptr()), y.cs, _ptr(y.scale), _ptr(y.shift), y.slope,
        C.c_void_p(res.data_ptr()) if res is not None else None, res.cs if res is not None else 0,
        _ptr(res.scale) if res is not None else None, _ptr(res.shift) if res is not None else None,
        res.slope if res is not None else 1.0,
        C.c_void_p(out.data_ptr()), out.cs, y.N, y.V, y.C, _same_dt(y, out, res), _stream()), 'inorm_lrelu_apply')


def _same_dt(*acts):
    """the streaming kernels take ONE storage type for all their activation operands and one for all their gradient operands (the
    engine chooses them per resolution level)"""
    dts = {a.dt for a in acts if a is not None}
    if len(dts) != 1:
        raise RuntimeError("operands of a streaming kernel must share one storage type (got %s): convert with ops.cast" % sorted(dts))
    return dts.pop()


def inorm_bwd_workspace(N, V, Cn):
    return _lib.load().mt_inorm_bwd_workspace(N, V, Cn)


def inorm_lrelu_bwd(g, y, gamma, beta, dgamma, dbeta, dbias, ws, part=None, part_c0=0):
    """g (Act over the gradient buffer, in place -> dy); y: Act with mean/rstd/slope of the forward.  part: [N, nblk, cs, 2] first-pass
    partials written by the convolution that produced g (mt_conv3d_t.bstats), this layer's channels at columns part_c0 .."""
    if part is not None:
        assert part.dim() == 4 and part.shape[0] == y.N and part.shape[3] == 2 and part.is_contiguous()
    _lib.check(_lib.load().mt_inorm_lrelu_bwd(
        C.c_void_p(g.data_ptr()), g.cs, C.c_void_p(y.data_ptr()), y.cs, _ptr(y.mean), _ptr(y.rstd), _ptr(gamma), _ptr(beta),
        y.slope, y.N, y.V, y.C, _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(part), int(part.shape[1]) if part is not None else 0,
        int(part.shape[2]) if part is not None else 0, int(part_c0), _ptr(ws), ws.numel() * ws.element_size(), g.dt, y.dt, _stream()),
        'inorm_lrelu_bwd')


def conv_bwd_stats_supported(p):
    return bool(_lib.load().mt_conv3d_bwd_stats_supported(C.byref(p)))


def set_bwd_stats(p, y_act, gamma, beta, c0):
    """fused first pass of the InstanceNorm backward of `y_act`'s layer in the epilogue of the convolution p (the last writer of that
    layer's output gradient): see mt_bwd_stats_t.  The caller keeps the tensors alive and provides p.stats_part."""
    assert y_act.dt == _lib.MT_F32, "the fused norm-backward statistics read an fp32 y"
    b = p.bstats
    b.y, b.ycs, b.c0, b.C = y_act.buf.data_ptr() + 4 * y_act.c0, y_act.cs, int(c0), y_act.C
    b.mean, b.rstd = y_act.mean.data_ptr(), y_act.rstd.data_ptr()
    b.gamma = gamma.data_ptr() if gamma is not None else None
    b.beta = beta.data_ptr() if beta is not None else None
    b.slope = float(y_act.slope)


def channel_sum_workspace(N, V, Cn):
    return _lib.load().mt_channel_sum_workspace(N, V, Cn)


def channel_sum(x, out, accumulate, ws):
    _lib.check(_lib.load().mt_channel_sum(C.c_void_p(x.data_ptr()), x.cs, x.N, x.V, x.C, _ptr(out), int(accumulate), _ptr(ws),
                                          ws.numel() * ws.element_size(), x.dt, _stream()), 'channel_sum')


def loss_workspace(B, V, Cn):
    return _lib.load().mt_loss_workspace(B, V, Cn)


def multitalent_loss_fwd(logits, target, valid, lut, stats, ws):
    _lib.check(_lib.load().mt_multitalent_loss_fwd(C.c_void_p(logits.data_ptr()), logits.cs, _ptr(target), logits.N, logits.V,
                                                   logits.C, _ptr(valid), _ptr(lut), _ptr(stats), _ptr(ws),
                                                   ws.numel() * ws.element_size(), _stream()), 'multitalent_loss_fwd')


def multitalent_loss_bwd(logits, target, valid, lut, gstats, dlogits):
    _lib.check(_lib.load().mt_multitalent_loss_bwd(C.c_void_p(logits.data_ptr()), logits.cs, _ptr(target), logits.N, logits.V,
                                                   logits.C, _ptr(valid), _ptr(lut), _ptr(gstats),
                                                   C.c_void_p(dlogits.data_ptr()), dlogits.cs, _stream()),
               'multitalent_loss_bwd')


def multitalent_hard_stats(logits, target, valid, lut, stats):
    """stats [B, C, 3] = exact (tp, fp, fn) counts of sigmoid(logits) > 0.5 per valid region (online evaluation)."""
    lib = _lib.load()
    n = lib.mt_hard_stats_workspace(logits.N, logits.C)
    ws = torch.empty((n + 7) // 8, dtype=torch.int64, device=stats.device)
    _lib.check(lib.mt_multitalent_hard_stats(C.c_void_p(logits.data_ptr()), logits.cs, _ptr(target), logits.N, logits.V, logits.C,
                                             _ptr(valid), _ptr(lut), _ptr(stats), _ptr(ws), ws.numel() * 8, _stream()),
               'multitalent_hard_stats')


def softmax_dice_ce_fwd(logits, target, stats, ws):
    _lib.check(_lib.load().mt_softmax_dice_ce_fwd(C.c_void_p(logits.data_ptr()), logits.cs, _ptr(target), logits.N, logits.V,
                                                  logits.C, _ptr(stats), _ptr(ws), ws.numel() * ws.element_size(), _stream()),
               'softmax_dice_ce_fwd')


def softmax_dice_ce_bwd(logits, target, gstats, dlogits):
    _lib.check(_lib.load().mt_softmax_dice_ce_bwd(C.c_void_p(logits.data_ptr()), logits.cs, _ptr(target), logits.N, logits.V,
                                                  logits.C, _ptr(gstats),
                                                  C.c_void_p(dlogits.data_ptr()), dlogits.cs, _stream()), 'softmax_dice_ce_bwd')


LOSS_CE_ALL_CHANNELS, LOSS_DICE_OVER_BATCH = 1, 2


def loss_combine(stats, dice, dice_stride, ce_coef, dice_coef, flags, c0, smooth_num, smooth_den, den_eps, clamp_min, out3, gstats, dice_grad_scale=1.0):
    """stats / gstats [L, B, C, 4]; dice = data pointer tensor of the (tp, fp, fn) the ratios are formed from (see mtseg.h)."""
    L, B, Cn = stats.shape[:3]
    _lib.check(_lib.load().mt_loss_combine(_ptr(stats), C.c_void_p(dice.data_ptr()), int(dice_stride), L, B, Cn, _ptr(ce_coef),
                                           _ptr(dice_coef), int(flags), int(c0), float(smooth_num), float(smooth_den), float(den_eps),
                                           float(clamp_min), float(dice_grad_scale), _ptr(out3), _ptr(gstats), _stream()), 'loss_combine')


def sumsq(x, out, ws):
    _lib.check(_lib.load().mt_sumsq(_ptr(x), x.numel(), _ptr(out), _ptr(ws), ws.numel() * ws.element_size(), _stream()), 'sumsq')


def sumsq_workspace(n):
    return _lib.load().mt_sumsq_workspace(n)


def sgd_nesterov(p, g, buf, lr, wd, mom, first_step, sumsq_dev, max_norm):
    _lib.check(_lib.load().mt_sgd_nesterov(_ptr(p), _ptr(g), _ptr(buf), p.numel(), float(lr), float(wd), float(mom),
                                           int(first_step), _ptr(sumsq_dev), float(max_norm), _stream()), 'sgd_nesterov')


def flip_accumulate(logits, flips, nonlin, weight, acc, first):
    D, H, W = logits.spatial
    _lib.check(_lib.load().mt_flip_accumulate(C.c_void_p(logits.data_ptr()), logits.cs, D, H, W, logits.C, int(flips[0]),
                                              int(flips[1]), int(flips[2]), int(nonlin), float(weight), _ptr(acc), int(first),
                                              _stream()), 'flip_accumulate')


def head_flip_accumulate(p, sample, flips, nonlin, weight, acc, first):
    _lib.check(_lib.load().mt_head_flip_accumulate(C.byref(p), int(sample), int(flips[0]), int(flips[1]), int(flips[2]), int(nonlin),
                                                   float(weight), _ptr(acc), int(first), _stream()), 'head_flip_accumulate')


def head_mirror_accumulate(p, sample0, flips, nonlin, weight, gauss, agg, nb, agg_shape, origin):
    """flips: list of (fD, fH, fW) booleans, one per sample (the mirror combinations of ONE tile)."""
    fl = (C.c_int32 * len(flips))(*[int(f[0]) | (int(f[1]) << 1) | (int(f[2]) << 2) for f in flips])
    _lib.check(_lib.load().mt_head_mirror_accumulate(C.byref(p), int(sample0), len(flips), C.cast(fl, C.c_void_p), int(nonlin), float(weight),
                                                     _ptr(gauss), _ptr(agg), _ptr(nb), agg_shape[0], agg_shape[1], agg_shape[2],
                                                     origin[0], origin[1], origin[2], _stream()), 'head_mirror_accumulate')


def extract_tiles(vol, patch, tiles, out):
    """vol [C,X,Y,Z] device tensor; tiles: list of ((x0,y0,z0), (fD,fH,fW)); out [len(tiles), C, *patch] — see mt_extract_tiles."""
    _check_dev(vol, out)
    assert vol.is_contiguous() and out.is_contiguous() and vol.dtype == torch.float32 and out.dtype == torch.float32
    flat = []
    for (x0, y0, z0), f in tiles:
        flat += [int(x0), int(y0), int(z0), int(f[0]) | (int(f[1]) << 1) | (int(f[2]) << 2)]
    desc = (C.c_int32 * len(flat))(*flat)
    _lib.check(_lib.load().mt_extract_tiles(_ptr(vol), vol.shape[0], vol.shape[1], vol.shape[2], vol.shape[3], _ptr(out), len(tiles),
                                            patch[0], patch[1], patch[2], C.cast(desc, C.c_void_p), _stream()), 'extract_tiles')
    return out


def tile_accumulate(acc, gauss, Cn, patch, agg, nb, agg_shape, origin):
    _lib.check(_lib.load().mt_tile_accumulate(_ptr(acc), _ptr(gauss), Cn, patch[0], patch[1], patch[2], _ptr(agg), _ptr(nb),
                                              agg_shape[0], agg_shape[1], agg_shape[2], origin[0], origin[1], origin[2],
                                              _stream()), 'tile_accumulate')


def normalize_threshold(agg, nb, Cn, V, class_order, use_regions, seg):
    _lib.check(_lib.load().mt_normalize_threshold(_ptr(agg), _ptr(nb), Cn, V, _ptr(class_order), int(use_regions), _ptr(seg),
                                                  _stream()), 'normalize_threshold')


def ncdhw_to_ndhwc(x, out=None, out_act=None):
    """x [N,C,D,H,W] contiguous -> NDHWC.  Writes into out_act (Act slice) if given."""
    N, Cn = x.shape[0], x.shape[1]
    V = x.shape[2] * x.shape[3] * x.shape[4]
    if out_act is None:
        if out is None:
            out = torch.empty((N,) + tuple(x.shape[2:]) + (Cn,), dtype=torch.float32, device=x.device)
        out_act = Act(out)
    _lib.check(_lib.load().mt_ncdhw_to_ndhwc(_ptr(x), C.c_void_p(out_act.data_ptr()), N, Cn, V, out_act.cs, _stream()),
               'ncdhw_to_ndhwc')
    return out_act.buf


def ndhwc_to_ncdhw(a, out=None):
    N, Cn = a.N, a.C
    D, H, W = a.spatial
    if out is None:
        out = torch.empty((N, Cn, D, H, W), dtype=torch.float32, device=a.buf.device)
    _lib.check(_lib.load().mt_ndhwc_to_ncdhw(C.c_void_p(a.data_ptr()), a.cs, _ptr(out), N, Cn, a.V, _stream()), 'ndhwc_to_ncdhw')
    return out


def downsample_seg_nearest(seg, out_spatial, remove_minus_one=False, out=None):
    """seg: [B, C, D, H, W] float32 label maps (contiguous) -> [B, C, *out_spatial]; see mt_downsample_seg_nearest."""
    _check_dev(seg)
    assert seg.dim() == 5 and seg.dtype == torch.float32 and seg.is_contiguous()
    B, Cn, D, H, W = seg.shape
    if out is None:
        out = torch.empty((B, Cn) + tuple(int(i) for i in out_spatial), dtype=torch.float32, device=seg.device)
    _lib.check(_lib.load().mt_downsample_seg_nearest(_ptr(seg), B * Cn, D, H, W, _ptr(out), *[int(i) for i in out_spatial],
                                                    int(remove_minus_one), _stream()), 'downsample_seg_nearest')
    return out


_parse_select_env()
_select_env = _select
