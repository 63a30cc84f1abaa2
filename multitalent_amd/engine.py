"""Execution engine: turns a Generic_UNet / FabiansUNet parameter-holder module into a static program
of fused HIP ops and runs forward / backward through the C ABI (include/mtseg.h).

Python here only sequences launches and owns buffers (torch tensors); all arithmetic is in
libmtseg_hip.so.  Design points (MI355X-first):
  * activations live NDHWC in HBM as RAW conv outputs + per-(n,c) scale/shift ("lazy activations"):
    InstanceNorm+LeakyReLU is applied by the CONSUMER on load, so a conv block is one read of its input
    and one write of its output; torch.cat is never materialised (the conv reads two sources);
  * every buffer is allocated once and reused across iterations (288 GB HBM: no allocator traffic);
  * all parameters are views into ONE flat buffer laid out in backward-completion order, gradients
    likewise, so clip/SGD are single kernels and the DDP all-reduce streams contiguous slices as
    soon as they are final (overlap with the rest of backward on a side stream).
"""
import contextlib
import os

import torch
from torch import nn

from . import ops
from .ops import Act, ConvGeom

LRELU_DEFAULT = 1e-2


class Val:
    """A node output: (lazy) activation + its gradient buffer."""

    def __init__(self, name, C):
        self.name, self.C = name, C
        self.act = None
        self.grad = None
        self.grad_init = False
        self.spatial = None


class _Op:
    params = ()

    def plan(self, eng, N):  # allocate buffers for batch N, given input spatial dims are set
        raise NotImplementedError

    def param_list(self):
        return []

    def inputs(self):
        """the Vals whose gradient buffers this op's backward writes (used to find the LAST writer of a gradient)."""
        return []


def _strides(w, **kw):
    return ops.conv_weight_strides(w, **kw)


class ConvNormOp(_Op):
    """conv (+bias) -> InstanceNorm (stats in the conv epilogue) -> LeakyReLU (lazy).  With norm=None it is a
    bare conv.  `lrelu=False` gives conv->norm only (residual branch / skip projection)."""

    def __init__(self, name, srcs, out, conv, norm, lrelu=True, pointwise=False):
        self.name, self.srcs, self.out, self.conv, self.norm, self.lrelu = name, srcs, out, conv, norm, lrelu
        self.kernel = tuple(conv.kernel_size)
        self.stride = tuple(conv.stride)
        self.pad = tuple(conv.padding)
        self.slope = LRELU_DEFAULT
        self.pointwise = pointwise   # 1x1x1 conv on the pointwise kernel (heads)
        # strided 1x1x1 skip projection of a residual block (conv_blocks.py:159-165): forward = the pointwise kernel gathering every
        # stride-th voxel, backward-data = the pointwise kernel scattering to those voxels (mt_pointwise_t.scatter), backward-weight =
        # the tiled kernel.  (Until round 3 all three ran on the runtime-geometry convolution kernels.)
        self._pw_strided_geom = (not pointwise and self.kernel == (1, 1, 1) and self.stride != (1, 1, 1) and self.pad == (0, 0, 0)
                                 and len(srcs) == 1 and os.environ.get('MT_PW_STRIDED', '1') != '0')
        self.pw_strided = False      # decided in plan(): the input size must be a multiple of the stride (the scatter's output grid)
        self.bwd_part = None         # (partials, first column): first pass of this layer's norm backward, fused into the last writer of out.grad

    def inputs(self):
        return list(self.srcs)

    def param_list(self):
        ps = [self.conv.weight]
        if self.conv.bias is not None:
            ps.append(self.conv.bias)
        if self.norm is not None:
            ps += [self.norm.weight, self.norm.bias]
        return ps

    def plan(self, eng, N):
        dev = eng.device
        sp = self.srcs[0].spatial
        self.geom = ConvGeom(sp, self.kernel, self.stride, self.pad)
        self.out.spatial = self.geom.out
        Cout = self.conv.out_channels
        self.pw_strided = self._pw_strided_geom and tuple(o * st for o, st in zip(self.geom.out, self.stride)) == tuple(self.geom.inp)
        self.mma = eng.op_mma(self.geom.out)          # bf16 matrix inputs for this layer (forward, backward-data and backward-weight)
        buf = eng.buffer(self.name + '.y', (N,) + self.geom.out + (Cout,), self._out_dtype(eng))
        if self.norm is not None:
            st = eng.buffer(self.name + '.stats', (4, N, Cout))
            self.out.act = Act(buf, scale=st[2], shift=st[3], slope=self.slope if self.lrelu else 1.0, mean=st[0], rstd=st[1])
        else:
            self.out.act = Act(buf)
        self.part = None
        self.wf = self.wb = None

    def _out_dtype(self, eng):
        return eng.val_dtype(self.geom.out)

    def _fwd_io(self, eng):
        """storage types of the forward launch: the operands as they are where the kernel takes them, fp32 copies elsewhere (Engine.io)"""
        Cout = self.conv.out_channels
        acts = [s.act for s in self.srcs]
        if self.pointwise or self.pw_strided:
            def build(ins, outs):
                a = ins[0]
                return ops.fill_pointwise(a, self.geom.out, a.spatial, self.stride, (1, 1, 1), Cout, eng.dummy, self.conv.bias, outs[0],
                                          mma=eng.mma if self.pointwise else 0)
            return eng.io(self.name + '.fwd', acts, [self.out.act], build, ops.pointwise_io_supported)
        build = lambda ins, outs: ops.fill_conv(ins, self.geom, Cout, bias=self.conv.bias, out0=outs[0], mma=self.mma)
        return eng.io(self.name + '.fwd', acts, [self.out.act], build, ops.conv_io_supported)

    def _fwd_params(self, eng, io=None):
        Cout = self.conv.out_channels
        io = io if io is not None else self._fwd_io(eng)
        p = io.build()
        if self.norm is not None and self.part is None:
            nsb = ops.pointwise_stats_blocks(p) if (self.pointwise or self.pw_strided) else ops.conv_stats_blocks(p)
            self.part = eng.buffer(self.name + '.part', (io.ins[0].N, nsb, Cout, 2))
        if self.part is not None:
            p.stats_part = self.part.data_ptr()
        return p

    def pack(self, eng, need_bwd):
        w = self.conv.weight
        Cout = self.conv.out_channels
        C0 = self.srcs[0].C
        C1 = self.srcs[1].C if len(self.srcs) > 1 else 0
        if self.pointwise:
            self.wf = ops.pack_conv_weights(w, C0, 0, Cout, (1, 1, 1), _strides(w), False, ops.POINTWISE_CK, out=self.wf)
            # training forward in mixed precision: fp16 products where the head kernel has them (the fused inference heads keep `wf`)
            lay = ops.pointwise_pack_layout(self._fwd_io(eng).build())
            self.wf16 = ops.pack_conv_weights(w, C0, 0, Cout, (1, 1, 1), _strides(w), False, ops.POINTWISE_CK, out=getattr(self, 'wf16', None),
                                              layout=lay) if lay != 1 else None
            if need_bwd and self.srcs[0].grad is not None and self.stride == (1, 1, 1):
                self.wb = ops.pack_conv_weights(w, Cout, 0, C0, (1, 1, 1), _strides(w, as_bwd_data=True), False, ops.POINTWISE_CK, out=self.wb)
            return
        if self.pw_strided:
            self.wf = ops.pack_conv_weights(w, C0, 0, Cout, (1, 1, 1), _strides(w), False, ops.POINTWISE_CK, out=self.wf)
            if need_bwd and self.srcs[0].grad is not None:
                self.wb = ops.pack_conv_weights(w, Cout, 0, C0, (1, 1, 1), _strides(w, as_bwd_data=True), False, ops.POINTWISE_CK, out=self.wb)
            return
        p = self._fwd_params(eng)
        self.ck_f = ops.conv_ck(p)
        self.wf = ops.pack_conv_weights(w, C0, C1, Cout, self.kernel, _strides(w), False, self.ck_f, out=self.wf,
                                        layout=ops.conv_pack_layout(p))
        if need_bwd and any(s.grad is not None for s in self.srcs):
            if self._use_strided_bwd(eng):
                self.wb = ops.pack_conv_weights(w, Cout, 0, C0, self.kernel, _strides(w, as_bwd_data=True), False, 16, out=self.wb,
                                                layout=ops.conv_bwd_data_strided_pack_layout(self._strided_bwd_io(eng, None).build()))
                return
            if self._use_parity_classes():
                cls = self._parity_classes()
                if self.wb is None:
                    self.wb = [None] * len(cls)
                for i, (geomc, place, tapmap) in enumerate(cls):
                    pb = ops.fill_conv([eng.as_fp32(Act(self.out.act.buf), geometry_only=True)], geomc, C0, place=place, mma=self.mma)
                    self.wb[i] = ops.pack_conv_weights(w, Cout, 0, C0, geomc.k, _strides(w, as_bwd_data=True), False,
                                                       ops.conv_ck(pb), out=self.wb[i], tapmap=tapmap)
                return
            pb = self._bwd_data_io(eng, None).build()     # the kernel choice (and with it the packed layout) must match the launch in backward()
            self.ck_b = ops.conv_ck(pb)
            self.wb = ops.pack_conv_weights(w, Cout, 0, C0 + C1, self.kernel, _strides(w, as_bwd_data=True), True, self.ck_b, out=self.wb,
                                            layout=ops.conv_pack_layout(pb))

    def _strided_bwd_io(self, eng, g):
        """mt_conv3d_bwd_data_strided takes the FORWARD geometry with src[0] = dY and out0 = dX."""
        s0 = self.srcs[0]
        gact = Act(g) if g is not None else Act(self._grad_like(eng, self.out))
        dx = Act(s0.grad) if s0.grad is not None else Act(self._grad_like(eng, s0))

        def build(ins, outs):
            p = ops.fill_conv(ins, self.geom, self.conv.out_channels, out0=outs[0], mma=self.mma)
            p.Cin = s0.C
            return p
        return eng.io(self.name + '.bwdd', [gact], [dx], build, ops.conv_bwd_data_strided_io_supported, grad_ins=True)

    def _strided_bwd_params(self, g, eng=None):
        return self._strided_bwd_io(eng if eng is not None else self._eng, g).build()

    @staticmethod
    def _grad_like(eng, val):
        """a tensor with the shape and storage type of val's gradient (geometry / kernel-choice queries before the buffers exist)"""
        if val.grad is not None:
            return val.grad
        return eng.buffer(val.name + '.grad', tuple(val.act.buf.shape[:4]) + (val.C,), eng.grad_dtype(val.spatial))

    def _use_strided_bwd(self, eng):
        if self.stride == (1, 1, 1) or len(self.srcs) != 1 or self.pointwise:
            return False
        return ops.conv3d_bwd_data_strided_supported(self._strided_bwd_io(eng, None).build())

    def _use_parity_classes(self):
        return self.stride != (1, 1, 1) and len(self.srcs) == 1 and not self.pointwise

    def _parity_classes(self):
        return ops.bwd_data_parity_classes(self.geom)

    def forward(self, eng):
        io = self._fwd_io(eng)
        p = self._fwd_params(eng, io)
        wf = self.wf
        if self.pointwise and ops.pointwise_pack_layout(p) != 1:
            wf = getattr(self, 'wf16', None)
            assert wf is not None, "head: fp16-product launch without fp16 weights (repack after a precision change)"
        p.wpack = wf.data_ptr()
        io.pre()
        if self.pointwise or self.pw_strided:
            ops.pointwise_fwd(p)
        else:
            ops.conv3d_fwd(p)
        io.post()
        if self.norm is not None:
            a = self.out.act
            nsb = self.part.shape[1]
            ops.inorm_finalize(self.part, a.N, nsb, a.C, a.V, self.norm.weight, self.norm.bias, self.norm.eps,
                               a.mean, a.rstd, a.scale, a.shift)

    def _bwd_data_io(self, eng, g):
        """backward-data as a stride-1 convolution of dY with the flipped weights; destinations = the sources' gradient buffers"""
        Cin = sum(s.C for s in self.srcs)
        geomT = ConvGeom(self.geom.out, self.kernel, (1, 1, 1), tuple(k - 1 - p for k, p in zip(self.kernel, self.pad)),
                         dil=self.stride, out_spatial=self.geom.inp)
        gact = Act(g) if g is not None else Act(self._grad_like(eng, self.out))  # geometry-only when g is None
        outs = [Act(self._grad_like(eng, s)) for s in self.srcs]

        def build(ins, outs_):
            return ops.fill_conv(ins, geomT, Cin, out0=outs_[0], out1=outs_[1] if len(outs_) > 1 else None,
                                 csplit=self.srcs[0].C if len(outs_) > 1 else None, mma=self.mma)
        return eng.io(self.name + '.bwdd', [gact], outs, build, ops.conv_io_supported, grad_ins=True)

    def backward(self, eng):
        g = self.out.grad
        assert g is not None and self.out.grad_init, "gradient of %s was never produced" % self.name
        gact = Act(g)
        dbias = eng.grad_of(self.conv.bias) if self.conv.bias is not None else None
        if self.norm is not None:
            a = self.out.act
            ws = eng.workspace(ops.inorm_bwd_workspace(a.N, a.V, a.C))
            part, part_c0 = self.bwd_part if self.bwd_part is not None else (None, 0)
            self.bwd_part = None
            ops.inorm_lrelu_bwd(gact, a, self.norm.weight, self.norm.bias, eng.grad_of(self.norm.weight),
                                eng.grad_of(self.norm.bias), dbias, ws, part=part, part_c0=part_c0)
        elif dbias is not None:
            ws = eng.workspace(ops.channel_sum_workspace(gact.N, gact.V, gact.C))
            ops.channel_sum(gact, dbias, False, ws)
        # backward-weight straight into the flat gradient buffer (torch parameter layout)
        dw = eng.grad_of(self.conv.weight)
        with eng.weight_stream() as side:
            iow = eng.io(self.name + '.bwdw', [s.act for s in self.srcs] + [gact], [],
                         lambda ins, outs: (ops.fill_conv(ins[:-1], self.geom, self.conv.out_channels, mma=self.mma), ins[-1]),
                         lambda py: ops.conv_bwd_weight_io_supported(py[0], py[1]), grad_ins=[False] * len(self.srcs) + [True])
            pw, yact = iow.build()
            iow.pre()
            ws = eng.workspace(ops.conv3d_bwd_weight_workspace(pw), side=side)
            ops.conv3d_bwd_weight(pw, yact, dw, _strides(self.conv.weight), False, ws)
        # backward-data into the sources' gradient buffers
        dsts = [s for s in self.srcs if s.grad is not None]
        if not dsts:
            return
        assert len(dsts) == len(self.srcs)
        inits = {s.grad_init for s in self.srcs}
        assert len(inits) == 1, "mixed accumulate state on the sources of %s" % self.name
        acc = inits.pop()
        if self.pointwise:
            if self.stride != (1, 1, 1):
                raise NotImplementedError("backward-data of a strided pointwise conv is handled by ResBlockOp")
            s0 = self.srcs[0]
            io = eng.io(self.name + '.bwdd', [gact], [Act(s0.grad)],
                        lambda ins, outs: ops.fill_pointwise(ins[0], self.geom.out, self.geom.out, (1, 1, 1), (1, 1, 1), s0.C, self.wb, None, outs[0]),
                        ops.pointwise_io_supported, grad_ins=True)
            p = io.build()
            p.accumulate = io.accumulate(acc)
            io.pre()
            ops.pointwise_fwd(p)
            io.post(acc)
        elif self.pw_strided:
            # dX[stride * m] (+)= W^T dY[m]; the other voxels of X receive nothing from this convolution
            s0 = self.srcs[0]
            if not acc:
                s0.grad.zero_()
                acc = True

            def build(ins, outs):
                p = ops.fill_pointwise(ins[0], self.geom.out, self.geom.out, (1, 1, 1), self.stride, s0.C, self.wb, None, outs[0])
                p.scatter = 1
                return p
            io = eng.io(self.name + '.bwdd', [gact], [Act(s0.grad)], build, ops.pointwise_io_supported, grad_ins=True)
            p = io.build()
            p.accumulate = io.accumulate(acc)
            for t, _ in io._cout:
                t.buf.zero_()                    # (an fp32 copy of the destination: the scatter only writes every stride-th voxel)
            io.pre()
            ops.pointwise_fwd(p)
            io.post(acc)
        elif self._use_strided_bwd(eng):
            io = self._strided_bwd_io(eng, g)
            p = io.build()
            p.wpack = self.wb.data_ptr()
            p.accumulate = io.accumulate(acc)
            io.pre()
            ops.conv3d_bwd_data_strided(p)
            io.post(acc)
        elif self._use_parity_classes():
            s0 = self.srcs[0]
            cls = self._parity_classes()
            full = 1
            for st in self.stride:
                full *= st
            if len(cls) < full and not acc:      # some input positions receive no gradient from this conv
                s0.grad.zero_()
                acc = True
            # (runtime-geometry kernel with strided output placement: fp32 operands; other storage types through fp32 copies)
            gin = eng.as_fp32(gact, reuse=False)
            dx = Act(s0.grad) if s0.grad.dtype == torch.float32 else Act(eng.buffer(self.name + '.bwdd.out0', tuple(s0.grad.shape)))
            if dx.buf is not s0.grad and acc:
                ops.cast(Act(s0.grad), dx)
            for (geomc, place, _), wb in zip(cls, self.wb):
                p = ops.fill_conv([gin], geomc, s0.C, wpack=wb, out0=dx, accumulate=acc, place=place, mma=self.mma)
                ops.conv3d_fwd(p)
            if dx.buf is not s0.grad:
                ops.cast(dx, Act(s0.grad))
        else:
            io = self._bwd_data_io(eng, g)
            p = io.build()
            p.wpack = self.wb.data_ptr()
            p.accumulate = io.accumulate(acc)
            if io.native:
                self._fuse_norm_bwd_stats(eng, p)
            io.pre()
            ops.conv3d_fwd(p)
            io.post(acc)
        for s in self.srcs:
            s.grad_init = True

    def _fuse_norm_bwd_stats(self, eng, p):
        """When this launch is the LAST writer of a source's gradient and that source is the output of a conv + InstanceNorm layer,
        the kernel also emits the first pass of that layer's norm backward (sum dz, sum dz zhat per block; mt_bwd_stats_t): the
        separate reduction over (g, y) — 45 % of the norm backward's time — disappears.  One source per launch."""
        if eng.fuse_norm_bwd not in (1, 2):
            return
        c0 = 0
        for s in self.srcs:
            prod = eng.producer.get(id(s))
            if (eng.pending.get(id(s), 0) == 1 and isinstance(prod, ConvNormOp) and not isinstance(prod, HeadOp) and prod.norm is not None
                    and s.act.mean is not None and ops.conv_bwd_stats_supported(p)):
                nsb = ops.conv_stats_blocks(p)
                part = eng.buffer(self.name + '.bwdpart', (s.act.N, nsb, int(p.Cout), 2))
                p.stats_part = part.data_ptr()
                ops.set_bwd_stats(p, s.act, prod.norm.weight, prod.norm.bias, c0)
                prod.bwd_part = (part, c0)
                return
            c0 += s.C


class TConvOp(_Op):
    """nn.ConvTranspose3d(kernel == stride, bias=False) (generic_UNet.py:335-336)."""

    def __init__(self, name, src, out, tu):
        self.name, self.src, self.out, self.tu = name, src, out, tu
        self.k = tuple(tu.kernel_size)
        assert tuple(tu.stride) == self.k and tu.bias is None

    def param_list(self):
        return [self.tu.weight]

    def inputs(self):
        return [self.src]

    def plan(self, eng, N):
        sp = self.src.spatial
        self.out.spatial = tuple(a * b for a, b in zip(sp, self.k))
        Cout = self.tu.out_channels
        self.out.act = Act(eng.buffer(self.name + '.y', (N,) + self.out.spatial + (Cout,), eng.val_dtype(self.out.spatial)))
        self.wf = self.wb = None

    def pack(self, eng, need_bwd):
        w = self.tu.weight
        Cin, Cout = self.tu.in_channels, self.tu.out_channels
        self.wf_layout = ops.pointwise_pack_layout(self._fwd_io(eng, eng.dummy).build())     # 4: fp16 products (mixed precision), else 1
        self.wf = ops.pack_conv_weights(w, Cin, 0, Cout, self.k, _strides(w, transposed_layout=True), False, ops.POINTWISE_CK, out=self.wf,
                                        layout=self.wf_layout)
        if need_bwd and self.src.grad is not None:
            p = self._bwd_io(eng).build()
            self.ck_b = ops.conv_ck(p)
            self.wb = ops.pack_conv_weights(w, Cout, 0, Cin, self.k, _strides(w, transposed_layout=True, as_bwd_data=True),
                                            False, self.ck_b, out=self.wb, layout=ops.conv_pack_layout(p))

    def _fwd_io(self, eng, wf):
        return eng.io(self.name + '.fwd', [self.src.act], [self.out.act],
                      lambda ins, outs: ops.fill_pointwise(ins[0], ins[0].spatial, ins[0].spatial, (1, 1, 1), self.k, self.tu.out_channels, wf, None, outs[0],
                                                           mma=eng.mma),
                      ops.pointwise_io_supported)

    def forward(self, eng):
        io = self._fwd_io(eng, self.wf)
        p = io.build()
        assert ops.pointwise_pack_layout(p) == self.wf_layout, "transposed conv: the packed weights do not match the launch (repack after a precision change)"
        io.pre()
        ops.pointwise_fwd(p)
        io.post()

    def _geom(self):
        return ConvGeom(self.out.spatial, self.k, self.k, (0, 0, 0))   # dX = conv(k = stride = pool kernel, pad 0) of dOut

    def _bwd_io(self, eng):
        g = ConvNormOp._grad_like(eng, self.out)
        dx = ConvNormOp._grad_like(eng, self.src)

        def build(ins, outs):
            p = ops.fill_conv(ins, self._geom(), self.tu.in_channels, out0=outs[0], mma=eng.mma)
            p.csplit = self.tu.in_channels
            return p
        return eng.io(self.name + '.bwdd', [Act(g)], [Act(dx)], build, ops.conv_io_supported, grad_ins=True)

    def backward(self, eng):
        g = self.out.grad
        assert g is not None and self.out.grad_init
        w = self.tu.weight
        # backward-weight: X = dOut (channels = Cout_t), Y = tconv input (lazy act, channels = Cin_t)
        with eng.weight_stream() as side:
            iow = eng.io(self.name + '.bwdw', [Act(g), self.src.act], [],
                         lambda ins, outs: (ops.fill_conv(ins[:1], self._geom(), self.tu.in_channels, mma=eng.mma), ins[1]),
                         lambda py: ops.conv_bwd_weight_io_supported(py[0], py[1]), grad_ins=[True, False])
            pw, yact = iow.build()
            iow.pre()
            ws = eng.workspace(ops.conv3d_bwd_weight_workspace(pw), side=side)
            ops.conv3d_bwd_weight(pw, yact, eng.grad_of(w), _strides(w, transposed_layout=True, as_bwd_data=True), False, ws)
        if self.src.grad is not None:
            io = self._bwd_io(eng)
            p = io.build()
            p.wpack = self.wb.data_ptr()
            p.accumulate = io.accumulate(self.src.grad_init)
            io.pre()
            ops.conv3d_fwd(p)
            io.post(self.src.grad_init)
            self.src.grad_init = True


class ResAddOp(_Op):
    """out = lrelu(main + residual) materialised (conv_blocks.py:210-213).  main = conv2->norm2 (lazy, slope 1);
    residual = block input (dense activation) or the skip projection conv->norm (lazy, slope 1)."""

    def __init__(self, name, main, res, out):
        self.name, self.main, self.res, self.out = name, main, res, out
        self.slope = LRELU_DEFAULT

    def inputs(self):
        return [self.main, self.res]

    def plan(self, eng, N):
        self.out.spatial = self.main.spatial
        self.out.act = Act(eng.buffer(self.name + '.a', (N,) + self.main.spatial + (self.main.C,), eng.val_dtype(self.main.spatial)))

    def pack(self, eng, need_bwd):
        pass

    def forward(self, eng):
        m = self.main.act
        y = Act(m.buf, scale=m.scale, shift=m.shift, slope=self.slope)   # outer lrelu slope; inner affine from norm2
        ops.inorm_lrelu_apply(y, self.out.act, res=self.res.act)

    def backward(self, eng):
        g = self.out.grad
        assert g is not None and self.out.grad_init
        m, r = self.main.act, self.res.act
        # g <- g * lrelu'(t), t = main + residual; the same tensor is the gradient of both branches
        from . import _lib
        import ctypes as C
        gm = self.main.grad
        # main branch gets its own buffer (it is transformed in place by the norm backward).  When the main branch is conv -> norm
        # (always, in the residual encoder) the kernel also takes the first pass of that norm's backward over the gradient it is
        # producing (mt_lrelu_bwd_stats): the norm backward's own reduction over (g', y) disappears.
        prod = eng.producer.get(id(self.main)) if eng.fuse_norm_bwd in (1, 3) else None
        nblk = _lib.load().mt_lrelu_bwd_stats_blocks(m.V, m.C)
        gdt, ydt = ops._same_dt(Act(g), Act(gm)), ops._same_dt(m, r)      # one resolution level: one gradient type, one activation type
        if (prod is not None and isinstance(prod, ConvNormOp) and prod.norm is not None and m.mean is not None and nblk > 0
                and g.shape[4] == m.C and gm.shape[4] == m.C and m.cs == m.C and r.cs == m.C):
            part = eng.buffer(self.name + '.bwdpart', (m.N, nblk, m.C, 2))
            _lib.check(_lib.load().mt_lrelu_bwd_stats(
                C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()), ops._ptr(m.scale), ops._ptr(m.shift), self.slope,
                C.c_void_p(r.data_ptr()), ops._ptr(r.scale), ops._ptr(r.shift), r.slope, C.c_void_p(gm.data_ptr()),
                ops._ptr(m.mean), ops._ptr(m.rstd), ops._ptr(part), m.N, m.V, m.C, gdt, ydt, ops._stream()), 'lrelu_bwd_stats')
            prod.bwd_part = (part, 0)
        else:
            _lib.check(_lib.load().mt_lrelu_bwd(
                C.c_void_p(g.data_ptr()), g.shape[4], C.c_void_p(m.data_ptr()), m.cs, ops._ptr(m.scale), ops._ptr(m.shift), self.slope,
                C.c_void_p(r.data_ptr()), r.cs, ops._ptr(r.scale), ops._ptr(r.shift), r.slope,
                C.c_void_p(gm.data_ptr()), gm.shape[4], m.N, m.V, m.C, gdt, ydt, ops._stream()), 'lrelu_bwd')
        self.main.grad_init = True
        # residual branch: g itself (already masked) is added to / becomes the residual's gradient
        if self.res.grad is not None:
            if self.res.grad.data_ptr() == g.data_ptr():        # shared buffer (Engine._plan): g, masked in place, is already there
                assert not self.res.grad_init, "gradient of %s was written before its residual add ran backward" % self.res.name
                self.res.grad_init = True
            elif self.res.grad_init:
                self.res.grad.add_(g)
            else:
                self.res.grad.copy_(g)
                self.res.grad_init = True


class HeadOp(ConvNormOp):
    """1x1x1 segmentation head (generic_UNet.py:349-351; generic_modular_UNet.py:244,251)."""

    def __init__(self, name, src, out, conv):
        super().__init__(name, [src], out, conv, None, lrelu=False, pointwise=True)

    def _out_dtype(self, eng):
        return torch.float32            # logits and their gradient stay fp32 (the losses are fp32 in the reference's autocast mode too)

    def backward(self, eng):
        """dX, dW and dbias of the head in ONE pass over (x, dlogits) (mt_head_bwd) when the head is narrow enough (<= 64 channels
        in and out: the full-resolution heads, where 47 logit channels make the separate kernels cost 3 ms of a Task100 step);
        otherwise the generic pointwise backward-data + tiled backward-weight of ConvNormOp."""
        import os
        s0 = self.srcs[0]
        Cout = self.conv.out_channels
        if (os.environ.get('MT_HEAD_BWD_FUSED', '1') == '0' or s0.grad is None or self.wb is None
                or not ops.head_bwd_supported(s0.C, Cout) or self.stride != (1, 1, 1)):
            return super().backward(eng)
        g = self.out.grad
        assert g is not None and self.out.grad_init, "gradient of %s was never produced" % self.name
        gact = Act(g)
        dbias = eng.grad_of(self.conv.bias) if self.conv.bias is not None else None
        w = self.conv.weight
        st = _strides(w)                                 # (s_ci, s_co, ...) of the [Cout, Cin, 1, 1, 1] weight
        io = eng.io(self.name + '.hbwd', [s0.act], [Act(s0.grad)], lambda ins, outs: (ins[0], outs[0]),
                    lambda xo: ops.head_bwd_io_supported(xo[0], xo[1], Cout))
        a, dx = io.build()
        io.pre()
        ws = eng.workspace(ops.head_bwd_workspace(a.N, a.V, s0.C, Cout))
        done = ops.head_bwd(a, gact, self.wb, dx, bool(io.accumulate(s0.grad_init)), eng.grad_of(w), st[0], st[1], dbias, False, ws)
        io.post(s0.grad_init)
        if dbias is not None and not done:
            ws2 = eng.workspace(ops.channel_sum_workspace(gact.N, gact.V, gact.C))
            ops.channel_sum(gact, dbias, False, ws2)
        s0.grad_init = True


class _IO:
    """Storage types of ONE launch (Engine.io): the operands as the graph has them where the kernel that serves the launch takes them,
    fp32 copies (mt_cast) elsewhere.  ins / outs are the EFFECTIVE operands the launch is built with; pre() fills the input copies,
    post() writes (or accumulates) the output copies back in the tensor's own storage type."""

    def __init__(self, eng, key, ins, outs, build, flags, grad_ins):
        self.eng, self._build = eng, build
        cast_in, cast_out = flags
        self.native = not (cast_in or cast_out)
        self._cin = [(a, g) for a, g in zip(ins, grad_ins) if cast_in and a.dtype != torch.float32]
        self.ins = [eng.as_fp32(a, geometry_only=True) if (cast_in and a.dtype != torch.float32) else a for a in ins]
        self._cout = []
        self.outs = []
        for i, o in enumerate(outs):
            if cast_out and o.dtype != torch.float32:
                t = Act(eng.buffer('%s.out%d' % (key, i), tuple(o.buf.shape[:4]) + (o.C,)))
                self._cout.append((t, o))
                self.outs.append(t)
            else:
                self.outs.append(o)

    def build(self):
        return self._build(self.ins, self.outs)

    def pre(self):
        for a, is_grad in self._cin:
            self.eng.as_fp32(a, reuse=not is_grad)

    def accumulate(self, acc):
        """accumulate flag of the launch: an output that goes through an fp32 copy is written fresh and accumulated by post()"""
        return 0 if self._cout else int(bool(acc))

    def post(self, acc=False):
        for t, o in self._cout:
            ops.cast(t, o, accumulate=bool(acc))
            self.eng.io_cast_bytes += t.buf.numel() * 4


class Engine:
    def __init__(self, module, ops_list, x_val, head_vals, final_head_index):
        self.module = module
        self.ops = ops_list
        self.x = x_val
        self.heads = head_vals              # ordered like the module's forward output (highest resolution first)
        self.final_head = final_head_index
        self.device = None
        self._buffers = {}
        self._ws = None
        self._planned = None
        self._packed_version = None
        self._pack_programs = {}
        self.flat = None
        self.flat_grad = None
        self._views = {}
        self.params_version = 0
        self.wstreams = []                  # side streams of the backward-weight launches (Engine.weight_stream)
        self._wnext = 0
        self._late_pack = None
        self._mid_pack = None
        self._ws_side = {}
        self.bwdw_streams = int(os.environ.get('MT_BWDW_STREAMS', '1'))
        self.grad_ready_hook = None         # callable(lo, hi) on flat_grad element ranges, in completion order
        self.dummy = None
        self.mma = 0                        # matrix input type of the convolutions: 0 fp32, 1 bf16 (mixed precision)
        # mixed precision: ACTIVATIONS are stored as fp16 and the forward convolutions multiply fp16 operands (the reference's autocast
        # arithmetic: 11 significand bits keep the LeakyReLU decisions of the forward pass — which is what the gradient's direction
        # hangs on, DESIGN.md 3.3 — four times closer to the exact ones than bf16 does); GRADIENTS are stored as bf16 and the backward
        # convolutions multiply bf16 operands (fp32's exponent range: no loss scaling).  MT_BF16_STORAGE=0: fp32 storage with bf16
        # matrix inputs only (the mode of rounds 1-3); MT_ACT_STORAGE=bf16: bf16 activations (measured: gradient cosine 0.86 instead of
        # 0.98 on the full-size residual encoder).  bf16_min_voxels > 0 keeps the levels with fewer voxels per sample in fp32, storage
        # and arithmetic (measured: no gain in accuracy, +2.5 ms per step: default 0).
        self.storage_bf16 = os.environ.get('MT_BF16_STORAGE', '1') != '0'
        self.act_storage = {'fp16': torch.float16, 'bf16': torch.bfloat16}[os.environ.get('MT_ACT_STORAGE', 'fp16')]
        self.bf16_min_voxels = int(os.environ.get('MT_BF16_MIN_VOXELS', '0'))
        self._io_cache = {}
        self._fp32_copies = {}
        self._iter = 0
        self.io_cast_bytes = 0              # bytes moved by storage-type conversions (diagnostic: 0 when every kernel takes its operands natively)
        # first pass of a norm backward taken by the kernel that produces the gradient (ConvNormOp._fuse_norm_bwd_stats, ResAddOp.backward):
        # 0 off, 1 on, 2 convolutions only, 3 residual adds only.  Default: on only when everything runs on ONE stream — beside the
        # weight-gradient stream the separate (bandwidth-bound) reduction overlaps the (matrix-bound) backward-weight launches and
        # fusing it into the matrix-bound kernel on the chain measured 0 ... +0.3 ms per step
        # Mixed precision beside that stream: the residual-add form only (3) — the add's backward is bandwidth-bound itself, so the
        # extra sums cost nothing there (residual encoder mixed 25.48 -> 25.27 ms, fp32 unchanged: tools/r4_run47.sh)
        self._fuse_norm_bwd = os.environ.get('MT_FUSE_NORM_BWD')
        self._one_stream_at_init = self.bwdw_streams == 0       # (the default follows the stream setting the engine was built with)
        self.producer, self.pending = {}, {}

    @property
    def fuse_norm_bwd(self):
        if self._fuse_norm_bwd is not None:
            return int(self._fuse_norm_bwd)
        if self._one_stream_at_init:
            return 1
        return 3 if self.mma else 0

    @fuse_norm_bwd.setter
    def fuse_norm_bwd(self, v):
        self._fuse_norm_bwd = v

    def set_precision(self, precision):
        """'fp32' (exact, default) or 'bf16': mixed precision — the reference's autocast mode (nnUNetTrainerV2.py:236-249) on
        gfx950 terms: bf16 matrix inputs with fp32 accumulation for the 3x3x3 stride-1 convolutions (forward, backward-data and
        backward-weight), fp32 activations, master weights, gradients, normalisation, loss and optimizer; no loss scaling needed."""
        mma = {'fp32': 0, 'bf16': 1, 0: 0, 1: 1, False: 0, True: 1}[precision]
        if mma != self.mma:
            self.mma = mma
            self._replan()

    def _replan(self):
        self._planned = None                # packed layouts, statistics tilings and storage types depend on the kernel choice
        self._packed_version = None
        self._pack_programs = {}
        self._io_cache = {}
        self._fp32_copies.clear()           # keyed by data_ptr: a re-planned buffer set may recycle an address with another shape / meaning

    # ---- mixed precision: which level computes / stores what ------------------------------------------
    def op_mma(self, out_spatial):
        """matrix input type of a layer whose output has this spatial size"""
        v = out_spatial[0] * out_spatial[1] * out_spatial[2]
        return 1 if (self.mma and v >= self.bf16_min_voxels) else 0

    def val_dtype(self, spatial):
        """storage type of an activation at this spatial size"""
        v = spatial[0] * spatial[1] * spatial[2]
        return self.act_storage if (self.mma and self.storage_bf16 and v >= self.bf16_min_voxels) else torch.float32

    def grad_dtype(self, spatial):
        """storage type of the gradient of an activation at this spatial size"""
        return torch.bfloat16 if self.val_dtype(spatial) != torch.float32 else torch.float32

    def io(self, key, ins, outs, build, supported, grad_ins=False):
        """Resolve the storage types of one launch.  ins / outs: Acts as the graph has them; build(ins', outs') -> launch parameters;
        supported(params) -> does the kernel that serves them take these storage types (the *_io_supported queries of the C ABI).
        Preference: as they are; fp32 copies of the bf16 inputs; an fp32 copy of the output; both (all-fp32 is always supported)."""
        gi = list(grad_ins) if isinstance(grad_ins, (list, tuple)) else [bool(grad_ins)] * len(ins)
        sig = (key, tuple(a.dt for a in ins), tuple(o.dt for o in outs))
        flags = self._io_cache.get(sig)
        if flags is None:
            if all(a.dtype == torch.float32 for a in list(ins) + list(outs)):
                flags = (False, False)
            else:
                flags = (True, True)
                for cand in ((False, False), (True, False), (False, True)):
                    if cand[0] and all(a.dtype == torch.float32 for a in ins):
                        continue
                    if cand[1] and all(o.dtype == torch.float32 for o in outs):
                        continue
                    t = _IO(self, key, ins, outs, build, cand, gi)
                    if supported(t.build()):
                        flags = cand
                        break
                if os.environ.get('MT_IO_DEBUG') and flags != (False, False):
                    print("[mt io] %s: %s copies (in %s, out %s)" % (key, 'input' if flags == (True, False) else 'output' if flags == (False, True) else 'input+output',
                                                                     [str(a.dtype) for a in ins], [str(o.dtype) for o in outs]))
            self._io_cache[sig] = flags
        return _IO(self, key, ins, outs, build, flags, gi)

    def as_fp32(self, a, reuse=True, geometry_only=False):
        """fp32 copy of the raw values of a bf16 activation / gradient (the lazy scale / shift stay with it).  reuse: a copy made
        earlier in this iteration ON THIS STREAM is still valid (activations; never gradients, which change in place)."""
        if a.dtype == torch.float32:
            return a
        stream = torch.cuda.current_stream().cuda_stream if a.buf.is_cuda else 0
        key = (a.buf.data_ptr(), a.c0, a.C, tuple(a.buf.shape[:4]), stream)
        ent = self._fp32_copies.get(key)
        if ent is None:
            ent = self._fp32_copies[key] = [torch.empty(tuple(a.buf.shape[:4]) + (a.C,), dtype=torch.float32, device=a.buf.device), -1]
        t = a.with_buf(ent[0])
        if geometry_only:
            return t
        if not (reuse and ent[1] == self._iter):
            ops.cast(Act(a.buf, a.c0, a.C), Act(ent[0]))
            ent[1] = self._iter if reuse else -1
            self.io_cast_bytes += ent[0].numel() * 4
        return t

    # ---- memory -----------------------------------------------------------------------------------
    def buffer(self, name, shape, dtype=torch.float32):
        t = self._buffers.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buffers[name] = t
        return t

    def workspace(self, nbytes, side=None):
        """Scratch for one launch.  side = index of the weight-gradient stream the launch runs on (None: the main stream): every
        stream has its own scratch, launches of one stream are ordered."""
        n = (int(nbytes) + 3) // 4 + 16
        if side is not None:
            t = self._ws_side.get(side)
            if t is None or t.numel() < n:
                if t is not None:
                    torch.cuda.synchronize()    # an earlier launch may still be using the smaller one
                t = self._ws_side[side] = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=self.device)
            return t
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=self.device)
        return self._ws

    def _side_stream(self):
        return torch.cuda.Stream()              # (a lower or higher stream priority than the main stream's: measured no different)

    @contextlib.contextmanager
    def weight_stream(self):
        """Context of a backward-weight launch; yields the index of the side stream it runs on (None: the main stream).
        The weight gradients are off the critical path of backward — only the clip / optimizer and the gradient all-reduce read them —
        while the backward-data chain is a sequence of dependent launches whose tails, low-resolution layers and small reductions
        leave CUs idle.  With side streams (MT_BWDW_STREAMS, default 1, 0 = off; more than one measured no different) a layer's backward-weight (+ its ordered reduction)
        is issued behind an event of the main stream (its output gradient is final) on one of the side streams, round-robin, and
        overlaps whatever follows; Engine.backward joins the streams at the end, GradAllReducer.ready orders its bucket behind them."""
        if not self.wstreams:
            yield None
            return
        i = self._wnext
        self._wnext = (i + 1) % len(self.wstreams)
        st = self.wstreams[i]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        st.wait_event(ev)
        with torch.cuda.stream(st):
            yield i

    def ordered_params(self):
        """Parameters in backward-completion order (reverse op order) — the flat-buffer layout."""
        seen, out = set(), []
        for op in reversed(self.ops):
            for p in op.param_list():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        rest = [p for p in self.module.parameters() if id(p) not in seen]
        return out + rest

    def attach(self, device):
        """Re-home all parameters into one flat fp32 buffer on `device` (views keep nn.Parameter identity)."""
        params = self.ordered_params()
        ok = self.flat is not None and self.flat.device == device
        if ok:
            lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
            ok = all(lo <= p.data_ptr() < hi for p in params)
        if ok:
            return
        self.device = device
        if device.type == 'cuda':
            from . import _lib
            _lib.probe_device(device.index if device.index is not None else None)
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4        # 16-byte aligned slots
        flat = torch.zeros(n, dtype=torch.float32, device=device)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.flat = flat
        self.flat_grad = torch.zeros_like(flat)
        self._views = {id(p): (o, self.flat_grad[o:o + p.numel()].view(p.shape)) for p, o in zip(params, offs)}
        self._op_ranges = []
        for op in reversed(self.ops):
            ps = op.param_list()
            if ps:
                lo = min(self._views[id(p)][0] for p in ps)
                hi = max(self._views[id(p)][0] + (p.numel() + 3) // 4 * 4 for p in ps)
                self._op_ranges.append((op, lo, hi))
        self.dummy = torch.zeros(64, dtype=torch.float32, device=device)
        self._buffers.clear()
        self._fp32_copies.clear()
        self._replan()
        self.params_version += 1

    def grad_of(self, p):
        return self._views[id(p)][1]

    def mark_params_dirty(self):
        self.params_version += 1

    # ---- planning ---------------------------------------------------------------------------------
    def _plan(self, N, spatial, need_grad):
        key = (N, tuple(spatial), need_grad)
        if self._planned == key:
            return
        self.x.spatial = tuple(spatial)
        for op in self.ops:
            op.plan(self, N)
        if need_grad:
            for op in self.ops:
                v = op.out
                if isinstance(op, HeadOp):
                    v.grad = None            # provided by the loss (dlogits)
                else:
                    v.grad = self.buffer(v.name + '.grad', tuple(v.act.buf.shape[:4]) + (v.C,), self.grad_dtype(v.spatial))
            # out = lrelu(main + res): the masked gradient of `out` IS the first contribution to the gradient of `res` (ResAddOp is
            # the last consumer of `res` in forward, so the first writer of its gradient in backward) — the two share one buffer and
            # the residual path costs no copy; along a chain of blocks the gradient flows through ONE buffer, masked in place
            for op in self.ops:
                if isinstance(op, ResAddOp) and op.res.grad is not None and tuple(op.res.grad.shape) == tuple(op.out.grad.shape):
                    op.out.grad = op.res.grad
        else:
            for op in self.ops:
                op.out.grad = None
        self.x.grad = None
        self._planned = key
        self._packed_version = None
        self._pack_programs = {}
        self._io_cache = {}
        self._fp32_copies.clear()

    def _pack(self, need_grad):
        ver = (self.params_version, self.flat._version, need_grad)
        if self._packed_version == ver:
            return
        # the program only holds POINTERS (weights inside the flat buffer, packed destinations): it stays valid until attach(),
        # _plan() or set_precision() clear it — parameter VALUES changing (every optimizer step) just re-runs it
        key = need_grad
        prog = self._pack_programs.get(key)
        if key not in self._pack_programs:   # record the ops' packing calls once; afterwards every step is one batched launch
            def record(bwd):
                rec, owner = [], []
                ops._pack_recorder = rec
                try:
                    for i, op in enumerate(self.ops):
                        op.pack(self, bwd)
                        owner += [i] * (len(rec) - len(owner))
                finally:
                    ops._pack_recorder = None
                return rec, owner
            rec, owner = record(need_grad)
            late, mid, mid_op = [], [], None
            if need_grad and self.bwdw_streams > 0 and self.flat.is_cuda:
                # Off the chain, on the side stream: (a) the packings only backward reads (flipped / transposed weights of the
                # backward-data launches) — Engine.backward waits for them; (b) the forward packings of the DEEP layers (almost all
                # of the bytes: the 240..320-channel stages), which the forward pass only needs after the full-resolution layers
                # have run for milliseconds — Engine.forward waits for them in front of op `mid_op`.  The main stream packs what the
                # first layers need (2 MB) and starts.
                fwd_dst = {r[1].data_ptr() for r in record(False)[0]}
                is_fwd = [r[1].data_ptr() in fwd_dst for r in rec]
                late = [r for r, f in zip(rec, is_fwd) if not f]
                acc = 0
                for r, f, o in zip(rec, is_fwd, owner):
                    if f:
                        acc += r[1].numel() * 4
                        if acc > (2 << 20) and o > 0 and os.environ.get('MT_PACK_SPLIT', '1') != '0':
                            mid_op = o
                            break
                if mid_op is not None:
                    mid = [r for r, f, o in zip(rec, is_fwd, owner) if f and o >= mid_op]
                    rec = [r for r, f, o in zip(rec, is_fwd, owner) if f and o < mid_op]
                else:
                    rec = [r for r, f in zip(rec, is_fwd) if f]
            mk = lambda rr: ops.PackProgram(rr, self.device) if rr else None
            prog = (mk(rec), mk(late), mk(mid), mid_op)
            self._pack_programs[key] = prog
        self._late_pack = None
        self._mid_pack = None
        if prog[0] is not None:
            prog[0].run()
        if prog[1] is not None or prog[2] is not None:
            if not self.wstreams:
                self.wstreams = [self._side_stream() for _ in range(self.bwdw_streams)]
            st = self.wstreams[0]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())      # behind the optimizer step that wrote the weights
            st.wait_event(ev)
            with torch.cuda.stream(st):
                if prog[2] is not None:
                    prog[2].run()
                    self._mid_pack = (prog[3], torch.cuda.Event())
                    self._mid_pack[1].record(st)
                if prog[1] is not None:
                    prog[1].run()
                    self._late_pack = torch.cuda.Event()
                    self._late_pack.record(st)
        self._packed_version = ver

    # ---- execution --------------------------------------------------------------------------------
    def forward_to_final_head(self, x):
        """Inference: everything but the segmentation heads.  Returns the mt_pointwise_t of the FINAL head (source = its lazily
        activated input for all N samples, packed weights, bias) for ops.head_flip_accumulate, which fuses the head with the
        nonlinearity, the un-flip and the accumulation of the sliding window (the logits are never stored)."""
        self.forward(x, need_grad=False, all_heads=False, _skip_final=True)
        op = next(o for o in self.ops if isinstance(o, HeadOp) and o.out is self.heads[self.final_head])
        a = op.srcs[0].act                      # the fused inference heads take fp32 and 16-bit sources (even channel stride)
        if a.dtype != torch.float32 and (a.cs & 1):
            a = self.as_fp32(a)
        p = ops.fill_pointwise(a, op.geom.out, a.spatial, (1, 1, 1), (1, 1, 1), op.conv.out_channels, op.wf, op.conv.bias, op.out.act)
        p._keep = (op.wf, op.conv.bias, a.buf)  # the struct only holds raw pointers
        return p

    def forward(self, x, need_grad=True, all_heads=True, _skip_final=False):
        """x: [N,C,D,H,W] float32 HIP tensor.  Returns list of NDHWC logits buffers (module output order)."""
        if not x.is_cuda:
            raise RuntimeError("multitalent_amd: the network runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
        self.attach(x.device)
        self._iter += 1
        N, Cin = x.shape[0], x.shape[1]
        spatial = tuple(x.shape[2:])
        self._plan(N, spatial, need_grad)
        x = x.contiguous().float()
        if Cin == 1:
            xb = x.reshape((N,) + spatial + (1,))       # NCDHW with C == 1 is already NDHWC
        else:
            xb = ops.ncdhw_to_ndhwc(x, out=self.buffer('x.ndhwc', (N,) + spatial + (Cin,)))
        self.x.act = Act(xb)
        self._pack(need_grad)
        skip_heads = set()
        if not all_heads:
            skip_heads = {id(self.heads[i]) for i in range(len(self.heads)) if i != self.final_head or _skip_final}
        for i, op in enumerate(self.ops):
            if self._mid_pack is not None and i >= self._mid_pack[0]:      # the deep layers' packed weights (Engine._pack)
                torch.cuda.current_stream().wait_event(self._mid_pack[1])
                self._mid_pack = None
            if isinstance(op, HeadOp) and id(op.out) in skip_heads:
                continue
            op.forward(self)
        if all_heads:
            return [h.act.buf for h in self.heads]
        return [self.heads[self.final_head].act.buf]

    def backward(self, dlogits):
        """dlogits: list (module output order) of NDHWC gradient tensors or None.  Fills flat_grad."""
        self.flat_grad.zero_()
        nside = self.bwdw_streams if self.flat_grad.is_cuda else 0
        if len(self.wstreams) != nside:
            self.wstreams = [self._side_stream() for _ in range(nside)]
        self._wnext = 0
        if self._late_pack is not None:         # the backward-only weight packings of this step (Engine._pack)
            torch.cuda.current_stream().wait_event(self._late_pack)
            self._late_pack = None
        for op in self.ops:
            op.out.grad_init = False
        for h, g in zip(self.heads, dlogits):
            if g is not None:
                assert tuple(g.shape) == tuple(h.act.buf.shape) and g.is_contiguous()
                h.grad = g
                h.grad_init = True
            else:
                h.grad_init = False
        done_hi = 0
        # how many backward launches still write each gradient buffer (the last writer may fuse the next norm backward's first pass)
        will_run = [op for op in self.ops if not (isinstance(op, HeadOp) and not op.out.grad_init)]
        self.producer = {id(op.out): op for op in self.ops}
        self.pending = {}
        for op in will_run:
            for v in op.inputs():
                self.pending[id(v)] = self.pending.get(id(v), 0) + 1
        for op in reversed(self.ops):
            if isinstance(op, HeadOp) and not op.out.grad_init:
                continue
            op.backward(self)
            for v in op.inputs():
                self.pending[id(v)] -= 1
            if self.grad_ready_hook is not None and op.param_list():
                lo = min(self._views[id(p)][0] for p in op.param_list())
                hi = max(self._views[id(p)][0] + (p.numel() + 3) // 4 * 4 for p in op.param_list())
                done_hi = max(done_hi, hi)
                self.grad_ready_hook(lo, done_hi)
        for st in self.wstreams:
            torch.cuda.current_stream().wait_stream(st)
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(self.flat_grad.numel(), self.flat_grad.numel())

    # ---- autograd seam ----------------------------------------------------------------------------
    def apply(self, x, all_heads=True):
        need_grad = torch.is_grad_enabled() and self.module.training
        if not x.is_cuda:
            raise RuntimeError("multitalent_amd: the network runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
        self.attach(x.device)   # parameters must already live in the flat device buffer when autograd records them
        if not need_grad:
            outs = self.forward(x, need_grad=False, all_heads=all_heads)
            return [o.permute(0, 4, 1, 2, 3) for o in outs]
        params = self.ordered_params()
        return list(_UNetFunction.apply(self, all_heads, x, *params))


class _UNetFunction(torch.autograd.Function):
    """Whole-network autograd node: forward/backward run on the engine; gradients of the parameters are
    returned as views of the engine's flat gradient buffer."""

    @staticmethod
    def forward(ctx, eng, all_heads, x, *params):
        ctx.eng, ctx.all_heads = eng, all_heads
        outs = eng.forward(x, need_grad=True, all_heads=all_heads)
        ctx.nparams = len(params)
        return tuple(o.permute(0, 4, 1, 2, 3) for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        gl = []
        for g in gouts:
            gl.append(None if g is None else g.permute(0, 2, 3, 4, 1).contiguous())
        if ctx.all_heads:
            dl = gl
        else:
            dl = [None] * len(eng.heads)
            dl[eng.final_head] = gl[0]
        eng.backward(dl)
        grads = tuple(eng.grad_of(p) for p in eng.ordered_params())
        return (None, None, None) + grads


# ------------------------------------------------------------------------------------------------------
def build_plain_unet_engine(net):
    """Program for Generic_UNet.forward (reference generic_UNet.py:379-401)."""
    ops_list = []
    x = Val('x', net.conv_blocks_context[0].blocks[0].conv.in_channels)
    cur = x
    skips = []
    num_pool = len(net.tu)

    def block(name, srcs, blk):
        out = Val(name, blk.conv.out_channels)
        ops_list.append(ConvNormOp(name, srcs, out, blk.conv, blk.instnorm, lrelu=True))
        return out

    for d in range(num_pool):
        st = net.conv_blocks_context[d]
        for j, blk in enumerate(st.blocks):
            cur = block('ctx%d.%d' % (d, j), [cur], blk)
        skips.append(cur)
    bott = net.conv_blocks_context[num_pool]
    for i, st in enumerate(bott):
        for j, blk in enumerate(st.blocks):
            cur = block('ctx%d.%d.%d' % (num_pool, i, j), [cur], blk)
    head_vals = []
    for u in range(num_pool):
        up = Val('tu%d' % u, net.tu[u].out_channels)
        ops_list.append(TConvOp('tu%d' % u, cur, up, net.tu[u]))
        srcs = [up, skips[-(u + 1)]]
        for i, st in enumerate(net.conv_blocks_localization[u]):
            for j, blk in enumerate(st.blocks):
                cur = block('loc%d.%d.%d' % (u, i, j), srcs, blk)
                srcs = [cur]
        hv = Val('seg%d' % u, net.seg_outputs[u].out_channels)
        ops_list.append(HeadOp('seg%d' % u, cur, hv, net.seg_outputs[u]))
        head_vals.append(hv)
    ordered = [head_vals[-1]] + head_vals[:-1][::-1]      # generic_UNet.py:396-399
    return Engine(net, ops_list, x, ordered, 0)


def build_resenc_unet_engine(net):
    """Program for FabiansUNet.forward (generic_modular_residual_UNet.py:355-358)."""
    enc, dec = net.encoder, net.decoder
    ops_list = []
    x = Val('x', enc.initial_conv.in_channels)
    stem = Val('stem', enc.initial_conv.out_channels)
    ops_list.append(ConvNormOp('stem', [x], stem, enc.initial_conv, enc.initial_norm, lrelu=True))
    # the stem output is consumed by conv1 AND the residual add of the first block: materialise it
    cur = Val('stem.a', stem.C)
    ops_list.append(_MaterialiseOp('stem.a', stem, cur))
    skips = []
    for s, layer in enumerate(enc.stages):
        for b, blk in enumerate(layer.convs):
            name = 'enc%d.%d' % (s, b)
            h1 = Val(name + '.c1', blk.conv1.out_channels)
            ops_list.append(ConvNormOp(name + '.c1', [cur], h1, blk.conv1, blk.norm1, lrelu=True))
            h2 = Val(name + '.c2', blk.conv2.out_channels)
            ops_list.append(ConvNormOp(name + '.c2', [h1], h2, blk.conv2, blk.norm2, lrelu=False))
            if blk.downsample_skip is not None:
                res = Val(name + '.sk', blk.out_planes)
                ops_list.append(ConvNormOp(name + '.sk', [cur], res, blk.downsample_skip[0], blk.downsample_skip[1],
                                           lrelu=False))
            else:
                res = cur
            out = Val(name, blk.out_planes)
            ops_list.append(ResAddOp(name, h2, res, out))
            cur = out
        skips.append(cur)
    skips = skips[::-1]
    cur = skips[0]
    head_vals = []
    for i in range(len(dec.tus)):
        up = Val('dtu%d' % i, dec.tus[i].out_channels)
        ops_list.append(TConvOp('dtu%d' % i, cur, up, dec.tus[i]))
        blk = dec.stages[i].convs[0]
        cur = Val('dec%d' % i, blk.conv.out_channels)
        ops_list.append(ConvNormOp('dec%d' % i, [up, skips[i + 1]], cur, blk.conv, blk.norm, lrelu=True))
        hv = Val('dseg%d' % i, dec.deep_supervision_outputs[i].out_channels)
        ops_list.append(HeadOp('dseg%d' % i, cur, hv, dec.deep_supervision_outputs[i]))
        head_vals.append(hv)
    ordered = head_vals[::-1]                               # generic_modular_UNet.py:288
    return Engine(net, ops_list, x, ordered, 0)


class _MaterialiseOp(_Op):
    """dense copy a = lrelu(IN(y)) of a lazy activation that has more than one kind of consumer."""

    def __init__(self, name, src, out):
        self.name, self.src, self.out = name, src, out

    def inputs(self):
        return [self.src]

    def plan(self, eng, N):
        self.out.spatial = self.src.spatial
        self.out.act = Act(eng.buffer(self.name, (N,) + self.src.spatial + (self.src.C,), eng.val_dtype(self.src.spatial)))

    def pack(self, eng, need_bwd):
        pass

    def forward(self, eng):
        ops.inorm_lrelu_apply(self.src.act, self.out.act)

    def backward(self, eng):
        # the gradient w.r.t. the materialised activation IS the gradient w.r.t. the lazy one: alias the buffer
        assert self.out.grad_init
        self.src.grad = self.out.grad
        self.src.grad_init = True
