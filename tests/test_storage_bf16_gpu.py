"""16-bit STORAGE of activations (fp16) and gradients (bf16): the mixed-precision mode's HBM format (reference: autocast keeps conv
inputs / outputs in half precision, MultiTalent_Trainer_DDP.py:340-354, network_trainer.py:400-402).

A kernel that reads a 16-bit tensor widens exactly and a kernel that writes one rounds its fp32 result once (RNE).  Where the
matrix type does not change with the storage type (bf16 products: backward-data, backward-weight) every 16-bit-storage kernel must
therefore agree with its fp32-storage form (validated against the oracle elsewhere) BIT FOR BIT after rounding on inputs that are
representable: out_16 == round(out_fp32).  The fp16 forward kernels (fp16 products) are checked against the same arithmetic restated
on the host (operands rounded to fp16, exact products, fp32-level sums); the streaming kernels against their formulas in torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from multitalent_amd import ops
    return ops


def rbf(x, dt=torch.bfloat16):
    return x.to(dt).to(torch.float32)


def nd(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def test_cast_roundtrip_strided_accumulate(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 5, 6, 7, 12), generator=g).to(dev)
    # dense fp32 -> bf16 -> fp32
    b = torch.empty_like(x, dtype=torch.bfloat16)
    ops.cast(ops.Act(x), ops.Act(b))
    assert torch.equal(b, x.to(torch.bfloat16))
    y = torch.full_like(x, float('nan'))
    ops.cast(ops.Act(b), ops.Act(y))
    assert torch.equal(y, b.float())
    # channel slices (strided on both sides) + accumulate: dst[..., 2:8] += src[..., 4:10]
    dst = torch.randn((2, 5, 6, 7, 10), generator=g).to(dev).to(torch.bfloat16)
    ref = dst.float().clone()
    ref[..., 2:8] = (ref[..., 2:8] + x[..., 4:10])
    ops.cast(ops.Act(x, 4, 6), ops.Act(dst, 2, 6), accumulate=True)
    assert torch.equal(dst, ref.to(torch.bfloat16))
    # odd sizes take the strided kernel
    x2 = torch.randn((1, 3, 3, 3, 5), generator=g).to(dev)
    b2 = torch.empty_like(x2, dtype=torch.bfloat16)
    ops.cast(ops.Act(x2), ops.Act(b2))
    assert torch.equal(b2, x2.to(torch.bfloat16))
    # fp16 <-> fp32 <-> bf16
    h = torch.empty_like(x, dtype=torch.float16)
    ops.cast(ops.Act(x), ops.Act(h))
    assert torch.equal(h, x.to(torch.float16))
    hb = torch.empty_like(x, dtype=torch.bfloat16)
    ops.cast(ops.Act(h), ops.Act(hb))
    assert torch.equal(hb, h.float().to(torch.bfloat16))
    h2 = torch.empty_like(x2, dtype=torch.float16)
    ops.cast(ops.Act(b2), ops.Act(h2))
    assert torch.equal(h2, b2.float().to(torch.float16))


@pytest.mark.parametrize("adt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,shape", [(30, (6, 10, 12)), (32, (4, 8, 16)), (60, (5, 7, 9)), (120, (3, 4, 8)), (7, (3, 5, 5))])
def test_streaming_kernels_16bit(dev, C, shape, adt):
    """inorm_lrelu_apply (+ residual), inorm_lrelu_bwd, lrelu_bwd (+ copy), lrelu_bwd_stats, channel_sum: activations of type adt
    (fp16 = the engine's choice, bf16), gradients bf16."""
    from multitalent_amd import _lib
    import ctypes as Ct
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    N = 2
    V = int(np.prod(shape))
    bf = lambda t: t.to(dev).to(torch.bfloat16)
    af = lambda t: t.to(dev).to(adt)
    ulp = 2.0 ** -7 if adt == torch.bfloat16 else 2.0 ** -10
    ADT = _lib.MT_BF16 if adt == torch.bfloat16 else _lib.MT_F16
    y = af(torch.randn((N,) + shape + (C,), generator=g) * 2 + 0.5)
    r = af(torch.randn((N,) + shape + (C,), generator=g))
    sc = (torch.rand((N, C), generator=g) + 0.5).to(dev)
    sh = torch.randn((N, C), generator=g).to(dev)
    rsc = (torch.rand((N, C), generator=g) + 0.5).to(dev)
    rsh = torch.randn((N, C), generator=g).to(dev)
    bc = lambda t: t[:, None, None, None, :]
    lre = lambda t, s: torch.where(t > 0, t, t * s)
    # ---- apply with residual
    out = torch.empty_like(y)
    ops.inorm_lrelu_apply(ops.Act(y, scale=sc, shift=sh, slope=0.01), ops.Act(out), res=ops.Act(r, scale=rsc, shift=rsh, slope=1.0))
    t = torch.addcmul(bc(sh), y.float(), bc(sc))        # fma
    ref = lre(t + lre(torch.addcmul(bc(rsh), r.float(), bc(rsc)), 1.0), 0.01)
    d = (out.float() - rbf(ref, adt)).abs().max()
    assert float(d) <= float(ref.abs().max()) * ulp, float(d)         # at most one ulp (fma vs mul+add before the rounding)
    assert float((out.float() != rbf(ref, adt)).float().mean()) < 0.01
    # ---- norm backward
    mean = y.float().mean((1, 2, 3))
    var = y.float().var((1, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.3).to(dev)
    gg = bf(torch.randn((N,) + shape + (C,), generator=g))
    g0 = gg.clone()
    dgam, dbet, dbias = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    a = ops.Act(y, scale=(gamma * rstd).contiguous(), shift=(beta - mean * gamma * rstd).contiguous(), slope=0.01, mean=mean.contiguous(), rstd=rstd.contiguous())
    ws = torch.empty(ops.inorm_bwd_workspace(N, V, C) // 4 + 16, device=dev)
    ops.inorm_lrelu_bwd(ops.Act(gg), a, gamma, beta, dgam, dbet, dbias, ws)
    zh = (y.float() - bc(mean)) * bc(rstd)
    z = zh * gamma + beta
    dz = torch.where(z > 0, g0.float(), g0.float() * 0.01)
    A = dz.mean((1, 2, 3))
    B = (dz * zh).mean((1, 2, 3))
    dy = (gamma * bc(rstd)) * (dz - bc(A) - zh * bc(B))
    err = (gg.float() - dy).abs().max() / dy.abs().max()
    assert float(err) < 2 ** -7, float(err)
    assert torch.allclose(dgam, (dz * zh).sum((0, 1, 2, 3)), rtol=1e-3, atol=1e-3 * V ** 0.5)
    assert torch.allclose(dbet, dz.sum((0, 1, 2, 3)), rtol=1e-3, atol=1e-3 * V ** 0.5)
    assert torch.allclose(dbias, gg.float().sum((0, 1, 2, 3)), rtol=1e-3, atol=1e-3 * V ** 0.5)      # the sum of what was stored
    # ---- lrelu backward with copy (+ fused statistics where the shape takes them)
    lib = _lib.load()
    g1 = bf(torch.randn((N,) + shape + (C,), generator=g))
    g1ref = g1.clone()
    cp = torch.empty_like(g1)
    tt = torch.addcmul(bc(sh), y.float(), bc(sc)) + lre(torch.addcmul(bc(rsh), r.float(), bc(rsc)), 1.0)
    want = torch.where(tt > 0, g1ref.float(), g1ref.float() * 0.01)
    _lib.check(lib.mt_lrelu_bwd(Ct.c_void_p(g1.data_ptr()), C, Ct.c_void_p(y.data_ptr()), C, ops._ptr(sc), ops._ptr(sh), 0.01,
                                Ct.c_void_p(r.data_ptr()), C, ops._ptr(rsc), ops._ptr(rsh), 1.0, Ct.c_void_p(cp.data_ptr()), C, N, V, C,
                                _lib.MT_BF16, ADT, ops._stream()), 'lrelu_bwd')
    assert float((g1.float() != rbf(want)).float().mean()) < 1e-3 and torch.equal(cp, g1)     # (fma vs mul+add can flip the sign of a t ~ 0)
    nblk = lib.mt_lrelu_bwd_stats_blocks(V, C)
    if nblk > 0:
        g2 = g1ref.clone()
        part = torch.zeros((N, nblk, C, 2), device=dev)
        _lib.check(lib.mt_lrelu_bwd_stats(Ct.c_void_p(g2.data_ptr()), Ct.c_void_p(y.data_ptr()), ops._ptr(sc), ops._ptr(sh), 0.01,
                                          Ct.c_void_p(r.data_ptr()), ops._ptr(rsc), ops._ptr(rsh), 1.0, None, ops._ptr(mean), ops._ptr(rstd),
                                          ops._ptr(part), N, V, C, _lib.MT_BF16, ADT, ops._stream()), 'lrelu_bwd_stats')
        assert torch.equal(g2, g1)
        s = part.sum(1)
        assert torch.allclose(s[..., 0], g2.float().sum((1, 2, 3)), rtol=1e-3, atol=1e-3 * V ** 0.5)
        assert torch.allclose(s[..., 1], (g2.float() * zh).sum((1, 2, 3)), rtol=1e-3, atol=1e-3 * V ** 0.5)
    # ---- channel sum
    o = torch.zeros(C, device=dev)
    ws2 = torch.empty(ops.channel_sum_workspace(N, V, C) // 4 + 16, device=dev)
    ops.channel_sum(ops.Act(y), o, False, ws2)
    assert torch.allclose(o, y.float().sum((0, 1, 2, 3)), rtol=1e-4, atol=1e-3 * V ** 0.5)


def _conv_pair(dev, srcs, lazy, w, bias, geom, split=None, accumulate=False, stats=True, strided_bwd=False):
    """the same mma = 1 problem with fp32 and with bf16 storage; returns ((out0, out1, part) fp32, (...) bf16, kernel names)"""
    ops = _ops()
    res, names = [], []
    N = srcs[0].shape[0]
    Cout = w.shape[0]
    g = torch.Generator().manual_seed(11)
    C0 = Cout if split is None else split
    prev0 = rbf(torch.randn((N,) + tuple(geom.out) + (C0,), generator=g))
    prev1 = rbf(torch.randn((N,) + tuple(geom.out) + (Cout - C0,), generator=g)) if split is not None else None
    for dt in (torch.float32, torch.bfloat16):
        acts = []
        for i, s in enumerate(srcs):
            buf = s.to(dev).to(dt)
            if lazy[i] is not None:
                sc, sh, sl = lazy[i]
                acts.append(ops.Act(buf, scale=sc.to(dev), shift=sh.to(dev), slope=sl))
            else:
                acts.append(ops.Act(buf))
        o0 = (prev0.to(dev) if accumulate else torch.full(prev0.shape, float('nan'))).to(dev).to(dt)
        o1 = None
        if split is not None:
            o1 = (prev1.to(dev) if accumulate else torch.full(prev1.shape, float('nan'))).to(dev).to(dt)
        bd = bias.to(dev) if bias is not None else None
        p = ops.fill_conv(acts, geom, Cout, out0=ops.Act(o0), out1=ops.Act(o1) if o1 is not None else None, csplit=split, bias=bd,
                          accumulate=accumulate, mma=1)
        assert ops.conv_io_supported(p), "storage types not taken natively"
        wd = w.to(dev).contiguous()
        wp = ops.pack_conv_weights(wd, acts[0].C, acts[1].C if len(acts) > 1 else 0, Cout, w.shape[2:], ops.conv_weight_strides(wd), False,
                                   ops.conv_ck(p), layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        part = None
        if stats:
            part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
            p.stats_part = part.data_ptr()
        names.append(ops.conv_kernel_name(p))
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        res.append((o0, o1, part))
    return res[0], res[1], names


@pytest.mark.parametrize("Cin,Cout,shape,k,two,split,acc", [
    (32, 32, (8, 16, 64), (3, 3, 3), False, None, False),
    (30, 60, (9, 14, 70), (3, 3, 3), False, None, False),          # ragged tiles, 14-channel tail chunk
    (30, 30, (6, 20, 40), (1, 3, 3), False, None, False),          # 1x3x3 (residual encoder stage 0)
    (60, 30, (8, 16, 64), (3, 3, 3), True, None, False),           # two sources (decoder: up | skip)
    (32, 60, (8, 16, 64), (3, 3, 3), False, 30, True),             # backward-data shape: two destinations, accumulate
    (64, 64, (8, 16, 32), (3, 3, 3), False, None, True),
])
def test_conv_bf16_kernel_bf16_storage_bitexact(dev, Cin, Cout, shape, k, two, split, acc):
    ops = _ops()
    ops.set_option('conv_bf16', 2)
    try:
        g = torch.Generator().manual_seed(5)
        N = 2
        pad = tuple((kk - 1) // 2 for kk in k)
        geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
        if two:
            srcs = [rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g)), rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g))]
            lazy = [None, (torch.rand((N, Cin // 2), generator=g) + 0.5, torch.randn((N, Cin // 2), generator=g), 0.01)]
        else:
            srcs = [rbf(torch.randn((N,) + shape + (Cin,), generator=g))]
            lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
        w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
        b = torch.randn(Cout, generator=g)
        f32, b16, names = _conv_pair(dev, srcs, lazy, w, b, geom, split=split, accumulate=acc, stats=not acc)
        assert names[0].startswith('conv_bf16_kernel') and names[1].startswith('conv_bf16_kernel'), names
        assert b16[0].dtype == torch.bfloat16
        assert torch.equal(b16[0], f32[0].to(torch.bfloat16)), float((b16[0].float() - f32[0]).abs().max())
        if split is not None:
            assert torch.equal(b16[1], f32[1].to(torch.bfloat16))
        if not acc:
            # statistics of the values AS STORED
            o = b16[0].float().double()
            s = b16[2].double().sum(1)
            assert torch.allclose(s[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
            assert torch.allclose(s[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    finally:
        ops.set_option('conv_bf16', 1)


def _host_conv_16(srcs, lazy, w, b, stride, pad, dt):
    """what a 16-bit matrix kernel computes, on the host: the lazily activated input rounded to dt, the weights rounded to dt, exact
    products, double sums"""
    import torch.nn.functional as F
    xs = []
    for s, lz in zip(srcs, lazy):
        t = s.permute(0, 4, 1, 2, 3)
        if lz is not None:
            sc, sh, sl = lz
            t = torch.addcmul(sh[:, :, None, None, None], t, sc[:, :, None, None, None])
            t = torch.maximum(t, t * sl)
        xs.append(rbf(t, dt))
    x = torch.cat(xs, 1)
    y = F.conv3d(x.double(), rbf(w, dt).double(), b.double() if b is not None else None, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 4, 1).float()


@pytest.mark.parametrize("Cin,Cout,shape,k,stride,two,odt", [
    (32, 32, (8, 16, 64), (3, 3, 3), (1, 1, 1), False, torch.float16),       # conv_bf16_kernel, fp16 products
    (30, 60, (9, 14, 70), (3, 3, 3), (1, 1, 1), False, torch.float16),
    (30, 30, (6, 20, 40), (1, 3, 3), (1, 1, 1), False, torch.float16),
    (60, 30, (8, 16, 64), (3, 3, 3), (1, 1, 1), True, torch.float16),
    (30, 60, (8, 18, 34), (3, 3, 3), (2, 2, 2), False, torch.float16),       # strided stage conv
    (32, 64, (6, 16, 32), (3, 3, 3), (1, 2, 2), False, torch.float32),       # ... into a level kept in fp32
    (64, 64, (3, 6, 6), (3, 3, 3), (1, 1, 1), False, torch.float16),         # tap-split kernel (low-resolution stages)
    (320, 320, (6, 12, 12), (3, 3, 3), (1, 1, 1), False, torch.float16),
])
def test_forward_convs_fp16_storage(dev, Cin, Cout, shape, k, stride, two, odt):
    """forward convolutions over fp16 activations: fp16 products (v_mfma_f32_32x32x16_f16), fp32 sums, fp16 (or fp32) output, statistics
    of the stored values — against the host restatement."""
    ops = _ops()
    tapsplit = k == (3, 3, 3) and stride == (1, 1, 1) and shape[2] <= 12
    ops.set_option('conv_bf16', 1 if tapsplit else 2)          # 2: the 16-bit matrix kernel also on the small grids of this test
    try:
        _forward_conv_fp16(dev, Cin, Cout, shape, k, stride, two, odt, tapsplit)
    finally:
        ops.set_option('conv_bf16', 1)


def _forward_conv_fp16(dev, Cin, Cout, shape, k, stride, two, odt, tapsplit):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, stride, pad)
    H = torch.float16
    if two:
        srcs = [rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g), H), rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g), H)]
        lazy = [None, (torch.rand((N, Cin // 2), generator=g) + 0.5, torch.randn((N, Cin // 2), generator=g), 0.01)]
    else:
        srcs = [rbf(torch.randn((N,) + shape + (Cin,), generator=g), H)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    b = torch.randn(Cout, generator=g)
    acts, keep = [], []
    for sx, lz in zip(srcs, lazy):
        buf = sx.to(dev).to(H)
        keep.append(buf)
        acts.append(ops.Act(buf) if lz is None else ops.Act(buf, scale=lz[0].to(dev), shift=lz[1].to(dev), slope=lz[2]))
    out = torch.full((N,) + tuple(geom.out) + (Cout,), float('nan'), device=dev).to(odt)
    bd = b.to(dev)
    p = ops.fill_conv(acts, geom, Cout, out0=ops.Act(out), bias=bd, mma=1)
    name = ops.conv_kernel_name(p)
    assert ops.conv_io_supported(p), name
    assert ops.conv_pack_layout(p) == 4, (name, ops.conv_pack_layout(p))                 # fp16 weight fragments
    wd = w.to(dev).contiguous()
    wp = ops.pack_conv_weights(wd, acts[0].C, acts[1].C if two else 0, Cout, k, ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=4)
    p.wpack = wp.data_ptr()
    part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
    p.stats_part = part.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    ref = _host_conv_16(srcs, lazy, w, b, stride, pad, H)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = (2.0 ** -10 if odt == H else 1e-4) * float(ref.abs().max())
    # (the activation t = x * scale + shift is an fma on the device: an fp16 rounding boundary crossed by it shows as one input ulp)
    assert float((got - ref).abs().max()) < 4 * tol + 2e-3, (name, float((got - ref).abs().max()), tol)
    assert float(((got - ref).abs() > tol).float().mean()) < 2e-3, name
    o = out.float().double()
    sm = part.double().sum(1)
    assert torch.allclose(sm[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
    assert torch.allclose(sm[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    if tapsplit:
        assert name.startswith('conv_tapsplit_kernel'), name


@pytest.mark.parametrize("Cin,Cout,shape,stride", [(30, 60, (8, 18, 34), (2, 2, 2)), (32, 64, (6, 16, 32), (1, 2, 2))])
@pytest.mark.parametrize("out_bf16", [True, False])
def test_strided_stage_conv_bf16_storage_bitexact(dev, Cin, Cout, shape, stride, out_bf16):
    """forward strided 3x3x3 (bf16 source; bf16 or fp32 destination) and its one-launch backward-data (dY bf16 or fp32, dX bf16)."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    N = 2
    geom = ops.ConvGeom(shape, (3, 3, 3), stride, (1, 1, 1))
    x = rbf(torch.randn((N,) + shape + (Cin,), generator=g))
    sc, sh = torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g)
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / np.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    wd = w.to(dev).contiguous()
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        a = ops.Act(x.to(dev).to(dt), scale=sc.to(dev), shift=sh.to(dev), slope=0.01)
        odt = torch.bfloat16 if (out_bf16 and dt == torch.bfloat16) else torch.float32
        out = torch.full((N,) + tuple(geom.out) + (Cout,), float('nan'), device=dev).to(odt)
        bd = b.to(dev)
        p = ops.fill_conv([a], geom, Cout, out0=ops.Act(out), bias=bd, mma=1)
        assert ops.conv_io_supported(p)
        assert ops.conv_kernel_name(p).startswith('conv_fast_strided_kernel') and ops.conv_kernel_name(p).endswith('true>')
        wp = ops.pack_conv_weights(wd, Cin, 0, Cout, (3, 3, 3), ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
        p.stats_part = part.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        outs[dt] = (out, part)
    o32, ob = outs[torch.float32][0], outs[torch.bfloat16][0]
    if out_bf16:
        assert torch.equal(ob, o32.to(torch.bfloat16))
        o = ob.float().double()
        s = outs[torch.bfloat16][1].double().sum(1)
        assert torch.allclose(s[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
        assert torch.allclose(s[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    else:
        assert torch.equal(ob, o32)
    # ---- backward-data: dX (bf16, accumulating) from dY (bf16 when the level below stores bf16, else fp32)
    dy = rbf(torch.randn((N,) + tuple(geom.out) + (Cout,), generator=g))
    prev = rbf(torch.randn((N,) + shape + (Cin,), generator=g))
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        ydt = torch.bfloat16 if (out_bf16 and dt == torch.bfloat16) else torch.float32
        dx = prev.to(dev).to(dt)
        dyd = dy.to(dev).to(ydt)                     # (kept alive: the struct only holds raw pointers)
        p = ops.fill_conv([ops.Act(dyd)], geom, Cout, out0=ops.Act(dx), accumulate=True, mma=1)
        p.Cin = Cin
        assert ops.conv3d_bwd_data_strided_supported(p) and ops.conv_bwd_data_strided_io_supported(p)
        wb = ops.pack_conv_weights(wd, Cout, 0, Cin, (3, 3, 3), ops.conv_weight_strides(wd, as_bwd_data=True), False, 16,
                                   layout=ops.conv_bwd_data_strided_pack_layout(p))
        p.wpack = wb.data_ptr()
        ops.conv3d_bwd_data_strided(p)
        torch.cuda.synchronize()
        res[dt] = dx
    # independent reference: autograd's transposed convolution on the host (weights rounded like the packed ones)
    import torch.nn.functional as F
    xr = torch.zeros((N, Cin) + shape, requires_grad=True)
    F.conv3d(xr, rbf(w), None, stride=stride, padding=1).backward(dy.permute(0, 4, 1, 2, 3).contiguous())
    ref = prev + xr.grad.permute(0, 2, 3, 4, 1)
    for dt in (torch.float32, torch.bfloat16):
        got = res[dt].float().cpu()
        bad = (got - ref).abs() > (2e-3 if dt == torch.float32 else 2.0 ** -7) * ref.abs().max()
        if bool(bad.any()):
            idx = bad.nonzero()
            raise AssertionError("%s storage: %d wrong elements; d %s h %s w %s c %s" % (
                dt, int(bad.sum()), idx[:, 1].unique().tolist()[:12], idx[:, 2].unique().tolist()[:20], idx[:, 3].unique().tolist()[:40], idx[:, 4].unique().tolist()[:40]))
    assert torch.equal(res[torch.bfloat16], res[torch.float32].to(torch.bfloat16)), \
        float((res[torch.bfloat16].float() - res[torch.float32]).abs().max())


@pytest.mark.parametrize("Cin,Cout,shape,k", [(32, 32, (6, 12, 64), (3, 3, 3)), (30, 60, (5, 10, 40), (3, 3, 3)), (30, 30, (4, 12, 64), (1, 3, 3))])
@pytest.mark.parametrize("xb,yb", [(torch.float16, True), (torch.bfloat16, True), (torch.float16, False), (None, True)])
def test_bwdw_wino_bf16_storage_bitexact(dev, Cin, Cout, shape, k, xb, yb):
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
    x = rbf(torch.randn((N,) + shape + (Cin,), generator=g), xb if xb is not None else torch.bfloat16)
    sc, sh = torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g)
    dy = rbf(torch.randn((N,) + shape + (Cout,), generator=g))
    res = []
    for xdt, ydt in ((torch.float32, torch.float32), (xb if xb is not None else torch.float32, torch.bfloat16 if yb else torch.float32)):
        a = ops.Act(x.to(dev).to(xdt), scale=sc.to(dev), shift=sh.to(dev), slope=0.01)
        y = ops.Act(dy.to(dev).to(ydt))
        p = ops.fill_conv([a], geom, Cout, mma=1)
        assert ops.conv_bwd_weight_io_supported(p, y)
        assert ops.conv_bwd_weight_kernel_name(p, y).startswith('conv_bwdw_wino_bf16_kernel'), ops.conv_bwd_weight_kernel_name(p, y)
        dw = torch.full((Cout, Cin) + k, float('nan'), device=dev)
        ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
        ops.conv3d_bwd_weight(p, y, dw, ops.conv_weight_strides(dw), False, ws)
        torch.cuda.synchronize()
        res.append(dw)
    assert torch.isfinite(res[1]).all()
    assert torch.equal(res[0], res[1]), float((res[0] - res[1]).abs().max())


def test_unsupported_storage_types_are_refused(dev):
    """a kernel that does not take bf16 operands must fail loudly, never read 2-byte data as fp32"""
    ops = _ops()
    geom = ops.ConvGeom((4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1))
    x = torch.zeros((1, 4, 8, 8, 16), device=dev, dtype=torch.bfloat16)
    out = torch.zeros((1, 4, 8, 8, 32), device=dev, dtype=torch.bfloat16)
    w = torch.zeros((32, 16, 3, 3, 3), device=dev)
    p = ops.fill_conv([ops.Act(x)], geom, 32, out0=ops.Act(out), mma=0)          # fp32 matrix path: fp32 storage only
    assert not ops.conv_io_supported(p)
    wp = ops.pack_conv_weights(w, 16, 0, 32, (3, 3, 3), ops.conv_weight_strides(w), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    with pytest.raises(RuntimeError, match="storage types"):
        ops.conv3d_fwd(p)
    y = ops.Act(torch.zeros((1, 4, 8, 8, 32), device=dev, dtype=torch.bfloat16))
    assert not ops.conv_bwd_weight_io_supported(p, y)
    dw = torch.zeros_like(w)
    ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
    with pytest.raises(RuntimeError, match="storage types"):
        ops.conv3d_bwd_weight(p, y, dw, ops.conv_weight_strides(dw), False, ws)
