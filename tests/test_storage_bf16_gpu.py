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
    ops.set_option('conv_x16', 0)              # conv_bf16_kernel itself (the one-destination 16-bit problems take conv_x16_kernel by default)
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
        ops.set_option('conv_x16', 1)


@pytest.mark.parametrize("H", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,shape,k,two,acc,nwg", [
    (32, 32, (8, 16, 64), (3, 3, 3), False, False, 2),           # two workgroups walk 16 items each: the prefetch across tile boundaries
    (30, 60, (11, 14, 90), (3, 3, 3), False, False, 3),          # ragged tiles in d, h and w; 14-channel tail chunk; 28-channel tail cout tile
    (30, 30, (8, 20, 60), (1, 3, 3), False, False, 1000),        # 1x3x3 (residual encoder stage 0); 60-byte voxel rows (12-byte tail pieces)
    (60, 30, (8, 16, 64), (3, 3, 3), True, False, 5),            # two sources (decoder: up | skip), one of them plain
    (64, 64, (8, 16, 32), (3, 3, 3), False, True, 2),            # accumulate (gradient of a skip connection)
    (30, 30, (7, 9, 40), (3, 3, 3), False, True, 1000),          # accumulate on ragged tiles
    (48, 36, (4, 8, 32), (3, 3, 3), False, False, 1000),         # three chunks (a pair and a single); 4-channel tail cout tile (8-byte tail pieces)
    (32, 34, (4, 8, 32), (1, 3, 3), False, False, 2),            # 2-channel tail cout tile (4-byte tail pieces)
])
def test_conv_x16_kernel(dev, Cin, Cout, shape, k, two, acc, nwg, H):
    """conv_x16_kernel (persistent, weights in LDS, register prefetch, 16-byte stores) against (a) the host restatement of the 16-bit
    arithmetic and (b) conv_bf16_kernel on the same buffers: same operands, same products, fp32 sums in a different tap order — the
    stored 16-bit values agree except for rare one-ulp differences; statistics are those of the values as stored.  `nwg` caps the
    number of persistent workgroups so that a workgroup walks several (tile, cout tile) items across sample boundaries."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
    lay = 4 if H == torch.float16 else 3
    if two:
        srcs = [rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g), H), rbf(torch.randn((N,) + shape + (Cin // 2,), generator=g), H)]
        lazy = [None, (torch.rand((N, Cin // 2), generator=g) + 0.5, torch.randn((N, Cin // 2), generator=g), 0.01)]
    else:
        srcs = [rbf(torch.randn((N,) + shape + (Cin,), generator=g), H)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)] if not acc else [None]
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    b = torch.randn(Cout, generator=g) if not acc else None
    prev = rbf(torch.randn((N,) + tuple(geom.out) + (Cout,), generator=g), H)
    acts, keep = [], []
    for sx, lz in zip(srcs, lazy):
        buf = sx.to(dev).to(H)
        keep.append(buf)
        acts.append(ops.Act(buf) if lz is None else ops.Act(buf, scale=lz[0].to(dev), shift=lz[1].to(dev), slope=lz[2]))
    bd = b.to(dev) if b is not None else None
    wd = w.to(dev).contiguous()
    outs = {}
    ops.set_option('conv_bf16', 2)
    try:
        for mode in (4096 if nwg >= 1000 else nwg, 0):          # conv_x16: n > 1 = wherever eligible with at most n workgroups (4096: no cap), 0 = conv_bf16_kernel
            ops.set_option('conv_x16', mode)
            out = (prev.to(dev) if acc else torch.full(prev.shape, float('nan'))).to(dev).to(H)
            p = ops.fill_conv(acts, geom, Cout, out0=ops.Act(out), bias=bd, accumulate=acc, mma=1)
            name = ops.conv_kernel_name(p)
            assert name.startswith('conv_x16_kernel<%d' % k[0] if mode else 'conv_bf16_kernel'), name
            assert ops.conv_io_supported(p) and ops.conv_pack_layout(p) == lay
            wp = ops.pack_conv_weights(wd, acts[0].C, acts[1].C if two else 0, Cout, k, ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=lay)
            p.wpack = wp.data_ptr()
            part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
            p.stats_part = part.data_ptr()
            ops.conv3d_fwd(p)
            torch.cuda.synchronize()
            outs[bool(mode)] = (out.float().cpu(), part.double().sum(1).cpu())
    finally:
        ops.set_option('conv_bf16', 1)
        ops.set_option('conv_x16', 1)
    got, st = outs[True]
    old, st_old = outs[False]
    assert torch.isfinite(got).all()
    ulp = 2.0 ** -10 if H == torch.float16 else 2.0 ** -7
    ref = _host_conv_16(srcs, lazy, w, b, (1, 1, 1), pad, H)
    if acc:
        ref = ref + prev
    tol = ulp * float(ref.abs().max())
    assert float((got - ref).abs().max()) < 4 * tol + 2e-3, float((got - ref).abs().max())
    assert float(((got - ref).abs() > tol).float().mean()) < 2e-3
    # against conv_bf16_kernel: at most one ulp apart, and almost everywhere identical
    d = (got - old).abs()
    assert float(d.max()) <= 2 * ulp * float(old.abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.02, float((d > 0).float().mean())
    o = got.double()
    assert torch.allclose(st[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
    assert torch.allclose(st[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)


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
        if stride == (1, 1, 1):
            _forward_conv_fp16(dev, Cin, Cout, shape, k, stride, two, odt, tapsplit)
        else:
            # strided stage convs: the tap-split form these small grids take by default, and conv_fast_strided_kernel
            for ts, prefix in ((1, 'conv_tapsplit_kernel<4, true, 2, %d, 2, %d, 2, 2>' % (2 if odt == torch.float16 else 0, stride[0])), (0, 'conv_fast_strided_kernel')):
                ops.set_option('conv_tapsplit', ts)
                _forward_conv_fp16(dev, Cin, Cout, shape, k, stride, two, odt, False, prefix)
    finally:
        ops.set_option('conv_bf16', 1)
        ops.set_option('conv_tapsplit', 1)


def _forward_conv_fp16(dev, Cin, Cout, shape, k, stride, two, odt, tapsplit, prefix=None, H=torch.float16):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, stride, pad)
    lay = 4 if H == torch.float16 else 3                        # fp16 / bf16 weight fragments
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
    assert ops.conv_pack_layout(p) == lay, (name, ops.conv_pack_layout(p))
    assert prefix is None or name.startswith(prefix), (name, prefix)
    wd = w.to(dev).contiguous()
    wp = ops.pack_conv_weights(wd, acts[0].C, acts[1].C if two else 0, Cout, k, ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=lay)
    p.wpack = wp.data_ptr()
    part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
    p.stats_part = part.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    ref = _host_conv_16(srcs, lazy, w, b, stride, pad, H)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = ((2.0 ** -10 if H == torch.float16 else 2.0 ** -7) if odt == H else 1e-4) * float(ref.abs().max())
    # (the activation t = x * scale + shift is an fma on the device: an fp16 rounding boundary crossed by it shows as one input ulp)
    assert float((got - ref).abs().max()) < 4 * tol + 2e-3, (name, float((got - ref).abs().max()), tol)
    assert float(((got - ref).abs() > tol).float().mean()) < 2e-3, name
    o = out.float().double()
    sm = part.double().sum(1)
    assert torch.allclose(sm[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
    assert torch.allclose(sm[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    if tapsplit:
        assert name.startswith('conv_tapsplit_kernel'), name


@pytest.mark.parametrize("Cin,Cout,shape,stride,odt", [
    (30, 60, (8, 18, 34), (2, 2, 2), torch.bfloat16),
    (32, 70, (6, 16, 32), (1, 2, 2), torch.float32),
])
def test_strided_tapsplit_bf16_storage(dev, Cin, Cout, shape, stride, odt):
    """the strided tap-split kernel over bf16 sources (bf16 products; bf16 or fp32 destination) against the host restatement — the
    instances the fp16 cases of test_forward_convs_fp16_storage do not reach."""
    ops = _ops()
    ops.set_option('conv_bf16', 2)
    try:
        prefix = 'conv_tapsplit_kernel<4, true, 1, %d, 1, %d, 2, 2>' % (1 if odt == torch.bfloat16 else 0, stride[0])
        _forward_conv_fp16(dev, Cin, Cout, shape, (3, 3, 3), stride, False, odt, False, prefix, H=torch.bfloat16)
    finally:
        ops.set_option('conv_bf16', 1)


@pytest.mark.parametrize("Cin,Cout,shape,stride", [(30, 60, (8, 18, 34), (2, 2, 2)), (32, 64, (6, 16, 32), (1, 2, 2))])
@pytest.mark.parametrize("out_bf16", [True, False])
def test_strided_stage_conv_bf16_storage_bitexact(dev, Cin, Cout, shape, stride, out_bf16):
    """forward strided 3x3x3 (bf16 source; bf16 or fp32 destination) and its one-launch backward-data (dY bf16 or fp32, dX bf16)."""
    ops = _ops()
    ops.set_option('conv_tapsplit', 0)           # this test is about conv_fast_strided_kernel (grids this small take the tap-split form by default)
    try:
        _strided_stage_conv_bf16_storage_bitexact(dev, Cin, Cout, shape, stride, out_bf16)
    finally:
        ops.set_option('conv_tapsplit', 1)


def _strided_stage_conv_bf16_storage_bitexact(dev, Cin, Cout, shape, stride, out_bf16):
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
        assert ops.conv_kernel_name(p).startswith('conv_fast_strided_kernel') and ', true' in ops.conv_kernel_name(p)
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


def _bwdw_host(xs, dy, k, pad):
    """dW of a stride-1 convolution in float64 from already activated / rounded operands: xs [N, D, H, W, Cin], dy [N, D, H, W, Cout]."""
    import torch.nn.functional as F
    x = torch.cat(xs, -1).permute(0, 4, 1, 2, 3).double().requires_grad_(False)
    g = dy.permute(0, 4, 1, 2, 3).double()
    w = torch.zeros((g.shape[1], x.shape[1]) + tuple(k), dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, None, 1, pad).backward(g)
    return w.grad


@pytest.mark.parametrize("cins,Cout,shape,k,lazy,cap", [
    ((32,), 32, (6, 12, 64), (3, 3, 3), True, 1),
    ((30,), 60, (5, 10, 40), (3, 3, 3), True, 1),          # ragged column (h 10 = 2.5 tiles, w 40 = 1.25 tiles), channel tails
    ((30,), 30, (4, 12, 64), (1, 3, 3), True, 1),          # residual encoder stage 0
    ((30, 30), 30, (7, 9, 35), (3, 3, 3), True, 5),        # two sources (decoder concat), five workgroups: ranges cut columns anywhere
    ((48,), 32, (3, 8, 33), (3, 3, 3), False, 3),          # three chunks (the last pair is half empty), plain (not lazy) source
    ((60,), 30, (9, 16, 48), (1, 3, 3), True, 7),
    ((32,), 32, (12, 4, 32), (3, 3, 3), True, 2),          # one column per sample, two workgroups: every segment boundary inside a column
])
@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16])
def test_bwdw_tr16_vs_host(dev, cins, Cout, shape, k, lazy, cap, xdt):
    """conv_bwdw_tr16_kernel (round 5: the direct bf16 backward-weight fed by ds_read_b64_tr_b16) against float64 autograd on the operands the
    kernel multiplies: X activated in fp32 (one fma, LeakyReLU) and rounded ONCE to bf16, dY bf16, fp32 accumulation -> 2e-4 of max|dW|
    (an fp32-ulp difference of the host's fma may flip the bf16 rounding of single elements).  `cap` > 1 limits the number of workgroups
    (mt_set_option bwdw_tr16) so that one workgroup's (column, plane) range spans several columns and starts / ends inside columns."""
    ops = _ops()
    g = torch.Generator().manual_seed(31 + sum(cins) + Cout)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
    C = sum(cins)
    buf = rbf(1.5 * torch.randn((N,) + shape + (C + 2,), generator=g), xdt)            # the sources are channel slices of a wider buffer
    dy = rbf(torch.randn((N,) + shape + (Cout,), generator=g))
    xd = buf.to(dev).to(xdt)
    srcs, hx, c0 = [], [], 0
    for ci in cins:
        sl = buf[..., c0:c0 + ci]
        if lazy:
            sc, sh = torch.rand((N, ci), generator=g) + 0.5, torch.randn((N, ci), generator=g)
            srcs.append(ops.Act(xd, c0, ci, scale=sc.to(dev), shift=sh.to(dev), slope=0.01))
            t = torch.addcmul(sh[:, None, None, None, :], sl, sc[:, None, None, None, :])
            hx.append(rbf(torch.maximum(t, t * 0.01)))
        else:
            srcs.append(ops.Act(xd, c0, ci))
            hx.append(rbf(sl))
        c0 += ci
    y = ops.Act(dy.to(dev).to(torch.bfloat16))
    p = ops.fill_conv(srcs, geom, Cout, mma=1)
    ops.set_option('bwdw_tr16', cap)
    ops.apply_selection(p)
    try:
        assert ops.conv_bwd_weight_io_supported(p, y)
        name = ops.conv_bwd_weight_kernel_name(p, y)
        assert name.startswith('conv_bwdw_tr16_kernel<%d' % k[0]), name
        dw = torch.full((Cout, C) + k, float('nan'), device=dev)
        ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
        ops.conv3d_bwd_weight(p, y, dw, ops.conv_weight_strides(dw), False, ws)
        dw2 = dw.clone()
        ops.conv3d_bwd_weight(p, y, dw2, ops.conv_weight_strides(dw2), True, ws)          # accumulate: 2 dW
        torch.cuda.synchronize()
    finally:
        ops.set_option('bwdw_tr16', 1)
    ref = _bwdw_host(hx, dy, k, pad)
    err = float((dw.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert torch.isfinite(dw).all() and err < 2e-4, err
    assert torch.equal(dw2, 2 * dw)



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


# ---- pointwise kernels (transposed convolutions, heads, their backward-data), gather, tiled backward-weight, stem: fp32 arithmetic with
# ---- 16-bit storage -> bit-exact against the fp32-storage launch of the same kernel family, rounded
def _pointwise(dev, x, lazy, w_pack_args, base, in_spatial, si, so, Cout, bias, xdt, odt, ocs=None, prev=None, scatter=False, stats=False, mma=0, want_layout=None):
    ops = _ops()
    xb = x.to(dev).to(xdt)
    a = ops.Act(xb) if lazy is None else ops.Act(xb, scale=lazy[0].to(dev), shift=lazy[1].to(dev), slope=lazy[2])
    N = x.shape[0]
    osp = tuple(b * s for b, s in zip(base, so))
    ocs = Cout if ocs is None else ocs
    out = (prev.to(dev) if prev is not None else torch.full((N,) + osp + (ocs,), float('nan'))).to(dev).to(odt)
    w, Cin, taps, strides = w_pack_args
    wd = w.to(dev).contiguous()
    bd = bias.to(dev) if bias is not None else None
    p = ops.fill_pointwise(a, base, in_spatial, si, so, Cout, wd, bd, ops.Act(out, 0, Cout), accumulate=prev is not None, mma=mma)
    p.scatter = 1 if scatter else 0
    lay = ops.pointwise_pack_layout(p)
    if want_layout is not None:
        assert lay == want_layout, (lay, want_layout)
    wp = ops.pack_conv_weights(wd, Cin, 0, Cout, taps, strides(wd), False, ops.POINTWISE_CK, layout=lay)
    p.wpack = wp.data_ptr()
    part = None
    if stats:
        part = torch.zeros((N, ops.pointwise_stats_blocks(p), Cout, 2), device=dev)
        p.stats_part = part.data_ptr()
    assert ops.pointwise_io_supported(p), (xdt, odt)
    ops.pointwise_fwd(p)
    torch.cuda.synchronize()
    return out, part


@pytest.mark.parametrize("Cin,Cout,base,k,ocs_mult", [
    (60, 30, (4, 8, 32), (2, 2, 2), 1),       # dense output, whole rows: wide epilogue, linear stores
    (60, 30, (4, 8, 32), (2, 2, 2), 2),       # into a concat slot (channel stride 2 * Cout); 30 % 4 != 0 -> plain epilogue
    (64, 32, (4, 8, 32), (2, 2, 2), 2),       # concat slot, wide epilogue with 8-byte pieces
    (120, 60, (3, 5, 9), (2, 2, 2), 1),       # ragged: plain epilogue, two output-channel tiles
    (60, 30, (5, 6, 16), (1, 2, 2), 1),
])
def test_transposed_conv_fp16_storage_bitexact(dev, Cin, Cout, base, k, ocs_mult):
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    N = 2
    x = rbf(torch.randn((N,) + base + (Cin,), generator=g), torch.float16)
    lazy = (torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)
    w = torch.randn((Cin, Cout) + k, generator=g) / np.sqrt(Cin)
    wargs = (w, Cin, k, lambda wd: ops.conv_weight_strides(wd, transposed_layout=True))
    o32, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), k, Cout, None, torch.float32, torch.float32, ocs=Cout * ocs_mult)
    o16, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), k, Cout, None, torch.float16, torch.float16, ocs=Cout * ocs_mult)
    assert torch.equal(o16[..., :Cout], o32[..., :Cout].to(torch.float16))


@pytest.mark.parametrize("Cin,Cout", [(30, 47), (30, 2), (32, 3), (60, 47), (120, 20)])
def test_heads_read_fp16_activations(dev, Cin, Cout):
    """pw_head_kernel (33..64 logits), pw_narrow_kernel (<= 4), pw_fast_kernel: fp16 input, fp32 logits — the same fp32 arithmetic"""
    ops = _ops()
    g = torch.Generator().manual_seed(32)
    N, base = 2, (4, 8, 16)
    x = rbf(torch.randn((N,) + base + (Cin,), generator=g), torch.float16)
    lazy = (torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    wargs = (w, Cin, (1, 1, 1), lambda wd: ops.conv_weight_strides(wd))
    o32, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), (1, 1, 1), Cout, b, torch.float32, torch.float32)
    o16, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), (1, 1, 1), Cout, b, torch.float16, torch.float32)
    assert torch.equal(o16, o32)


@pytest.mark.parametrize("Cin,Cout,base,k,ocs_mult", [
    (60, 30, (4, 8, 32), (2, 2, 2), 1), (60, 30, (4, 8, 32), (2, 2, 2), 2), (64, 32, (4, 8, 32), (2, 2, 2), 2), (120, 60, (3, 5, 9), (2, 2, 2), 1),
    (60, 30, (5, 6, 16), (1, 2, 2), 1),
])
def test_transposed_conv_fp16_products(dev, Cin, Cout, base, k, ocs_mult):
    """mt_pointwise_t.mma = 1 (ABI v3): the forward transposed conv over fp16 activations multiplies in fp16 (pack layout 4) — against the host
    sum with the activated input and the weights rounded to fp16 (exact products), output within one fp16 rounding"""
    ops = _ops()
    g = torch.Generator().manual_seed(61)
    N = 2
    x = rbf(torch.randn((N,) + base + (Cin,), generator=g), torch.float16)
    lazy = (torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)
    w = torch.randn((Cin, Cout) + k, generator=g) / np.sqrt(Cin)
    wargs = (w, Cin, k, lambda wd: ops.conv_weight_strides(wd, transposed_layout=True))
    o16, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), k, Cout, None, torch.float16, torch.float16, ocs=Cout * ocs_mult, mma=1, want_layout=4)
    t = torch.addcmul(lazy[1][:, None, None, None, :], x, lazy[0][:, None, None, None, :])
    a = rbf(torch.maximum(t, t * 0.01), torch.float16).double()
    wh = rbf(w, torch.float16).double()
    ref = torch.einsum('ndhwi,ioabc->ndahbwco', a, wh).reshape((N, base[0] * k[0], base[1] * k[1], base[2] * k[2], Cout))
    got = o16[..., :Cout].float().cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max()) + 1e-6, float((got - ref).abs().max())
    # mma = 0 keeps the fp32-product kernel and layout 1
    _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), k, Cout, None, torch.float16, torch.float16, ocs=Cout * ocs_mult, mma=0, want_layout=1)


@pytest.mark.parametrize("Cin,Cout,lay", [(30, 47, 4), (60, 47, 4), (30, 2, 1), (120, 20, 1)])
def test_heads_fp16_products(dev, Cin, Cout, lay):
    """pw_head_kernel with fp16 products (33..64 logits); the narrow and generic head kernels keep fp32 products (layout 1)"""
    ops = _ops()
    g = torch.Generator().manual_seed(62)
    N, base = 2, (4, 8, 16)
    x = rbf(torch.randn((N,) + base + (Cin,), generator=g), torch.float16)
    lazy = (torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    wargs = (w, Cin, (1, 1, 1), lambda wd: ops.conv_weight_strides(wd))
    o, _ = _pointwise(dev, x, lazy, wargs, base, base, (1, 1, 1), (1, 1, 1), Cout, b, torch.float16, torch.float32, mma=1, want_layout=lay)
    t = torch.addcmul(lazy[1][:, None, None, None, :], x, lazy[0][:, None, None, None, :])
    a = torch.maximum(t, t * 0.01)
    if lay == 4:
        ref = torch.einsum('ndhwi,oi->ndhwo', rbf(a, torch.float16).double(), rbf(w[:, :, 0, 0, 0], torch.float16).double()) + b.double()
        tol = 1e-5
    else:
        ref = torch.einsum('ndhwi,oi->ndhwo', a.double(), w[:, :, 0, 0, 0].double()) + b.double()
        tol = 1e-5
    got = o.cpu().double()
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6, float((got - ref).abs().max())


@pytest.mark.parametrize("stride", [(2, 2, 2), (1, 2, 2)])
def test_strided_projection_pointwise_forward_and_scatter_backward(dev, stride):
    """the strided 1x1x1 skip projection of a residual block (conv_blocks.py:159-165) on the pointwise kernel: forward gathers every
    stride-th voxel (fp16 -> fp16 with statistics), backward-data scatters to them (mt_pointwise_t.scatter, bf16 -> bf16 accumulating)"""
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(33)
    N, Cin, Cout, shape = 2, 30, 60, (6, 12, 16)
    osp = tuple(-(-s // st) for s, st in zip(shape, stride))
    x = rbf(torch.randn((N,) + shape + (Cin,), generator=g), torch.float16)
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    wargs = (w, Cin, (1, 1, 1), lambda wd: ops.conv_weight_strides(wd))
    o16, part = _pointwise(dev, x, None, wargs, osp, shape, stride, (1, 1, 1), Cout, None, torch.float16, torch.float16, stats=True)
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3), w, None, stride=stride).permute(0, 2, 3, 4, 1)
    assert float((o16.float().cpu() - ref).abs().max()) < 2.0 ** -10 * float(ref.abs().max()) + 1e-6
    o = o16.float().double()
    sm = part.double().sum(1)
    assert torch.allclose(sm[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
    assert torch.allclose(sm[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    # backward-data: dX[stride * m] += W^T dY[m]
    dy = rbf(torch.randn((N,) + osp + (Cout,), generator=g))
    prev = rbf(torch.randn((N,) + shape + (Cin,), generator=g))
    wb = (w, Cout, (1, 1, 1), lambda wd: ops.conv_weight_strides(wd, as_bwd_data=True))
    d16, _ = _pointwise(dev, dy, None, wb, osp, osp, (1, 1, 1), stride, Cin, None, torch.bfloat16, torch.bfloat16, prev=prev, scatter=True)
    xr = torch.zeros((N, Cin) + shape, requires_grad=True)
    F.conv3d(xr, w, None, stride=stride).backward(dy.permute(0, 4, 1, 2, 3).contiguous())
    want = prev + xr.grad.permute(0, 2, 3, 4, 1)
    got = d16.float().cpu()
    assert float((got - want).abs().max()) < 2.0 ** -7 * float(want.abs().max())
    untouched = (xr.grad.permute(0, 2, 3, 4, 1) == 0).all(-1)              # voxels the convolution never read keep their old gradient
    assert torch.equal(got[untouched], prev[untouched])


@pytest.mark.parametrize("Cin,Cout", [(30, 47), (30, 2), (32, 4)])
@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("base", [(4, 8, 16), (3, 7, 11)])       # (3, 7, 11): V = 231 (odd), V * Cin * 2 % 16 != 0 — 16-byte dX pieces straddle the sample end
def test_head_backward_fp16_x_bf16_dx_bitexact(dev, Cin, Cout, acc, base):
    ops = _ops()
    g = torch.Generator().manual_seed(34)
    N = 3 if base[0] == 3 else 2
    V = int(np.prod(base))
    x = rbf(torch.randn((N,) + base + (Cin,), generator=g), torch.float16)
    sc, sh = torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g)
    dy = torch.randn((N,) + base + (Cout,), generator=g)
    prev = rbf(torch.randn((N,) + base + (Cin,), generator=g))
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    wd = w.to(dev).contiguous()
    wb = ops.pack_conv_weights(wd, Cout, 0, Cin, (1, 1, 1), ops.conv_weight_strides(wd, as_bwd_data=True), False, ops.POINTWISE_CK)
    st = ops.conv_weight_strides(wd)
    res = []
    for xdt, ddt in ((torch.float32, torch.float32), (torch.float16, torch.bfloat16)):
        a = ops.Act(x.to(dev).to(xdt), scale=sc.to(dev), shift=sh.to(dev), slope=0.01)
        dx = (prev if acc else torch.full(prev.shape, float('nan'))).to(dev).to(ddt)
        dyd = dy.to(dev)
        assert ops.head_bwd_io_supported(a, ops.Act(dx), Cout)
        dw = torch.zeros_like(wd)
        db = torch.zeros(Cout, device=dev)
        ws = torch.empty(ops.head_bwd_workspace(N, V, Cin, Cout) // 4 + 16, device=dev)
        done = ops.head_bwd(a, ops.Act(dyd), wb, ops.Act(dx), acc, dw, st[0], st[1], db, False, ws)
        torch.cuda.synchronize()
        res.append((dx, dw, db if done else None))
    assert torch.equal(res[1][0], res[0][0].to(torch.bfloat16))
    assert torch.equal(res[1][1], res[0][1])
    if res[0][2] is not None:
        assert torch.equal(res[1][2], res[0][2])


@pytest.mark.parametrize("k", [(2, 2, 2), (1, 2, 2)])
@pytest.mark.parametrize("acc", [False, True])
def test_gather_backward_of_transposed_conv_bf16_bitexact(dev, k, acc):
    ops = _ops()
    g = torch.Generator().manual_seed(35)
    N, Ct_in, Ct_out, low = 2, 60, 30, (4, 6, 16)
    hi = tuple(a * b for a, b in zip(low, k))
    dy = rbf(torch.randn((N,) + hi + (Ct_out,), generator=g))
    prev = rbf(torch.randn((N,) + low + (Ct_in,), generator=g))
    w = torch.randn((Ct_in, Ct_out) + k, generator=g) / np.sqrt(Ct_out * np.prod(k))
    wd = w.to(dev).contiguous()
    geom = ops.ConvGeom(hi, k, k, (0, 0, 0))
    res = []
    for dt in (torch.float32, torch.bfloat16):
        dyd = dy.to(dev).to(dt)
        dx = (prev if acc else torch.full(prev.shape, float('nan'))).to(dev).to(dt)
        p = ops.fill_conv([ops.Act(dyd)], geom, Ct_in, out0=ops.Act(dx), accumulate=acc)
        p.csplit = Ct_in
        assert ops.conv_kernel_name(p).startswith('conv_gather_kernel') and ops.conv_io_supported(p)
        wp = ops.pack_conv_weights(wd, Ct_out, 0, Ct_in, k, ops.conv_weight_strides(wd, transposed_layout=True, as_bwd_data=True), False, ops.conv_ck(p))
        p.wpack = wp.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        res.append(dx)
    assert torch.equal(res[1], res[0].to(torch.bfloat16))


@pytest.mark.parametrize("k", [(2, 2, 2), (1, 2, 2)])
@pytest.mark.parametrize("acc,odt", [(False, torch.bfloat16), (True, torch.bfloat16), (False, torch.float32)])
def test_gather_backward_of_transposed_conv_bf16_products(dev, k, acc, odt):
    """mixed precision (mma = 1): a bf16 gradient is the A fragment of v_mfma_f32_32x32x16_bf16 as loaded, the weights are pack layout 3 —
    against the host sum with the weights rounded to bf16 (exact products), channel tail 30 of 32"""
    ops = _ops()
    g = torch.Generator().manual_seed(38)
    N, Ct_in, Ct_out, low = 2, 60, 30, (4, 6, 21)
    hi = tuple(a * b for a, b in zip(low, k))
    dy = rbf(torch.randn((N,) + hi + (Ct_out,), generator=g))
    prev = rbf(torch.randn((N,) + low + (Ct_in,), generator=g), odt) if odt != torch.float32 else torch.randn((N,) + low + (Ct_in,), generator=g)
    w = torch.randn((Ct_in, Ct_out) + k, generator=g) / np.sqrt(Ct_out * np.prod(k))
    wd = w.to(dev).contiguous()
    geom = ops.ConvGeom(hi, k, k, (0, 0, 0))
    dyd = dy.to(dev).to(torch.bfloat16)
    dx = (prev if acc else torch.full(prev.shape, float('nan'))).to(dev).to(odt)
    p = ops.fill_conv([ops.Act(dyd)], geom, Ct_in, out0=ops.Act(dx), accumulate=acc, mma=1)
    p.csplit = Ct_in
    name = ops.conv_kernel_name(p)
    assert name.startswith('conv_gather_kernel') and name.endswith('true>') and ops.conv_io_supported(p), name
    assert ops.conv_pack_layout(p) == 3
    wp = ops.pack_conv_weights(wd, Ct_out, 0, Ct_in, k, ops.conv_weight_strides(wd, transposed_layout=True, as_bwd_data=True), False, ops.conv_ck(p),
                               layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    # host: dX[n, v, ci] = sum over taps, co of dY[n, s v + tap, co] * bf16(W[ci, co, tap])
    wb = rbf(w).double()
    d = dy.double().reshape((N, low[0], k[0], low[1], k[1], low[2], k[2], Ct_out))
    ref = torch.einsum('ndahbwcq,iqabc->ndhwi', d, wb)
    if acc:
        ref = ref + prev.double()
    got = dx.float().cpu().double()
    assert torch.isfinite(got).all()
    tol = 2.0 ** -8 if odt != torch.float32 else 1e-5
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6, float((got - ref).abs().max())


@pytest.mark.parametrize("kind", ["strided222", "strided122", "tconv222", "head", "proj222", "k133"])
def test_tiled_backward_weight_16bit_storage_bitexact(dev, kind):
    """conv_bwdw_fast_kernel: X / dY widened on load, fp32 products — identical to the fp32-storage launch"""
    ops = _ops()
    g = torch.Generator().manual_seed(36)
    N = 2
    cfg = {'strided222': ((3, 3, 3), (2, 2, 2), (1, 1, 1), 30, 60, (8, 12, 34), torch.float16, torch.bfloat16, True),
           'strided122': ((3, 3, 3), (1, 2, 2), (1, 1, 1), 32, 64, (5, 12, 34), torch.float16, torch.bfloat16, True),
           'tconv222': ((2, 2, 2), (2, 2, 2), (0, 0, 0), 30, 60, (8, 12, 36), torch.bfloat16, torch.float16, False),   # X = dOut, Y = tconv input (lazy)
           'head': ((1, 1, 1), (1, 1, 1), (0, 0, 0), 60, 47, (4, 8, 34), torch.float16, torch.float32, True),
           'proj222': ((1, 1, 1), (2, 2, 2), (0, 0, 0), 30, 60, (8, 12, 34), torch.float16, torch.bfloat16, True),
           'k133': ((1, 3, 3), (1, 1, 1), (0, 1, 1), 30, 30, (4, 8, 12), torch.float16, torch.bfloat16, True)}[kind]
    k, stride, pad, Cin, Cout, shape, xdt, ydt, xlazy = cfg
    geom = ops.ConvGeom(shape, k, stride, pad)
    x = rbf(torch.randn((N,) + shape + (Cin,), generator=g), xdt)
    y = rbf(torch.randn((N,) + tuple(geom.out) + (Cout,), generator=g), ydt if ydt != torch.float32 else torch.bfloat16)
    lz = (torch.rand((N, Cin if xlazy else Cout), generator=g) + 0.5, torch.randn((N, Cin if xlazy else Cout), generator=g))
    res = []
    ops.set_option('bwdw_cw', 1)       # (several cout tiles per workgroup exist for fp32 storage only: same products, another order of the sums)
    try:
        for xd, yd in ((torch.float32, torch.float32), (xdt, ydt)):
            xb, yb = x.to(dev).to(xd), y.to(dev).to(yd)
            a = ops.Act(xb, scale=lz[0].to(dev), shift=lz[1].to(dev), slope=0.01) if xlazy else ops.Act(xb)
            ya = ops.Act(yb) if xlazy else ops.Act(yb, scale=lz[0].to(dev), shift=lz[1].to(dev), slope=0.01)
            p = ops.fill_conv([a], geom, Cout, mma=0)
            name = ops.conv_bwd_weight_kernel_name(p, ya)
            assert name.startswith('conv_bwdw_fast_kernel'), name
            assert ops.conv_bwd_weight_io_supported(p, ya), (name, xd, yd)
            dw = torch.full((Cout, Cin) + k, float('nan'), device=dev)
            ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
            torch.cuda.synchronize()
            res.append(dw)
    finally:
        ops.set_option('bwdw_cw', 4)
    assert torch.isfinite(res[1]).all() and torch.equal(res[0], res[1]), float((res[0] - res[1]).abs().max())


@pytest.mark.parametrize("kind", ["strided222", "strided222_narrow", "strided222_deep", "strided122", "tconv222", "tconv122", "proj222", "proj122", "k333_shallow", "two_sources",
                                  "strided222_narrow_cw4", "strided122_narrow_cw2", "tconv222_cw4", "strided222_cw4", "strided122_deep_cw2"])
def test_tiled_backward_weight_bf16_products(dev, kind):
    """conv_bwdw_fast16_kernel (mixed precision: 16-bit X and dY, bf16 products, fp32 accumulation) against autograd on the host with the
    same operand rounding (activated operands rounded to bf16, exact products); ragged tiles, both tile shapes, accumulate"""
    import torch.nn.functional as F
    ops = _ops()
    ops.set_option('bwdw_cw', 104)          # several cout tiles per workgroup also on volumes this small
    try:
        _tiled_backward_weight_bf16_products(dev, kind)
    finally:
        ops.set_option('bwdw_cw', 4)


def _tiled_backward_weight_bf16_products(dev, kind):
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(46)
    N = 2
    f16, b16, f32 = torch.float16, torch.bfloat16, torch.float32
    cfg = {'strided222': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (30,), 60, (8, 12, 70), f16, b16, True),
           'strided222_narrow': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (64,), 40, (6, 18, 26), f16, b16, True),      # Wo <= 16: the 8 x 16 tile
           'strided222_deep': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (32,), 64, (23, 10, 66), f16, b16, True),        # odd depth: 12 output planes, ragged h tiles, D segments
           'strided122': ((3, 3, 3), (1, 2, 2), (1, 1, 1), (32,), 64, (5, 12, 34), f16, b16, True),
           'tconv222': ((2, 2, 2), (2, 2, 2), (0, 0, 0), (30,), 60, (8, 12, 36), b16, f16, False),   # X = dOut, Y = tconv input (lazy)
           'tconv122': ((1, 2, 2), (1, 2, 2), (0, 0, 0), (32,), 64, (3, 8, 72), b16, f16, False),
           'proj222': ((1, 1, 1), (2, 2, 2), (0, 0, 0), (30,), 60, (8, 12, 34), f16, b16, True),
           'proj122': ((1, 1, 1), (1, 2, 2), (0, 0, 0), (32,), 64, (3, 12, 34), f16, b16, True),
           'k333_shallow': ((3, 3, 3), (1, 1, 1), (1, 1, 1), (30,), 30, (2, 9, 40), f16, b16, True),          # Do < 3: not a marching problem
           'two_sources': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (30, 18), 60, (6, 10, 36), f16, b16, True),
           # several cout tiles per workgroup (option bwdw_cw): four / two (ten cout tiles) / four; marching form: four / two
           'strided222_narrow_cw4': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (48,), 128, (6, 18, 26), f16, b16, True),
           'strided122_narrow_cw2': ((3, 3, 3), (1, 2, 2), (1, 1, 1), (40,), 320, (3, 8, 16), f16, b16, True),
           'tconv222_cw4': ((2, 2, 2), (2, 2, 2), (0, 0, 0), (30,), 120, (8, 12, 36), b16, f16, False),
           'strided222_cw4': ((3, 3, 3), (2, 2, 2), (1, 1, 1), (30,), 120, (8, 12, 70), f16, b16, True),
           'strided122_deep_cw2': ((3, 3, 3), (1, 2, 2), (1, 1, 1), (32,), 64, (7, 18, 40), f16, b16, True)}[kind]
    k, stride, pad, Cins, Cout, shape, xdt, ydt, xlazy = cfg
    geom = ops.ConvGeom(shape, k, stride, pad)
    xs = [rbf(torch.randn((N,) + shape + (C,), generator=g), xdt) for C in Cins]
    y = rbf(torch.randn((N,) + tuple(geom.out) + (Cout,), generator=g), ydt if ydt != f32 else b16)
    xl = [(torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g)) for C in Cins]
    yl = (torch.rand((N, Cout), generator=g) + 0.5, torch.randn((N, Cout), generator=g))
    keep = [x.to(dev).to(xdt) for x in xs]
    acts = [ops.Act(b, scale=l[0].to(dev), shift=l[1].to(dev), slope=0.01) if xlazy else ops.Act(b) for b, l in zip(keep, xl)]
    yb = y.to(dev).to(ydt)
    ya = ops.Act(yb) if xlazy else ops.Act(yb, scale=yl[0].to(dev), shift=yl[1].to(dev), slope=0.01)
    p = ops.fill_conv(acts, geom, Cout, mma=1)
    name = ops.conv_bwd_weight_kernel_name(p, ya)
    # the strided 3x3x3 stage convs with Wo > 16 and Do >= 3 take the marching form (a ring of input planes), everything else the tiled one
    marching = kind in ('strided222', 'strided122', 'two_sources', 'strided222_deep', 'strided222_cw4', 'strided122_deep_cw2')
    assert name.startswith('conv_bwdw_march16_kernel' if marching else 'conv_bwdw_fast16_kernel'), name
    assert ops.conv_bwd_weight_io_supported(p, ya), name
    Cin = sum(Cins)
    dw = torch.full((Cout, Cin) + k, float('nan'), device=dev)
    ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
    ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
    torch.cuda.synchronize()

    def act(t, l):
        u = torch.addcmul(l[1][:, None, None, None, :], t, l[0][:, None, None, None, :])
        return torch.maximum(u, u * 0.01)
    xa = [rbf(act(x, l)) if xlazy else x for x, l in zip(xs, xl)]              # (a bf16 X without activation is copied exactly)
    yy = y if xlazy else rbf(act(y, yl))
    xin = torch.cat(xa, -1).permute(0, 4, 1, 2, 3).double()
    w0 = torch.zeros((Cout, Cin) + k, dtype=torch.float64, requires_grad=True)
    F.conv3d(xin, w0, None, stride=stride, padding=pad).backward(yy.permute(0, 4, 1, 2, 3).double().contiguous())
    ref = w0.grad.float()
    got = dw.cpu()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-3, err                       # (fma vs mul + add before the bf16 rounding of the activated operand: isolated operand ulps)
    dw2 = dw.clone()
    ops.conv3d_bwd_weight(p, ya, dw2, ops.conv_weight_strides(dw2), True, ws)
    torch.cuda.synchronize()
    assert torch.allclose(dw2, 2 * dw, rtol=1e-6, atol=1e-6)
    if True:                  # one cout tile per workgroup: the same products, fp32 sums in another order
        ops.set_option('bwdw_cw', 1)
        try:
            dw1 = torch.full((Cout, Cin) + k, float('nan'), device=dev)
            ops.conv3d_bwd_weight(ops.apply_selection(p), ya, dw1, ops.conv_weight_strides(dw1), False, ws)
            torch.cuda.synchronize()
        finally:
            ops.set_option('bwdw_cw', 104)
            ops.apply_selection(p)
        assert float((dw1 - dw).abs().max()) <= 1e-5 * float(dw.abs().max())
    # mma decides between kernels of the same result up to the product type: the fp32-product kernel is within bf16 operand rounding
    p0 = ops.fill_conv(acts, geom, Cout, mma=0)
    assert ops.conv_bwd_weight_kernel_name(p0, ya).startswith('conv_bwdw_fast_kernel')


def test_stem_kernels_16bit_storage_bitexact(dev):
    """first convolution (one input channel, fp32 network input): fp16 output with statistics; its backward-weight reads a bf16 dY"""
    ops = _ops()
    g = torch.Generator().manual_seed(37)
    N, Cout, shape = 2, 30, (6, 10, 40)
    geom = ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    x = torch.randn((N,) + shape + (1,), generator=g).to(dev)
    w = torch.randn((Cout, 1, 3, 3, 3), generator=g) / 5
    b = torch.randn(Cout, generator=g).to(dev)
    wd = w.to(dev).contiguous()
    outs = []
    for odt in (torch.float32, torch.float16):
        out = torch.full((N,) + shape + (Cout,), float('nan'), device=dev).to(odt)
        p = ops.fill_conv([ops.Act(x)], geom, Cout, out0=ops.Act(out), bias=b)
        assert ops.conv_kernel_name(p).startswith('conv_stem_kernel') and ops.conv_io_supported(p)
        wp = ops.pack_conv_weights(wd, 1, 0, Cout, (3, 3, 3), ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
        p.stats_part = part.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        outs.append((out, part))
    assert torch.equal(outs[1][0], outs[0][0].to(torch.float16))
    o = outs[1][0].float().double()
    sm = outs[1][1].double().sum(1)
    assert torch.allclose(sm[..., 0], o.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * o[0, ..., 0].numel() ** 0.5)
    assert torch.allclose(sm[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-4)
    dy = rbf(torch.randn((N,) + shape + (Cout,), generator=g))
    res = []
    for ydt in (torch.float32, torch.bfloat16):
        ya = ops.Act(dy.to(dev).to(ydt))
        p = ops.fill_conv([ops.Act(x)], geom, Cout)
        assert ops.conv_bwd_weight_kernel_name(p, ya).startswith('conv_bwdw_stem_kernel') and ops.conv_bwd_weight_io_supported(p, ya)
        dw = torch.full((Cout, 1, 3, 3, 3), float('nan'), device=dev)
        ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
        ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
        torch.cuda.synchronize()
        res.append(dw)
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("Cins,Cout,shape,k", [((320,), 320, (6, 12, 12), (3, 3, 3)), ((320,), 320, (3, 6, 6), (3, 3, 3)),
                                               ((320, 320), 320, (6, 12, 12), (3, 3, 3)), ((48,), 40, (5, 7, 9), (3, 3, 3)),
                                               ((30,), 30, (2, 8, 16), (1, 3, 3))])
@pytest.mark.parametrize("xdt,ydt", [(torch.float16, torch.bfloat16), (torch.float32, torch.float32)])
def test_lowres_backward_weight_im2col_gemm(dev, Cins, Cout, shape, k, xdt, ydt):
    """bwdw_gemm_kernel (mixed precision, W <= 16): dW = im2col(act(X))^T dY with bf16 products — against autograd on the host with the
    same operand rounding (activated X and dY rounded to bf16, exact products)"""
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    N = 2
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
    xs = [rbf(torch.randn((N,) + shape + (C,), generator=g), xdt if xdt != torch.float32 else torch.bfloat16) for C in Cins]
    lz = [(torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g)) for C in Cins]
    dy = rbf(torch.randn((N,) + shape + (Cout,), generator=g))
    keep = [x.to(dev).to(xdt) for x in xs]
    acts = [ops.Act(b, scale=l[0].to(dev), shift=l[1].to(dev), slope=0.01) for b, l in zip(keep, lz)]
    ya = ops.Act(dy.to(dev).to(ydt))
    p = ops.fill_conv(acts, geom, Cout, mma=1)
    assert ops.conv_bwd_weight_kernel_name(p, ya) == 'bwdw_gemm_kernel', ops.conv_bwd_weight_kernel_name(p, ya)
    assert ops.conv_bwd_weight_io_supported(p, ya)
    Cin = sum(Cins)
    dw = torch.full((Cout, Cin) + k, float('nan'), device=dev)
    ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 64, device=dev)
    ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
    torch.cuda.synchronize()
    # host: activated inputs rounded to bf16, exact products
    xa = []
    for x, l in zip(xs, lz):
        t = torch.addcmul(l[1][:, None, None, None, :], x, l[0][:, None, None, None, :])
        xa.append(rbf(torch.maximum(t, t * 0.01)))
    xin = torch.cat(xa, -1).permute(0, 4, 1, 2, 3).double()
    w0 = torch.zeros((Cout, Cin) + k, dtype=torch.float64, requires_grad=True)
    F.conv3d(xin, w0, None, padding=pad).backward(dy.permute(0, 4, 1, 2, 3).double().contiguous())
    ref = w0.grad.float()
    got = dw.cpu()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-3, err                       # (fma vs mul + add before the bf16 rounding of the activated input: isolated operand ulps)
    # accumulate
    dw2 = dw.clone()
    ops.conv3d_bwd_weight(p, ya, dw2, ops.conv_weight_strides(dw2), True, ws)
    torch.cuda.synchronize()
    assert torch.allclose(dw2, 2 * dw, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("nonlin", [1, 2])
def test_fused_inference_heads_take_the_16bit_source(dev, nonlin):
    """mt_head_flip_accumulate / mt_head_mirror_accumulate over the fp16 decoder output of the mixed mode: identical to the same launch
    over an fp32 copy of the stored values (the kernels widen exactly and compute in fp32)"""
    import test_mixed_precision_gpu as M
    from multitalent_amd.engine import HeadOp
    ops = _ops()
    net = M._net(dev)
    net.eval()
    eng = net.engine()
    eng.set_precision('bf16')
    try:
        g = torch.Generator().manual_seed(12)
        patch = (16, 32, 48)
        x = torch.randn((2, 1) + patch, generator=g).to(dev)
        with torch.no_grad():
            hp = eng.forward_to_final_head(x)
        op = next(o for o in eng.ops if isinstance(o, HeadOp) and o.out is eng.heads[eng.final_head])
        a16 = op.srcs[0].act
        assert a16.dtype == torch.float16 and hp.src.dtype == 2, (a16.dtype, hp.src.dtype)      # no cast in front of the fused head
        a32 = a16.with_buf(a16.buf.float())
        hp32 = ops.fill_pointwise(a32, op.geom.out, a32.spatial, (1, 1, 1), (1, 1, 1), op.conv.out_channels, op.wf, op.conv.bias, op.out.act)
        C = op.conv.out_channels
        res = []
        for p in (hp, hp32):
            acc = torch.full((C,) + patch, float('nan'), device=dev)
            ops.head_flip_accumulate(p, 1, (True, False, True), nonlin, 0.5, acc, True)
            ops.head_flip_accumulate(p, 0, (False, True, False), nonlin, 0.5, acc, False)
            shape = (20, 40, 56)
            agg = torch.zeros((C,) + shape, device=dev)
            nb = torch.zeros(shape, device=dev)
            gs = torch.rand(patch, generator=g).to(dev) if not res else res[0][3]
            ops.head_mirror_accumulate(p, 0, [(False, False, False), (True, True, False)], nonlin, 0.5, gs, agg, nb, shape, (2, 4, 8))
            torch.cuda.synchronize()
            res.append((acc, agg, nb, gs))
        assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        assert float(res[0][1].abs().max()) > 0
    finally:
        eng.set_precision('fp32')
