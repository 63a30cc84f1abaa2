"""Parity of every HIP kernel (called through the C ABI) against the oracle's torch-fp32 CPU ops.
Tolerances: 1e-4 relative-to-scale for single kernels (fp32 MFMA = exact fp32 FMA chains; only the
summation order differs), stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from multitalent_amd import ops
    return ops


def to_ndhwc(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def _bf16_round(x):
    from oracle.reference_ops import bf16_round
    return bf16_round(x)


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_conv(dev, srcs_cpu, w, b, stride, pad, lazy=None, stats=False):
    """srcs_cpu: list of NCDHW cpu tensors (raw); lazy: list of (scale[N,C], shift[N,C], slope) or None per src."""
    ops = _ops()
    N = srcs_cpu[0].shape[0]
    acts = []
    for i, s in enumerate(srcs_cpu):
        buf = to_ndhwc(s).to(dev)
        if lazy is not None and lazy[i] is not None:
            sc, sh, sl = lazy[i]
            acts.append(ops.Act(buf, scale=sc.to(dev).contiguous(), shift=sh.to(dev).contiguous(), slope=sl))
        else:
            acts.append(ops.Act(buf))
    Cout = w.shape[0]
    geom = ops.ConvGeom(srcs_cpu[0].shape[2:], w.shape[2:], stride, pad)
    out = torch.full((N,) + geom.out + (Cout,), float('nan'), device=dev)
    bd = b.to(dev) if b is not None else None   # keep alive: the struct only holds raw pointers
    p = ops.fill_conv(acts, geom, Cout, out0=ops.Act(out), bias=bd)
    ck = ops.conv_ck(p)
    wd = w.to(dev).contiguous()
    C0 = acts[0].C
    C1 = acts[1].C if len(acts) > 1 else 0
    wp = ops.pack_conv_weights(wd, C0, C1, Cout, w.shape[2:], ops.conv_weight_strides(wd), False, ck, layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    part = None
    if stats:
        nsb = ops.conv_stats_blocks(p)
        part = torch.zeros((N, nsb, Cout, 2), device=dev)
        p.stats_part = part.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    return out, part


def ref_inputs(srcs_cpu, lazy):
    xs = []
    for i, s in enumerate(srcs_cpu):
        if lazy is not None and lazy[i] is not None:
            sc, sh, sl = lazy[i]
            t = s * sc[:, :, None, None, None] + sh[:, :, None, None, None]
            t = torch.where(t > 0, t, t * sl)
            xs.append(t)
        else:
            xs.append(s)
    return torch.cat(xs, 1)


@pytest.mark.parametrize("N,Cin,Cout,shape,k,stride", [
    (1, 30, 30, (4, 8, 32), (3, 3, 3), (1, 1, 1)),
    (2, 30, 30, (5, 9, 37), (3, 3, 3), (1, 1, 1)),      # ragged tiles
    (1, 1, 30, (4, 8, 32), (3, 3, 3), (1, 1, 1)),        # stem (taps-as-K kernel)
    (2, 1, 40, (5, 9, 37), (3, 3, 3), (1, 1, 1)),        # stem, ragged tiles, two cout tiles
    (1, 30, 60, (6, 12, 34), (3, 3, 3), (2, 2, 2)),      # strided
    (1, 60, 47, (3, 6, 12), (1, 3, 3), (1, 1, 1)),       # anisotropic kernel, Cout not multiple of 32
    (1, 320, 320, (3, 6, 6), (3, 3, 3), (1, 2, 2)),      # bottleneck
    (1, 30, 2, (4, 8, 16), (1, 1, 1), (1, 1, 1)),        # 1x1x1 via the conv kernel
    (1, 17, 33, (3, 5, 7), (3, 3, 3), (1, 1, 1)),        # odd everything
    (2, 16, 64, (24, 48, 64), (3, 3, 3), (1, 1, 1)),     # >= 512 workgroups: the full-tile kernel (smaller grids take the tap-split kernel)
    (2, 20, 40, (23, 45, 70), (3, 3, 3), (1, 1, 1)),     # same, ragged
])
def test_conv_fwd(dev, N, Cin, Cout, shape, k, stride):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, Cin) + shape, generator=g)
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    b = torch.randn(Cout, generator=g)
    pad = tuple((kk - 1) // 2 for kk in k)
    out, part = run_conv(dev, [x], w, b, stride, pad, stats=True)
    ref = F.conv3d(x, w, b, stride=stride, padding=pad)
    got = to_ncdhw(out.cpu())
    assert got.shape == ref.shape
    if k == (3, 3, 3) and stride == (1, 1, 1):
        ops = _ops()
        geom = ops.ConvGeom(shape, k, stride, pad)
        name = ops.conv_kernel_name(ops.fill_conv([ops.Act(torch.empty((N,) + shape + (Cin,), device=dev))], geom, Cout,
                                                  out0=ops.Act(out)))
        big = N * np.prod([-(-s // t) for s, t in zip(shape, (2, 4, 32))]) * -(-Cout // 32) >= 300
        assert name.startswith(('conv_stem_kernel',) if Cin == 1 else (('conv_fast_kernel', 'conv_wino') if big else ('conv_tapsplit_kernel',))), name
    assert relerr(got, ref) < 1e-5
    # per-block statistics partials sum to per-(n,c) sums
    s = part.cpu().double().sum(1)
    V = ref[0, 0].numel()
    assert np.allclose(s[..., 0].numpy(), ref.double().sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3 * np.sqrt(V))
    assert np.allclose(s[..., 1].numpy(), (ref.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)


@pytest.mark.parametrize("N,cins,Cout,shape,stride,lazy", [
    (2, (240,), 320, (6, 24, 24), (2, 2, 2), True),       # the 120-workgroup stage conv of the benchmark networks
    (2, (320,), 320, (3, 12, 12), (1, 2, 2), True),       # bottleneck, 40 workgroups in the standard tiling
    (1, (30,), 60, (6, 12, 34), (2, 2, 2), False),        # ragged tiles in every direction
    (2, (24, 16), 70, (5, 9, 11), (2, 2, 2), True),       # odd extents, two sources, three cout tiles with a tail
    (1, (40,), 33, (3, 7, 9), (1, 2, 2), False),
])
def test_conv_tapsplit_strided(dev, N, cins, Cout, shape, stride, lazy):
    """conv_tapsplit_kernel<2, false, 0, 0, 1, SD, 2, 2>: the strided 3x3x3 stage convs on under-filled grids with the taps split over
    the waves, against F.conv3d (fp32) and against conv_fast_strided_kernel (option conv_tapsplit = 0), statistics included."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    xs = [torch.randn((N, ci) + shape, generator=g) for ci in cins]
    lz = [(torch.rand((N, ci), generator=g) + 0.5, torch.randn((N, ci), generator=g) * 0.3, 0.01) for ci in cins] if lazy else None
    Cin = sum(cins)
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / np.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    ref = F.conv3d(ref_inputs(xs, lz), w, b, stride=stride, padding=1)
    geom = ops.ConvGeom(shape, (3, 3, 3), stride, (1, 1, 1))
    probe = ops.fill_conv([ops.Act(torch.empty((N,) + shape + (ci,), device=dev)) for ci in cins], geom, Cout,
                          out0=ops.Act(torch.empty((N,) + tuple(geom.out) + (Cout,), device=dev)))
    res = {}
    try:
        for mode, kernel in ((1, 'conv_tapsplit_kernel<2, false, 0, 0, 1, %d, 2, 2>' % stride[0]), (0, 'conv_fast_strided')):
            ops.set_option('conv_tapsplit', mode)
            assert ops.conv_kernel_name(ops.apply_selection(probe)).startswith(kernel), ops.conv_kernel_name(probe)
            out, part = run_conv(dev, xs, w, b, stride, (1, 1, 1), lazy=lz, stats=True)
            got = to_ncdhw(out.cpu())
            assert relerr(got, ref) < 1e-5, mode
            sums = part.cpu().double().sum(1)
            V = ref[0, 0].numel()
            assert np.allclose(sums[..., 0].numpy(), ref.double().sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3 * np.sqrt(V))
            assert np.allclose(sums[..., 1].numpy(), (ref.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
            res[mode] = got
    finally:
        ops.set_option('conv_tapsplit', 1)
    assert relerr(res[1], res[0]) < 1e-5


def test_conv_fwd_two_lazy_sources(dev):
    """concat(tconv output [identity], skip [InstanceNorm+LeakyReLU on load]) -> conv, generic_UNet.py:392."""
    g = torch.Generator().manual_seed(2)
    N, shape = 2, (4, 8, 32)
    x0 = torch.randn((N, 30) + shape, generator=g)
    x1 = torch.randn((N, 30) + shape, generator=g)
    sc = torch.rand((N, 30), generator=g) + 0.5
    sh = torch.randn((N, 30), generator=g)
    w = torch.randn((30, 60, 3, 3, 3), generator=g) / 40
    b = torch.randn(30, generator=g)
    lazy = [None, (sc, sh, 0.01)]
    out, _ = run_conv(dev, [x0, x1], w, b, (1, 1, 1), (1, 1, 1), lazy=lazy)
    ref = F.conv3d(ref_inputs([x0, x1], lazy), w, b, padding=1)
    assert relerr(to_ncdhw(out.cpu()), ref) < 1e-5


@pytest.mark.parametrize("Cin,Cout,shape,k,stride,pad", [
    (30, 60, (6, 12, 34), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (32, 64, (7, 13, 33), (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # odd input sizes: classes of different extent
    (30, 60, (6, 12, 32), (3, 3, 3), (1, 2, 2), (1, 1, 1)),
    (24, 40, (5, 8, 10), (1, 1, 1), (1, 2, 2), (0, 0, 0)),        # strided projection: three classes carry no taps
    (16, 24, (4, 8, 8), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
])
def test_conv_bwd_data_parity_classes(dev, Cin, Cout, shape, k, stride, pad):
    """dX of a strided nn.Conv3d as one exact stride-1 convolution per parity class of the input position, written in place
    with output stride/offset (no multiplication of inserted zeros)."""
    ops = _ops()
    g = torch.Generator().manual_seed(33)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g, requires_grad=True)
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    y = F.conv3d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dyb = ops.Act(to_ndhwc(dy).to(dev))
    fgeom = ops.ConvGeom(shape, k, stride, pad)
    assert fgeom.out == tuple(dy.shape[2:])
    base = torch.randn((N,) + shape + (Cin,), generator=g)
    dx = base.to(dev)                                   # accumulate on top of an existing gradient
    wd = w.to(dev).contiguous()
    keep = []
    for geomc, place, tapmap in ops.bwd_data_parity_classes(fgeom):
        p = ops.fill_conv([dyb], geomc, Cin, out0=ops.Act(dx), accumulate=True, place=place)
        wp = ops.pack_conv_weights(wd, Cout, 0, Cin, geomc.k, ops.conv_weight_strides(wd, as_bwd_data=True), False,
                                   ops.conv_ck(p), tapmap=tapmap)
        keep.append(wp)
        p.wpack = wp.data_ptr()
        assert ops.conv_kernel_name(p).startswith('conv_rt_kernel')
        ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    assert relerr(to_ncdhw(dx.cpu()) - to_ncdhw(base), x.grad) < 1e-5


@pytest.mark.parametrize("Cin,Cout,shape,stride,acc", [
    (30, 60, (6, 12, 34), (2, 2, 2), False),
    (32, 64, (7, 13, 33), (2, 2, 2), True),        # odd input sizes: the last parity class is one position shorter
    (30, 60, (6, 12, 32), (1, 2, 2), False),
    (64, 33, (5, 9, 40), (1, 2, 2), True),
    (17, 20, (3, 5, 7), (2, 2, 2), False),
    (240, 320, (6, 24, 24), (2, 2, 2), True),      # the benchmark networks' under-filled layers
    (320, 320, (3, 12, 12), (1, 2, 2), True),
])
@pytest.mark.parametrize("mma", [0, 1])
def test_conv_bwd_data_strided_one_launch(dev, Cin, Cout, shape, stride, acc, mma):
    """mt_conv3d_bwd_data_strided: all parity classes of dX from one staged dY tile, vs autograd of F.conv3d.  mma = 1: the bf16
    variant (and the bf16 strided FORWARD kernel), against the same arithmetic with bf16-rounded operands (1e-4)."""
    ops = _ops()
    g = torch.Generator().manual_seed(34)
    N, k, pad = 2, (3, 3, 3), (1, 1, 1)
    x = torch.randn((N, Cin) + shape, generator=g, requires_grad=True)
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * 27)
    rnd = _bf16_round if mma else (lambda v: v)
    y = F.conv3d(x.double() if mma else x, rnd(w).double() if mma else w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(rnd(dy).double() if mma else dy)
    if mma and Cin >= 16 and Cin % 2 == 0:       # (odd or small Cin stays on the exact fp32 kernel)
        ops.set_mma(1)
        try:       # forward strided kernel in bf16 on the same layer
            out, part = run_conv(dev, [x.detach()], w, None, stride, pad, stats=True)
            ref = F.conv3d(_bf16_round(x.detach()).double(), _bf16_round(w).double(), None, stride=stride, padding=pad)
            assert relerr(to_ncdhw(out.cpu()), ref) < 1e-4
            assert np.allclose(part.cpu().double().sum(1)[..., 1].numpy(), (ref ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
        finally:
            ops.set_mma(0)
    # dY arrives lazily activated in the residual encoder; exercise that path too: dy = lrelu(raw*sc+sh)
    dyb = ops.Act(to_ndhwc(dy).to(dev))
    fgeom = ops.ConvGeom(shape, k, stride, pad)
    base = torch.randn((N,) + shape + (Cin,), generator=g) if acc else torch.full((N,) + shape + (Cin,), float('nan'))
    dx = base.to(dev)
    wd = w.to(dev).contiguous()
    p = ops.fill_conv([dyb], fgeom, Cout, out0=ops.Act(dx), accumulate=acc, mma=mma)
    p.Cin = Cin
    assert ops.conv3d_bwd_data_strided_supported(p)
    lay = ops.conv_bwd_data_strided_pack_layout(p)
    assert lay == (3 if (mma and Cout >= 16 and Cout % 2 == 0) else 1)
    wp = ops.pack_conv_weights(wd, Cout, 0, Cin, k, ops.conv_weight_strides(wd, as_bwd_data=True), False, 16, layout=lay)
    p.wpack = wp.data_ptr()
    # grids this small take conv_bwdd_strided_ks_kernel (one tile per workgroup, K split over the waves) where the types fit;
    # option conv_tapsplit = 0: the full-tile kernel
    names = set()
    try:
        for ts in (1, 0):
            ops.set_option('conv_tapsplit', ts)
            ops.apply_selection(p)
            names.add(ops.conv_bwd_data_strided_kernel_name(p).split('<')[0])
            dx.copy_(base.to(dev))
            ops.conv3d_bwd_data_strided(p)
            torch.cuda.synchronize()
            got = to_ncdhw(dx.cpu())
            if acc:
                got = got - to_ncdhw(base)
            assert relerr(got, x.grad) < (1e-4 if lay == 3 else (2e-2 if mma else 1e-5)), ts
    finally:
        ops.set_option('conv_tapsplit', 1)
    assert names == ({'conv_bwdd_strided_ks_kernel', 'conv_bwdd_strided_kernel'} if (lay == 1 and Cout % 2 == 0) else {'conv_bwdd_strided_kernel'}), names


@pytest.mark.parametrize("Cin,Cout,shape,k,stride", [
    (30, 30, (4, 8, 32), (3, 3, 3), (1, 1, 1)),
    (30, 60, (6, 12, 34), (3, 3, 3), (2, 2, 2)),
    (30, 60, (6, 12, 32), (3, 3, 3), (1, 2, 2)),
    (24, 40, (5, 7, 9), (1, 3, 3), (1, 1, 1)),
])
def test_conv_bwd_data(dev, Cin, Cout, shape, k, stride):
    """dX of nn.Conv3d = stride-1 conv of the zero-inserted dY with flipped, transposed weights."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g, requires_grad=True)
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    pad = tuple((kk - 1) // 2 for kk in k)
    y = F.conv3d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dyb = ops.Act(to_ndhwc(dy).to(dev))
    geom = ops.ConvGeom(dy.shape[2:], k, (1, 1, 1), tuple(kk - 1 - pp for kk, pp in zip(k, pad)), dil=stride, out_spatial=shape)
    dx = torch.full((N,) + shape + (Cin,), float('nan'), device=dev)
    p = ops.fill_conv([dyb], geom, Cin, out0=ops.Act(dx))
    ck = ops.conv_ck(p)
    wd = w.to(dev).contiguous()
    wp = ops.pack_conv_weights(wd, Cout, 0, Cin, k, ops.conv_weight_strides(wd, as_bwd_data=True), True, ck, layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    assert relerr(to_ncdhw(dx.cpu()), x.grad) < 1e-5


@pytest.mark.parametrize("Cin,Cout,shape,k,stride", [
    (30, 30, (4, 8, 32), (3, 3, 3), (1, 1, 1)),
    (30, 60, (6, 12, 34), (3, 3, 3), (2, 2, 2)),
    (60, 47, (3, 6, 12), (1, 1, 1), (1, 1, 1)),
    (17, 33, (3, 5, 7), (3, 3, 3), (1, 1, 1)),
    (40, 24, (5, 7, 9), (1, 3, 3), (1, 2, 2)),
    (1, 32, (4, 8, 32), (3, 3, 3), (1, 1, 1)),       # stem: taps-as-M kernel
    (1, 40, (5, 9, 37), (3, 3, 3), (1, 1, 1)),       # stem, ragged tiles, two cout tiles
    (32, 32, (6, 8, 32), (3, 3, 3), (1, 1, 1)),      # marching kernel, several planes per column
    (24, 40, (5, 9, 37), (3, 3, 3), (1, 1, 1)),      # Winograd backward-weight: ragged tiles in H and W, two cout tiles
    (16, 70, (4, 6, 40), (3, 3, 3), (1, 1, 1)),      # ... three cout tiles, one cin chunk
    (30, 30, (3, 5, 20), (3, 3, 3), (1, 1, 1)),      # ... odd H (half tile), channel tails
    (30, 60, (6, 12, 34), (1, 1, 1), (2, 2, 2)),     # strided 1x1x1 skip conv of a residual block
    (60, 120, (5, 9, 12), (1, 1, 1), (1, 2, 2)),
])
def test_conv_bwd_weight(dev, Cin, Cout, shape, k, stride):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g)
    w = (torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))).requires_grad_(True)
    pad = tuple((kk - 1) // 2 for kk in k)
    y = F.conv3d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xa = ops.Act(to_ndhwc(x).to(dev))
    ya = ops.Act(to_ndhwc(dy).to(dev))
    geom = ops.ConvGeom(shape, k, stride, pad)
    p = ops.fill_conv([xa], geom, Cout)
    ws = torch.empty(max(ops.conv3d_bwd_weight_workspace(p) // 4, 1), device=dev)
    dw = torch.full(w.shape, float('nan'), device=dev)
    ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w.grad) < 2e-5
    if k == (3, 3, 3) and stride == (1, 1, 1) and Cin > 1 and shape[2] > 16 and shape[0] >= 3:
        # these shapes take the Winograd kernel by default: the direct marching kernel must agree too
        ops.set_option('bwdw_wino', 0)
        ops.apply_selection(p)
        try:
            dw2 = torch.full(w.shape, float('nan'), device=dev)
            ws2 = torch.empty(max(ops.conv3d_bwd_weight_workspace(p) // 4, 1), device=dev)
            ops.conv3d_bwd_weight(p, ya, dw2, ops.conv_weight_strides(dw2), False, ws2)
            torch.cuda.synchronize()
        finally:
            ops.set_option('bwdw_wino', 1)
        assert relerr(dw2.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("cins,Cout,shape,lazy", [
    ((32,), 32, (5, 8, 64), True),           # the residual encoder's stage-0 block: one cout tile, two cin chunks
    ((32, 32), 32, (3, 9, 40), True),        # decoder concat 64 -> 32: two sources, odd H (half tile), ragged W
    ((30,), 30, (1, 5, 20), False),          # a single plane, channel tails
    ((24,), 70, (7, 6, 33), False),          # three cout tiles, more planes than ring slots
])
def test_conv_bwd_weight_winograd_1x3x3(dev, cins, Cout, shape, lazy):
    """conv_bwdw_wino_kernel<2, CW, KD = 1>: F(3x3, 2x2) backward-weight of the 1x3x3 stride-1 convolutions (residual encoder stage 0,
    generic_modular_residual_UNet.py:69-71) against host autograd and against the direct tiled kernel it replaces (bwdw_wino = 0)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    N, k = 2, (1, 3, 3)
    xs, acts, hosts = [], [], []
    for ci in cins:
        x = torch.randn((N, ci) + shape, generator=g)
        if lazy:
            sc, sh = torch.rand((N, ci), generator=g) + 0.5, torch.randn((N, ci), generator=g) * 0.3
            hosts.append(F.leaky_relu(x * sc[:, :, None, None, None] + sh[:, :, None, None, None], 0.01))
            acts.append(ops.Act(to_ndhwc(x).to(dev), scale=sc.to(dev), shift=sh.to(dev), slope=0.01))
        else:
            hosts.append(x)
            acts.append(ops.Act(to_ndhwc(x).to(dev)))
    Cin = sum(cins)
    w = (torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * 9)).requires_grad_(True)
    y = F.conv3d(torch.cat(hosts, 1), w, None, stride=1, padding=(0, 1, 1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ya = ops.Act(to_ndhwc(dy).to(dev))
    p = ops.fill_conv(acts, ops.ConvGeom(shape, k, (1, 1, 1), (0, 1, 1)), Cout)
    assert ops.conv_bwd_weight_kernel_name(p, ya) == 'conv_bwdw_wino_kernel<2, KD = 1>'
    res = {}
    try:
        for mode in (1, 0):
            ops.set_option('bwdw_wino', mode)
            ops.apply_selection(p)
            ws = torch.full((max(ops.conv3d_bwd_weight_workspace(p) // 4, 1),), float('nan'), device=dev)
            dw = torch.full(w.shape, float('nan'), device=dev)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), True, ws)       # accumulate: twice the gradient
            torch.cuda.synchronize()
            res[mode] = dw.cpu()
            assert relerr(res[mode], 2 * w.grad) < 2e-5, mode
    finally:
        ops.set_option('bwdw_wino', 1)
    assert ops.conv_bwd_weight_kernel_name(ops.apply_selection(p), ya) == 'conv_bwdw_wino_kernel<2, KD = 1>'
    assert relerr(res[1], res[0]) < 2e-5


@pytest.mark.parametrize("Cin,Cout,shape,k,stride,lazy", [
    (30, 60, (6, 12, 68), (3, 3, 3), (2, 2, 2), False),      # two cout tiles -> 2 per workgroup, tile 4 x 32, ragged in W
    (30, 120, (6, 12, 34), (3, 3, 3), (2, 2, 2), True),      # four -> 4 per workgroup, lazily activated X
    (20, 240, (4, 10, 20), (3, 3, 3), (2, 2, 2), False),     # eight -> 4 per workgroup, two block rows; tile 8 x 16
    (40, 320, (3, 8, 16), (3, 3, 3), (1, 2, 2), False),      # ten -> 2 per workgroup, anisotropic stride, three cin chunks
    (30, 70, (5, 9, 37), (3, 3, 3), (2, 2, 2), False),       # three cout tiles: stays at one per workgroup
    (16, 64, (3, 9, 40), (1, 3, 3), (1, 1, 1), True),        # 1x3x3 (residual encoder, stage 0)
    (30, 128, (2, 6, 70), (1, 3, 3), (1, 1, 1), False),
    (30, 60, (4, 8, 36), (2, 2, 2), (2, 2, 2), False),       # ConvTranspose3d(k = s) weight geometry
    (24, 120, (3, 8, 20), (1, 2, 2), (1, 2, 2), False),
    (24, 40, (5, 9, 37), (3, 3, 3), (1, 1, 1), False),       # Winograd backward-weight, two cout tiles per workgroup: ragged tiles, channel tails
    (30, 128, (6, 10, 70), (3, 3, 3), (1, 1, 1), True),      # ... four cout tiles = two workgroup columns, lazily activated X, two cin chunks
    (60, 64, (3, 5, 33), (3, 3, 3), (1, 1, 1), False),       # ... odd H (half tile), four cin chunks
])
def test_conv_bwd_weight_cout_tiles_per_workgroup(dev, Cin, Cout, shape, k, stride, lazy):
    """conv_bwdw_fast_kernel with 2 / 4 cout tiles per workgroup (option bwdw_cw, default 4: a wave takes one cout tile and 2 / 4 of the
    tile's four k-step blocks; the staged X tile feeds 2 / 4 times the MFMAs) and conv_bwdw_wino_kernel with 2 (a wave takes 16 output
    channels and both tile rows): against host autograd (F.conv3d, fp32) and against the one-tile-per-workgroup form of the same
    kernel, accumulate mode included."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g)
    sc = (torch.rand((N, Cin), generator=g) + 0.5) if lazy else None
    sh = torch.randn((N, Cin), generator=g) * 0.3 if lazy else None
    xa_host = F.leaky_relu(x * sc[:, :, None, None, None] + sh[:, :, None, None, None], 0.01) if lazy else x
    w = (torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))).requires_grad_(True)
    pad = tuple((kk - 1) // 2 if kk == 3 else 0 for kk in k)
    y = F.conv3d(xa_host, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xa = ops.Act(to_ndhwc(x).to(dev), scale=sc.to(dev), shift=sh.to(dev), slope=0.01) if lazy else ops.Act(to_ndhwc(x).to(dev))
    ya = ops.Act(to_ndhwc(dy).to(dev))
    geom = ops.ConvGeom(shape, k, stride, pad)
    p = ops.fill_conv([xa], geom, Cout)
    res = {}
    try:
        for cw in (4, 1):
            ops.set_option('bwdw_cw', 100 + cw if cw > 1 else 1)     # (+ 100: also on volumes this small)
            ops.apply_selection(p)
            ws = torch.full((max(ops.conv3d_bwd_weight_workspace(p) // 4, 1),), float('nan'), device=dev)
            dw = torch.full(w.shape, float('nan'), device=dev)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), True, ws)       # accumulate: twice the gradient
            torch.cuda.synchronize()
            res[cw] = dw.cpu()
            assert relerr(res[cw], 2 * w.grad) < 2e-5, cw
    finally:
        ops.set_option('bwdw_cw', 4)
    assert relerr(res[4], res[1]) < 1e-5


@pytest.mark.parametrize("N,Cin,Cout,shape,two", [(2, 64, 64, (3, 12, 12), False), (2, 48, 40, (3, 6, 6), False), (1, 32, 64, (3, 10, 7), True)])
def test_conv_tapsplit_bf16(dev, N, Cin, Cout, shape, two):
    """low-resolution layers in mixed precision: conv_tapsplit_kernel<2, true> (taps split over the waves, one bf16 MFMA per tap),
    forward with lazy input(s) + statistics against the same arithmetic on the CPU (operands rounded to bf16): 1e-4."""
    ops = _ops()
    ops.set_mma(1)
    try:
        g = torch.Generator().manual_seed(31)
        srcs = [torch.randn((N, Cin) + shape, generator=g)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
        if two:
            srcs.append(torch.randn((N, Cin) + shape, generator=g)); lazy.append(None)
        ct = Cin * len(srcs)
        w = torch.randn((Cout, ct, 3, 3, 3), generator=g) / np.sqrt(ct * 27)
        b = torch.randn(Cout, generator=g)
        acts = [ops.Act(to_ndhwc(s).to(dev)) for s in srcs]
        pq = ops.fill_conv(acts, ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1)), Cout)
        assert ops.conv_kernel_name(pq).startswith('conv_tapsplit_kernel<2, true')
        out, part = run_conv(dev, srcs, w, b, (1, 1, 1), (1, 1, 1), lazy=lazy, stats=True)
        ref = F.conv3d(_bf16_round(ref_inputs(srcs, lazy)).double(), _bf16_round(w).double(), b.double(), padding=1)
        assert relerr(to_ncdhw(out.cpu()), ref) < 1e-4
        assert np.allclose(part.cpu().double().sum(1)[..., 1].numpy(), (ref ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
    finally:
        ops.set_mma(0)


@pytest.mark.parametrize("N,Cin,Cout,shape", [(2, 30, 30, (5, 18, 70)), (1, 32, 64, (3, 8, 32))])
def test_conv_bf16_1x3x3(dev, N, Cin, Cout, shape):
    """the 1x3x3 form of conv_bf16_kernel (first stage of the residual encoder): forward with lazy input + statistics and the
    flipped-weight backward-data, against the same arithmetic on the CPU (operands rounded to bf16): 1e-4."""
    ops = _ops()
    ops.set_option('conv_bf16', 2)
    ops.set_mma(1)
    try:
        g = torch.Generator().manual_seed(29)
        srcs = [torch.randn((N, Cin) + shape, generator=g)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
        w = torch.randn((Cout, Cin, 1, 3, 3), generator=g) / np.sqrt(Cin * 9)
        b = torch.randn(Cout, generator=g)
        out, part = run_conv(dev, srcs, w, b, (1, 1, 1), (0, 1, 1), lazy=lazy, stats=True)
        xin = ref_inputs(srcs, lazy)
        ref = F.conv3d(_bf16_round(xin).double(), _bf16_round(w).double(), b.double(), padding=(0, 1, 1))
        assert relerr(to_ncdhw(out.cpu()), ref) < 1e-4
        assert np.allclose(part.cpu().double().sum(1)[..., 1].numpy(), (ref ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
        dy = torch.randn(ref.shape, generator=g)
        x = torch.zeros_like(xin, dtype=torch.float64).requires_grad_(True)
        F.conv3d(x, _bf16_round(w).double(), None, padding=(0, 1, 1)).backward(_bf16_round(dy).double())
        dx = torch.full((N,) + shape + (Cin,), float('nan'), device=dev)
        dyd = to_ndhwc(dy).to(dev)
        geomT = ops.ConvGeom(shape, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        p = ops.fill_conv([ops.Act(dyd)], geomT, Cin, out0=ops.Act(dx))
        assert ops.conv_kernel_name(p).startswith('conv_bf16')
        wd = w.to(dev).contiguous()
        wp = ops.pack_conv_weights(wd, Cout, 0, Cin, (1, 3, 3), ops.conv_weight_strides(wd, as_bwd_data=True), True, ops.conv_ck(p),
                                   layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        assert relerr(to_ncdhw(dx.cpu()), x.grad) < 1e-4
    finally:
        ops.set_option('conv_bf16', 1)
        ops.set_mma(0)


@pytest.mark.parametrize("Cin,Cout,shape,lazy", [
    (32, 32, (6, 8, 32), False),
    (24, 40, (5, 9, 37), True),       # ragged tiles in H and W, two cout tiles, lazily activated input
    (16, 70, (4, 6, 40), False),      # three cout tiles, one cin chunk
    (30, 30, (3, 5, 20), True),       # odd H (half tile), channel tails
])
def test_conv_bwd_weight_bf16_mixed_precision(dev, Cin, Cout, shape, lazy):
    """mt_conv3d_t.mma = 1 with FP32 storage on both sides (the MT_BF16_STORAGE=0 experiment mode): since round 5 these launches take the
    fp32 backward-weight kernels (the bf16 Winograd marching kernels are gone; 16-bit storage is served by conv_bwdw_tr16_kernel,
    tests/test_storage_bf16_gpu.py::test_bwdw_tr16_vs_host).  Against autograd, the accumulating form (dW += ...) on top."""
    ops = _ops()
    g = torch.Generator().manual_seed(14)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g)
    lz = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)] if lazy else None
    xin = ref_inputs([x], lz)
    w = (torch.randn((Cout, Cin, 3, 3, 3), generator=g) / np.sqrt(Cin * 27)).requires_grad_(True)
    y = F.conv3d(xin, w, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xb = to_ndhwc(x).to(dev)
    xa = ops.Act(xb, scale=lz[0][0].to(dev).contiguous(), shift=lz[0][1].to(dev).contiguous(), slope=0.01) if lazy else ops.Act(xb)
    ya = ops.Act(to_ndhwc(dy).to(dev))
    geom = ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    p = ops.fill_conv([xa], geom, Cout, mma=1)
    ws = torch.empty(max(ops.conv3d_bwd_weight_workspace(p) // 4, 1), device=dev)
    base = torch.randn(w.shape, generator=g)
    dw = base.to(dev)
    ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), True, ws)
    torch.cuda.synchronize()
    got = dw.cpu() - base
    err = relerr(got, w.grad)
    assert err < 1e-4, err


@pytest.mark.parametrize("Cin,Cout,shape", [(30, 30, (5, 9, 37)), (32, 64, (3, 8, 40)), (16, 30, (1, 6, 20))])
def test_conv_bwd_weight_bf16_1x3x3(dev, Cin, Cout, shape):
    """1x3x3 weight gradient (residual encoder stage 0) with mma = 1 and fp32 storage: the fp32 tiled kernel since round 5; vs autograd."""
    ops = _ops()
    g = torch.Generator().manual_seed(15)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g)
    w = (torch.randn((Cout, Cin, 1, 3, 3), generator=g) / np.sqrt(Cin * 9)).requires_grad_(True)
    y = F.conv3d(x, w, None, padding=(0, 1, 1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xa, ya = ops.Act(to_ndhwc(x).to(dev)), ops.Act(to_ndhwc(dy).to(dev))
    geom = ops.ConvGeom(shape, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    p = ops.fill_conv([xa], geom, Cout, mma=1)
    ws = torch.empty(max(ops.conv3d_bwd_weight_workspace(p) // 4, 1), device=dev)
    dw = torch.full(w.shape, float('nan'), device=dev)
    ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
    torch.cuda.synchronize()
    err = relerr(dw.cpu(), w.grad)
    assert err < 1e-4, err


@pytest.mark.parametrize("Cin,Cout,base,so,extra", [
    (60, 30, (4, 6, 10), (2, 2, 2), 3),
    (320, 320, (3, 6, 6), (1, 2, 2), 3),
    (30, 47, (4, 8, 16), (1, 1, 1), 3),
    (60, 30, (3, 4, 32), (2, 2, 2), 30),     # rows of 32 base voxels, 16-byte aligned concat slot: the wide (LDS-transposed) epilogue
    (20, 14, (2, 3, 64), (1, 2, 2), 14),
    (24, 32, (2, 2, 32), (2, 2, 2), 0),
    (60, 30, (3, 4, 32), (2, 2, 2), 0),      # dense output (the engine's transposed convs): one linear run per (kd, kh) pair
    (20, 14, (2, 3, 64), (1, 2, 2), 0),
])
def test_pointwise_tconv_and_heads(dev, Cin, Cout, base, so, extra):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    N = 2
    x = torch.randn((N, Cin) + base, generator=g)
    xa = ops.Act(to_ndhwc(x).to(dev))
    if so == (1, 1, 1):
        w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
        ref = F.conv3d(x, w)
        wd = w.to(dev).contiguous()
        strides = ops.conv_weight_strides(wd)
    else:
        w = torch.randn((Cin, Cout) + so, generator=g) / np.sqrt(Cin)
        ref = F.conv_transpose3d(x, w, stride=so)
        wd = w.to(dev).contiguous()
        strides = ops.conv_weight_strides(wd, transposed_layout=True)
    wp = ops.pack_conv_weights(wd, Cin, 0, Cout, so, strides, False, ops.POINTWISE_CK)
    outshape = tuple(b * s for b, s in zip(base, so))
    out = torch.full((N,) + outshape + (Cout + extra,), float('nan'), device=dev)   # write into a wider buffer (concat slot)
    oa = ops.Act(out, c0=0, C=Cout)
    p = ops.fill_pointwise(xa, base, base, (1, 1, 1), so, Cout, wp, None, oa)
    ops.pointwise_fwd(p)
    torch.cuda.synchronize()
    got = to_ncdhw(out[..., :Cout].cpu())
    assert relerr(got, ref) < 1e-5
    assert torch.isnan(out[..., Cout:]).all()


def test_instance_norm_fwd_bwd(dev):
    """conv epilogue partials -> finalize -> lazy apply, and the fused IN+LeakyReLU backward vs autograd."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    N, Cn, shape = 2, 30, (5, 9, 37)
    V = int(np.prod(shape))
    y = (torch.randn((N, Cn) + shape, generator=g) * 2 + 0.7).requires_grad_(True)
    gamma = (torch.rand(Cn, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(Cn, generator=g).requires_grad_(True)
    a = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    ga = torch.randn(a.shape, generator=g)
    a.backward(ga)
    yb = to_ndhwc(y.detach()).to(dev)
    # statistics partials the way the conv epilogue would write them: one block per sample here
    part = torch.stack([yb.sum((1, 2, 3)), (yb * yb).sum((1, 2, 3))], -1)[:, None].contiguous()  # [N,1,C,2]
    mean = torch.empty((N, Cn), device=dev); rstd = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean)
    gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
    ops.inorm_finalize(part, N, 1, Cn, V, gd, bd, 1e-5, mean, rstd, scale, shift)
    act = ops.Act(yb, scale=scale, shift=shift, slope=0.01, mean=mean, rstd=rstd)
    out = torch.empty_like(yb)
    ops.inorm_lrelu_apply(act, ops.Act(out))
    torch.cuda.synchronize()
    assert relerr(to_ncdhw(out.cpu()), a.detach()) < 2e-5
    gbuf = to_ndhwc(ga).to(dev)
    dgamma = torch.zeros(Cn, device=dev); dbeta = torch.zeros(Cn, device=dev); dbias = torch.empty(Cn, device=dev)
    ws = torch.empty(ops.inorm_bwd_workspace(N, V, Cn) // 4, device=dev)
    ops.inorm_lrelu_bwd(ops.Act(gbuf), act, gd, bd, dgamma, dbeta, dbias, ws)
    torch.cuda.synchronize()
    assert relerr(to_ncdhw(gbuf.cpu()), y.grad) < 5e-5
    assert relerr(dgamma.cpu(), gamma.grad) < 5e-5
    assert relerr(dbeta.cpu(), beta.grad) < 5e-5
    assert float(dbias.abs().max()) < 1e-3 * float(ga.abs().sum() / Cn) + 1e-3   # sum dy == 0 analytically


def test_sgd_nesterov_and_clip(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    n = 100003
    p0 = torch.randn(n, generator=g); g1 = torch.randn(n, generator=g) * 0.2; g2 = torch.randn(n, generator=g) * 0.001
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([pr], 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    nflat = (n + 3) // 4 * 4
    pd = torch.zeros(nflat, device=dev); pd[:n] = p0.to(dev)
    gd = torch.zeros(nflat, device=dev); buf = torch.zeros(nflat, device=dev)
    ss = torch.zeros(1, device=dev); ws = torch.empty(ops.sumsq_workspace(nflat) // 4, device=dev)
    for step, gg in enumerate([g1, g2]):
        pr.grad = gg.clone()
        torch.nn.utils.clip_grad_norm_([pr], 12)
        opt.step()
        gd[:n] = gg.to(dev)
        ops.sumsq(gd, ss, ws)
        ops.sgd_nesterov(pd, gd, buf, 1e-2, 3e-5, 0.99, step == 0, ss, 12.0)
        torch.cuda.synchronize()
        assert abs(float(ss.cpu()) - float((gg.double() ** 2).sum())) < 1e-4 * float((gg.double() ** 2).sum())
        assert relerr(pd[:n].cpu(), pr.detach()) < 1e-6


def test_layout_transposes(dev):
    ops = _ops()
    x = torch.randn(2, 5, 3, 7, 9)
    nd = ops.ncdhw_to_ndhwc(x.to(dev))
    assert torch.equal(nd.cpu(), to_ndhwc(x))
    back = ops.ndhwc_to_ncdhw(ops.Act(nd))
    assert torch.equal(back.cpu(), x)


def test_ds_label_pyramid_on_device(dev):
    """mt_downsample_seg_nearest == oracle restatement of DownsampleSegForDSTransform2(order 0) + RemoveLabelTransform(-1, 0),
    bit-exact (labels are copied, never interpolated)."""
    from oracle import reference_ops as R
    from multitalent_amd.training.data_augmentation.downsampling import downsample_seg_for_ds_transform2
    rng = np.random.RandomState(5)
    for shape, scales in [((48, 192, 192), [(1, 1, 1), (0.5, 0.5, 0.5), (0.25, 0.25, 0.25), (0.125, 0.125, 0.125), (0.0625, 0.0625, 0.0625)]),
                          ((48, 96, 80), [(1, 1, 1), (1, 0.5, 0.5), (0.5, 0.25, 0.25)]),
                          ((37, 45, 51), [(0.5, 0.5, 0.5), (0.25, 0.5, 0.125)])]:
        seg = rng.randint(-1, 48, size=(2, 1) + shape).astype(np.float32)
        ref = R.downsample_seg_for_ds_transform2(R.remove_label(seg), scales, 0)
        got = downsample_seg_for_ds_transform2(torch.from_numpy(seg).to(dev), scales, 0, None, remove_minus_one=True)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert tuple(g.shape) == r.shape
            assert np.array_equal(g.cpu().numpy(), r)


@pytest.mark.parametrize("N,Cin,Cout,shape,two_src", [
    (2, 32, 32, (8, 16, 64), False),
    (2, 30, 30, (9, 18, 70), False),        # ragged tiles, channel tail (chunks 8,8,8,6), Cout not a multiple of 32
    (1, 64, 40, (12, 20, 33), False),
    (1, 30, 30, (5, 7, 19), True),          # concat input: chunks never straddle the two sources
    (1, 16, 70, (4, 4, 16), False),         # exactly one tile, three cout tiles
    (2, 20, 16, (8, 8, 32), False),         # odd chunk count (8,8,4): the persistent kernel's pairs end in a phantom chunk
    (1, 24, 32, (6, 9, 20), True),          # 3 + 3 chunks: the middle pair straddles the two sources
])
@pytest.mark.parametrize("waves", [8, 803])
def test_conv_winograd(dev, N, Cin, Cout, shape, two_src, waves):
    """conv_wino8p_kernel (3D Winograd F(2x2x2,3x3x3)) forced on small shapes: forward with lazy inputs + statistics, and the
    flipped-weight backward-data form with two destinations; vs F.conv3d / autograd (tolerance 1e-5: +-1 and 1/2 transforms)."""
    ops = _ops()
    ops.set_option('conv_wino', 2)
    # 8: one worker per CU (here one tile each), 803: only 3 workers per output-channel tile (every worker walks over several tiles:
    # cross-tile pipeline, ragged last iteration)
    ops.set_option('wino_persist', {8: 1, 803: 3}[waves])
    try:
        g = torch.Generator().manual_seed(21)
        srcs = [torch.randn((N, Cin) + shape, generator=g)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
        if two_src:
            srcs.append(torch.randn((N, Cin) + shape, generator=g))
            lazy.append(None)
        Ct = Cin * len(srcs)
        w = torch.randn((Cout, Ct, 3, 3, 3), generator=g) / np.sqrt(Ct * 27)
        b = torch.randn(Cout, generator=g)
        out, part = run_conv(dev, srcs, w, b, (1, 1, 1), (1, 1, 1), lazy=lazy, stats=True)
        xin = ref_inputs(srcs, lazy)
        ref = F.conv3d(xin, w, b, padding=1)
        assert relerr(to_ncdhw(out.cpu()), ref) < 1e-5
        s = part.cpu().double().sum(1)
        assert np.allclose(s[..., 0].numpy(), ref.double().sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3 * np.sqrt(ref[0, 0].numel()))
        assert np.allclose(s[..., 1].numpy(), (ref.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
        # backward-data through the same kernel (flipped, transposed weights), accumulate into two destinations
        x = xin.clone().requires_grad_(True)
        y = F.conv3d(x, w, None, padding=1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        C0 = Ct // 2 if Ct % 4 == 0 else Ct
        base0 = torch.randn((N,) + shape + (C0,), generator=g)
        base1 = torch.randn((N,) + shape + (max(Ct - C0, 1),), generator=g)
        d0, d1 = base0.to(dev), base1.to(dev)
        geomT = ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        dyd = to_ndhwc(dy).to(dev)            # keep alive: the parameter struct only holds raw pointers
        p = ops.fill_conv([ops.Act(dyd)], geomT, Ct, out0=ops.Act(d0), out1=ops.Act(d1) if C0 < Ct else None,
                          csplit=C0, accumulate=True)
        assert ops.conv_kernel_name(p).startswith('conv_wino') == (Cout % 2 == 0)
        wd = w.to(dev).contiguous()
        wp = ops.pack_conv_weights(wd, Cout, 0, Ct, (3, 3, 3), ops.conv_weight_strides(wd, as_bwd_data=True), True, ops.conv_ck(p),
                                   layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        got = torch.cat([d0.cpu() - base0] + ([d1.cpu() - base1] if C0 < Ct else []), -1)
        assert relerr(to_ncdhw(got), x.grad) < 1e-5
    finally:
        ops.set_option('conv_wino', 1)
        ops.set_option('wino_persist', 1)


@pytest.mark.parametrize("N,Cin,Cout,shape,two_src", [
    (2, 32, 32, (8, 16, 64), False),
    (2, 30, 30, (9, 18, 70), False),        # ragged tiles, channel tail (chunks 16 + 14), Cout not a multiple of 32
    (1, 64, 40, (12, 20, 33), False),
    (1, 30, 30, (5, 7, 19), True),          # concat input: chunks never straddle the two sources
    (1, 16, 70, (4, 4, 16), False),
])
def test_conv_bf16_mixed_precision(dev, N, Cin, Cout, shape, two_src):
    """conv_bf16_kernel (mt_conv3d_t.mma = 1: bf16 matrix inputs, fp32 accumulation), forced on small shapes.  Tight check
    against the SAME arithmetic restated on the CPU (inputs rounded to bf16 after the fp32 InstanceNorm+LeakyReLU, weights
    rounded to bf16, fp32 products/sums): 1e-4; loose check against the exact fp32 convolution: 2e-2 of the largest output
    (bf16 has 8 mantissa bits).  Forward with lazy inputs + statistics, and the flipped-weight backward-data form with two
    destinations."""
    ops = _ops()
    ops.set_option('conv_bf16', 2)
    ops.set_mma(1)
    try:
        g = torch.Generator().manual_seed(23)
        srcs = [torch.randn((N, Cin) + shape, generator=g)]
        lazy = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)]
        if two_src:
            srcs.append(torch.randn((N, Cin) + shape, generator=g))
            lazy.append(None)
        Ct = Cin * len(srcs)
        w = torch.randn((Cout, Ct, 3, 3, 3), generator=g) / np.sqrt(Ct * 27)
        b = torch.randn(Cout, generator=g)
        out, part = run_conv(dev, srcs, w, b, (1, 1, 1), (1, 1, 1), lazy=lazy, stats=True)
        xin = ref_inputs(srcs, lazy)
        from oracle.reference_ops import conv3d_mixed_precision
        ref_same = conv3d_mixed_precision(xin, w, b)
        ref_exact = F.conv3d(xin, w, b, padding=1)
        got = to_ncdhw(out.cpu())
        assert relerr(got, ref_same) < 1e-4
        assert relerr(got, ref_exact) < 2e-2
        s = part.cpu().double().sum(1)
        assert np.allclose(s[..., 0].numpy(), ref_same.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3 * np.sqrt(ref_same[0, 0].numel()))
        assert np.allclose(s[..., 1].numpy(), (ref_same ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
        # backward-data through the same kernel (flipped, transposed weights), accumulate into two destinations
        dy = torch.randn(ref_exact.shape, generator=g)
        x = torch.zeros_like(xin, dtype=torch.float64).requires_grad_(True)
        F.conv3d(x, _bf16_round(w).double(), None, padding=1).backward(_bf16_round(dy).double())
        C0 = Ct // 2 if Ct % 4 == 0 else Ct
        base0 = torch.randn((N,) + shape + (C0,), generator=g)
        base1 = torch.randn((N,) + shape + (max(Ct - C0, 1),), generator=g)
        d0, d1 = base0.to(dev), base1.to(dev)
        geomT = ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        dyd = to_ndhwc(dy).to(dev)            # keep alive: the parameter struct only holds raw pointers
        p = ops.fill_conv([ops.Act(dyd)], geomT, Ct, out0=ops.Act(d0), out1=ops.Act(d1) if C0 < Ct else None,
                          csplit=C0, accumulate=True)
        assert ops.conv_kernel_name(p).startswith('conv_bf16') == (Cout % 2 == 0 and Cout >= 16)
        wd = w.to(dev).contiguous()
        wp = ops.pack_conv_weights(wd, Cout, 0, Ct, (3, 3, 3), ops.conv_weight_strides(wd, as_bwd_data=True), True, ops.conv_ck(p),
                                   layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        gotb = torch.cat([d0.cpu() - base0] + ([d1.cpu() - base1] if C0 < Ct else []), -1)
        tol = 1e-4 if ops.conv_kernel_name(p).startswith('conv_bf16') else 2e-2      # odd Cout: exact fp32 kernel on unrounded data
        assert relerr(to_ncdhw(gotb), x.grad) < tol
    finally:
        ops.set_option('conv_bf16', 1)
        ops.set_mma(0)


@pytest.mark.parametrize("N,Cin,Cout,shape,k", [
    (2, 30, 60, (6, 12, 34), (2, 2, 2)),
    (1, 64, 33, (5, 9, 21), (2, 2, 2)),          # odd sizes: the last input plane/row/column is unused
    (2, 60, 30, (4, 8, 16), (1, 2, 2)),
    (1, 17, 40, (4, 6, 10), (2, 2, 2)),          # odd channel stride: scalar loads
    (2, 30, 70, (32, 64, 128), (2, 2, 2)),       # a grid that fills the chip: two cout tiles per wave, three tiles (the last wave's second one is idle)
    (2, 30, 60, (64, 64, 130), (2, 2, 2)),       # ... two tiles, a ragged last voxel block
])
def test_conv_kernel_equals_stride_gather(dev, N, Cin, Cout, shape, k):
    """kernel == stride, pad 0 (the backward-data form of ConvTranspose3d(k = s)): conv_gather_kernel (one cout tile per wave, or two
    on well-filled fp32 grids), lazy input, accumulate."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    x = torch.randn((N, Cin) + shape, generator=g)
    sc, sh = torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g)
    w = torch.randn((Cout, Cin) + k, generator=g) / np.sqrt(Cin * np.prod(k))
    xin = ref_inputs([x], [(sc, sh, 0.01)])
    ref = F.conv3d(xin, w, None, stride=k, padding=0)
    xb = to_ndhwc(x).to(dev)
    act = ops.Act(xb, scale=sc.to(dev).contiguous(), shift=sh.to(dev).contiguous(), slope=0.01)
    geom = ops.ConvGeom(shape, k, k, (0, 0, 0))
    base = torch.randn((N,) + geom.out + (Cout,), generator=g)
    out = base.to(dev)
    p = ops.fill_conv([act], geom, Cout, out0=ops.Act(out), accumulate=True)
    assert ops.conv_kernel_name(p).startswith('conv_gather_kernel')
    wgs2 = -(-int(np.prod(geom.out)) // 128) * N * -(-(-(-Cout // 32)) // 2)          # workgroups of the two-tile form
    assert ops.conv_kernel_name(p).endswith('false, 2>') == (Cout > 32 and wgs2 >= 4 * torch.cuda.get_device_properties(dev).multi_processor_count), \
        ops.conv_kernel_name(p)
    assert ops.conv_kernel_name(p).endswith('false, 2>') == (int(np.prod(shape)) >= 32 * 64 * 128)               # (the two large cases take it)
    wd = w.to(dev).contiguous()
    wp = ops.pack_conv_weights(wd, Cin, 0, Cout, k, ops.conv_weight_strides(wd), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    assert relerr(to_ncdhw(out.cpu() - base), ref) < 1e-5


def test_extract_tiles_matches_torch_flip(dev):
    """mt_extract_tiles: the network batch of several tiles with mirror flips == torch.flip of the slices (neural_network.py:531-586)."""
    from multitalent_amd import ops
    g = torch.Generator().manual_seed(3)
    for Cn in (1, 2):
        vol = torch.randn((Cn, 21, 37, 45), generator=g).to(dev)
        patch = (8, 16, 20)
        tiles = []
        for i, (o, f) in enumerate([((0, 0, 0), (0, 0, 0)), ((13, 21, 25), (1, 0, 0)), ((5, 7, 9), (0, 1, 1)), ((13, 0, 25), (1, 1, 1)),
                                     ((1, 2, 3), (0, 0, 1)), ((2, 21, 0), (1, 1, 0))]):
            tiles.append((o, f))
        out = torch.full((len(tiles), Cn) + patch, float('nan'), device=dev)
        ops.extract_tiles(vol, patch, tiles, out)
        for k, ((x0, y0, z0), f) in enumerate(tiles):
            ref = vol[:, x0:x0 + patch[0], y0:y0 + patch[1], z0:z0 + patch[2]]
            axes = tuple(a + 1 for a in range(3) if f[a])
            if axes:
                ref = torch.flip(ref, axes)
            assert torch.equal(out[k], ref), k
    with pytest.raises(RuntimeError):
        ops.extract_tiles(vol, patch, [((14, 0, 0), (0, 0, 0))], out[:1])          # leaves the volume


@pytest.mark.parametrize("B,C,V,wide", [(3, 47, 5 * 7 * 11, False), (2, 47, 48 * 20 * 21, False), (2, 5, 1237, False), (1, 64, 4099, False),
                                         (2, 47, 3001, True)])
def test_multitalent_loss_kernels_flat_and_strided(dev, B, C, V, wide):
    """mt_multitalent_loss_fwd / _bwd: the flat coalesced kernels (contiguous logits) and the channel-strided fallback (logits are a
    channel slice of a wider buffer) against the formula of MultiTalent_Trainer_DDP.py:574-594 in torch fp64."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + C)
    cs = C + 3 if wide else C
    buf = (2.5 * torch.randn((B, V, 1, 1, cs), generator=g)).to(dev)
    a = ops.Act(buf, 0, C)
    target = torch.randint(-1, 50, (B, V), generator=g).float().to(dev)
    lut_h = torch.randint(0, 2 ** 62, (C,), generator=g, dtype=torch.int64)
    valid_h = torch.randint(0, 2 ** 62, (B,), generator=g, dtype=torch.int64) & ((1 << C) - 1 if C < 63 else -1)
    lut, valid = lut_h.to(dev), valid_h.to(dev)
    stats = torch.empty((B, C, 4), device=dev)
    ws = torch.empty(ops.loss_workspace(B, V, C) // 4 + 16, device=dev)
    ops.multitalent_loss_fwd(a, target, valid, lut, stats, ws)
    x = buf[..., :C].reshape(B, V, C).double().cpu()
    t = target.cpu().long()
    y = torch.zeros((B, V, C), dtype=torch.float64)
    for c in range(C):
        m = int(lut_h[c])
        y[:, :, c] = ((t >= 0) & (t < 64) & (((torch.tensor(m, dtype=torch.int64) >> t.clamp(0, 63)) & 1) == 1)).double()
    act = torch.tensor([[(int(valid_h[b]) >> c) & 1 for c in range(C)] for b in range(B)], dtype=torch.float64)[:, None, :]
    sg = torch.sigmoid(x)
    bce = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    ref = torch.stack(((bce * act).sum(1), (sg * y * act).sum(1), (sg * (1 - y) * act).sum(1), ((1 - sg) * y * act).sum(1)), -1)
    assert torch.allclose(stats.cpu().double(), ref, rtol=2e-5, atol=1e-3)
    gst = torch.randn((B, C, 4), generator=g)
    dbuf = torch.full((B, V, 1, 1, cs), float('nan'), device=dev)
    ops.multitalent_loss_bwd(a, target, valid, lut, gst.to(dev), ops.Act(dbuf, 0, C))
    gd = gst.double()[:, None]
    dref = act * (gd[..., 0] * (sg - y) + sg * (1 - sg) * (y * (gd[..., 1] - gd[..., 3]) + (1 - y) * gd[..., 2]))
    got = dbuf[..., :C].reshape(B, V, C).cpu().double()
    assert torch.allclose(got, dref, rtol=1e-4, atol=1e-5)
    if wide:
        assert torch.isnan(dbuf[..., C:]).all()            # the neighbouring channels of the wider buffer are untouched


@pytest.mark.parametrize("C,V,masks", [
    (47, 48 * 20 * 21, [0b111 << 5, (1 << 20) | (1 << 46)]),                 # 3 and 2 valid regions: sparse forward, wide backward
    (47, 4 * 1201, [1 << 0, 0]),                                             # one region; a sample with NO valid region
    (47, 6000, [(1 << 23) - 1, (1 << 24) - 1, (1 << 47) - 1]),               # 23 (sparse), 24 and 47 (flat form inside the same launch)
    (47, 3001, [0b1011 << 10, 0b1 << 40]),                                   # V * C not a multiple of 4: the dword backward
    (5, 1236, [0b10001, 0b00100]),
    (1, 4096, [0b1, 0b0, 0b1]),                                              # a single channel must not take the quad-unwrapping backward
    (2, 4096, [0b01, 0b11, 0b10]),
    (3, 4096, [0b101, 0b010, 0b111]),
])
def test_multitalent_loss_few_valid_regions(dev, C, V, masks):
    """round 4: mt_loss_fwd_sparse_kernel / mt_loss_bwd_wide_kernel (a sample carries the regions of its dataset only) against the formula of
    MultiTalent_Trainer_DDP.py:574-594 in torch fp64, and bit-identical results on a second run"""
    ops = _ops()
    B = len(masks)
    g = torch.Generator().manual_seed(C * 7 + V)
    buf = (2.5 * torch.randn((B, V, 1, 1, C), generator=g)).to(dev)
    a = ops.Act(buf, 0, C)
    target = torch.randint(-1, 50, (B, V), generator=g).float().to(dev)
    lut_h = torch.randint(0, 2 ** 62, (C,), generator=g, dtype=torch.int64)
    valid_h = torch.tensor(masks, dtype=torch.int64)
    lut, valid = lut_h.to(dev), valid_h.to(dev)
    ws = torch.empty(ops.loss_workspace(B, V, C) // 4 + 16, device=dev)
    runs = []
    for _ in range(2):
        stats = torch.full((B, C, 4), float('nan'), device=dev)
        ops.multitalent_loss_fwd(a, target, valid, lut, stats, ws)
        runs.append(stats.clone())
    assert torch.equal(runs[0], runs[1])
    x = buf.reshape(B, V, C).double().cpu()
    t = target.cpu().long()
    y = torch.zeros((B, V, C), dtype=torch.float64)
    for c in range(C):
        m = int(lut_h[c])
        y[:, :, c] = ((t >= 0) & (t < 64) & (((torch.tensor(m, dtype=torch.int64) >> t.clamp(0, 63)) & 1) == 1)).double()
    act = torch.tensor([[(int(valid_h[b]) >> c) & 1 for c in range(C)] for b in range(B)], dtype=torch.float64)[:, None, :]
    sg = torch.sigmoid(x)
    bce = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    ref = torch.stack(((bce * act).sum(1), (sg * y * act).sum(1), (sg * (1 - y) * act).sum(1), ((1 - sg) * y * act).sum(1)), -1)
    assert torch.allclose(runs[0].cpu().double(), ref, rtol=2e-5, atol=1e-3)
    gst = torch.randn((B, C, 4), generator=g)
    dbuf = torch.full((B, V, 1, 1, C), float('nan'), device=dev)
    ops.multitalent_loss_bwd(a, target, valid, lut, gst.to(dev), ops.Act(dbuf, 0, C))
    gd = gst.double()[:, None]
    dref = act * (gd[..., 0] * (sg - y) + sg * (1 - sg) * (y * (gd[..., 1] - gd[..., 3]) + (1 - y) * gd[..., 2]))
    got = dbuf.reshape(B, V, C).cpu().double()
    assert torch.allclose(got, dref, rtol=1e-4, atol=1e-5)
    assert (got[act.expand(B, V, C) == 0] == 0).all()          # exact zeros where the sample carries no region


@pytest.mark.parametrize("N,Cin,Cout,V,lazy,acc,bias", [(2, 30, 47, 48 * 20 * 21 + 5, True, False, True), (1, 30, 2, 1000, True, True, False),
                                                        (2, 60, 47, 777, False, True, True), (1, 64, 5, 4096, True, False, True), (3, 8, 64, 33, True, False, True),
                                                        (2, 30, 2, 48 * 20 * 21 + 5, True, False, True), (1, 32, 3, 1000, False, True, True),
                                                        (2, 30, 4, 777, True, True, False), (3, 32, 1, 4099, True, False, True),
                                                        # V % 64 != 0 and V * Cin * 4 % 16 != 0 with N > 1 — the staged forms write dX as 16-byte pieces past a
                                                        # sample's last row and rely on the bounds check; the next sample's first rows must survive (accumulate on and off)
                                                        (3, 30, 47, 2051, True, True, True), (3, 30, 47, 2051, True, False, True), (3, 30, 3, 2051, True, True, True)])
@pytest.mark.parametrize("sliced", [False, True])
def test_head_bwd_fused(dev, N, Cin, Cout, V, lazy, acc, bias, sliced):
    """mt_head_bwd: dX, dW and dbias of a 1x1x1 head in one pass vs autograd of F.conv3d on the activated input (fp64).
    sliced: x is a channel slice of a wider buffer — the wide heads then fetch their operands straight from global memory instead of
    through the staged LDS image of the dense case."""
    if sliced and (Cout <= 4 and Cin in (30, 32)):
        pytest.skip("the narrow form takes dense tensors only; its slice falls to the same wide kernel as the other cases")
    ops = _ops()
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = torch.randn((N, V, 1, 1, Cin), generator=g)
    sc = torch.rand((N, Cin), generator=g) + 0.5
    sh = torch.randn((N, Cin), generator=g)
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    dy = torch.randn((N, V, 1, 1, Cout), generator=g)
    dx0 = torch.randn((N, V, 1, 1, Cin), generator=g)
    if sliced:
        xw = torch.full((N, V, 1, 1, Cin + 6), float('nan'))
        xw[..., 2:2 + Cin] = x
        xd = xw.to(dev)
        a = ops.Act(xd, 2, Cin, scale=sc.to(dev), shift=sh.to(dev), slope=0.01) if lazy else ops.Act(xd, 2, Cin)
    else:
        xd = x.to(dev)
        a = ops.Act(xd, scale=sc.to(dev), shift=sh.to(dev), slope=0.01) if lazy else ops.Act(xd)
    wd = w.to(dev).contiguous()
    wb = ops.pack_conv_weights(wd, Cout, 0, Cin, (1, 1, 1), ops.conv_weight_strides(wd, as_bwd_data=True), False, ops.POINTWISE_CK)
    dxd = dx0.to(dev).clone()
    dw = torch.full_like(wd, float('nan'))
    db = torch.full((Cout,), float('nan'), device=dev) if bias else None
    ws = torch.empty(ops.head_bwd_workspace(N, V, Cin, Cout) // 4 + 16, device=dev)
    st = ops.conv_weight_strides(wd)
    dyd = dy.to(dev)
    done = ops.head_bwd(a, ops.Act(dyd), wb, ops.Act(dxd), acc, dw, st[0], st[1], db, False, ws)
    torch.cuda.synchronize()
    # reference
    xa = x.double()
    if lazy:
        t = xa * sc.double()[:, None, None, None, :] + sh.double()[:, None, None, None, :]
        xa = torch.where(t > 0, t, 0.01 * t)
    xa = xa.reshape(N * V, Cin)
    dyf = dy.double().reshape(N * V, Cout)
    w2 = w.double().reshape(Cout, Cin)
    ref_dx = (dyf @ w2).reshape(N, V, 1, 1, Cin) + (dx0.double() if acc else 0)
    ref_dw = (dyf.t() @ xa).reshape(Cout, Cin, 1, 1, 1)
    assert relerr(dxd.cpu().double(), ref_dx) < 1e-5
    assert relerr(dw.cpu().double(), ref_dw) < 1e-5
    assert done == (Cin % 32 != 0 or (Cout <= 4 and Cin in (30, 32)))          # the narrow form (head_bwd_narrow_kernel) always produces dbias
    if bias and done:
        assert relerr(db.cpu().double(), dyf.sum(0)) < 1e-5


def test_device_probe_and_per_device_setup(dev):
    """mt_probe_device: the device is gfx950 and raw buffer loads behave as the vector-load kernels assume (per-dword range
    check, dword-aligned 16-byte loads) — the library refuses to load otherwise."""
    from multitalent_amd import _lib
    arch = _lib.probe_device(0)
    assert arch.startswith('gfx950'), arch
    assert _lib._probed[0] == arch


@pytest.mark.parametrize("Wi,kernel", [(522, 'conv_wino8p_kernel'), (524, 'conv_fast_kernel')])
def test_winograd_plane_near_the_packed_offset_limit(dev, Wi, kernel):
    """Persistent Winograd kernel: a task's linear offset (ld*Hi + lh)*Wi + lw (ld, lh <= 5, lw <= 17) is packed into 20 bits.
    Hi = 400: Wi = 522 is the largest even width that fits ((5*400+5)*522+17 < 2^20), Wi = 524 must take a direct
    kernel (the old guard 5*Hi*Wi < 2^20 let it through and the offset spilled into the ld bits).  Both against F.conv3d."""
    ops = _ops()
    Hi, D, cin, cout = 400, 4, 16, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, cin, D, Hi, Wi), generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(cin * 27)
    b = torch.randn(cout, generator=g)
    ops.set_option('conv_wino', 2)
    try:
        xa = ops.Act(to_ndhwc(x).to(dev))
        p = ops.fill_conv([xa], ops.ConvGeom((D, Hi, Wi), (3, 3, 3), (1, 1, 1), (1, 1, 1)), cout)
        assert ops.conv_kernel_name(p).startswith(kernel), ops.conv_kernel_name(p)
        out, _ = run_conv(dev, [x], w, b, (1, 1, 1), (1, 1, 1))
        ref = F.conv3d(x, w, b, padding=1)
        assert relerr(to_ncdhw(out.cpu()), ref) < 2e-5
    finally:
        ops.set_option('conv_wino', 1)


@pytest.mark.parametrize("split,acc", [(False, False), (True, True), (False, True)])
def test_winograd_backward_data_emits_norm_backward_statistics(dev, split, acc):
    """mt_bwd_stats_t: the persistent Winograd kernel, as the last writer of g = dL/d lrelu(IN(y)), also emits per block
    A = sum dz and B = sum dz zhat (dz = g lrelu'(z), zhat = (y - mean) rstd, z = zhat gamma + beta) for one destination's channels;
    mt_inorm_lrelu_bwd consumes them (`part`) instead of running its own reduction.  Against the same quantities in torch and
    against the un-fused mt_inorm_lrelu_bwd on the same (g, y)."""
    ops = _ops()
    ops.set_option('conv_wino', 2)
    try:
        g_ = torch.Generator().manual_seed(77)
        N, Cd, shape, C0 = 2, 32, (6, 10, 36), 30            # the conv writes C0 (+ C1) gradient channels from Cd dY channels
        C1 = 16 if split else 0
        Ct = C0 + C1
        dyd = torch.randn((N,) + shape + (Cd,), generator=g_).to(dev)
        w = (torch.randn((Cd, Ct, 3, 3, 3), generator=g_) / np.sqrt(Cd * 27)).to(dev)            # forward weight [Cout = Cd, Cin = Ct]
        d0 = torch.randn((N,) + shape + (C0,), generator=g_).to(dev) if acc else torch.full((N,) + shape + (C0,), float('nan'), device=dev)
        d1 = (torch.randn((N,) + shape + (C1,), generator=g_).to(dev) if acc else torch.full((N,) + shape + (C1,), float('nan'), device=dev)) if split else None
        base0 = d0.clone() if acc else torch.zeros_like(d0)
        base1 = (d1.clone() if acc else torch.zeros_like(d1)) if split else None
        # the normalised layer whose output gradient is destination `which` (the second one when split)
        Cn, c0 = (C1, C0) if split else (C0, 0)
        y = torch.randn((N,) + shape + (Cn,), generator=g_).to(dev) * 2 + 0.5
        mean = y.mean((1, 2, 3)); rstd = 1.0 / torch.sqrt(y.var((1, 2, 3), unbiased=False) + 1e-5)
        gamma = (torch.rand(Cn, generator=g_) + 0.5).to(dev); beta = torch.randn(Cn, generator=g_).to(dev) * 0.3
        yact = ops.Act(y, scale=(gamma * rstd).contiguous(), shift=(beta - mean * gamma * rstd).contiguous(), slope=0.01, mean=mean.contiguous(), rstd=rstd.contiguous())
        geomT = ops.ConvGeom(shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        p = ops.fill_conv([ops.Act(dyd)], geomT, Ct, out0=ops.Act(d0), out1=ops.Act(d1) if split else None, csplit=C0, accumulate=acc)
        wp = ops.pack_conv_weights(w, Cd, 0, Ct, (3, 3, 3), ops.conv_weight_strides(w, as_bwd_data=True), True, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
        p.wpack = wp.data_ptr()
        assert ops.conv_kernel_name(p) == 'conv_wino8p_kernel' and ops.conv_bwd_stats_supported(p)
        part = torch.full((N, ops.conv_stats_blocks(p), Ct, 2), float('nan'), device=dev)
        p.stats_part = part.data_ptr()
        ops.set_bwd_stats(p, yact, gamma, beta, c0)
        assert ops.conv_kernel_name(p) == 'conv_wino8pb_kernel'
        ops.conv3d_fwd(p)
        torch.cuda.synchronize()
        g = (d1 if split else d0)                              # the finished gradient of the normalised layer's output
        zh = (y - mean[:, None, None, None, :]) * rstd[:, None, None, None, :]
        z = zh * gamma + beta
        dz = torch.where(z > 0, g, g * 0.01)
        A, B = dz.double().sum((1, 2, 3)), (dz * zh).double().sum((1, 2, 3))
        got = part.double().sum(1)                              # [N, Ct, 2]
        sl = slice(c0, c0 + Cn)
        scale = float(dz.abs().double().sum((1, 2, 3)).max())
        assert float((got[:, sl, 0] - A).abs().max()) < 1e-5 * scale and float((got[:, sl, 1] - B).abs().max()) < 1e-5 * scale * 3
        other = torch.ones(Ct, dtype=torch.bool); other[sl] = False
        assert (not bool(other.any())) or float(got[:, other].abs().max()) == 0.0          # channels of the other destination: zeros
        # the gradient itself is what the plain kernel writes
        q = ops.fill_conv([ops.Act(dyd)], geomT, Ct, out0=ops.Act(base0), out1=ops.Act(base1) if split else None, csplit=C0, accumulate=acc, wpack=wp)
        ops.conv3d_fwd(q)
        torch.cuda.synchronize()
        assert torch.equal(base0, d0) and (not split or torch.equal(base1, d1))
        # mt_inorm_lrelu_bwd with the fused partials == without
        outs = []
        for use in (False, True):
            gg = g.clone()
            dga, dbe, dbi = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
            ws = torch.empty(ops.inorm_bwd_workspace(N, yact.V, Cn) // 4 + 16, device=dev)
            ops.inorm_lrelu_bwd(ops.Act(gg), yact, gamma, beta, dga, dbe, dbi, ws, part=part if use else None, part_c0=c0)
            torch.cuda.synchronize()
            outs.append((gg, dga, dbe, dbi))
        for i_, (a_, b_) in enumerate(zip(outs[0], outs[1])):
            # dbias = sum dy is mathematically ZERO behind an InstanceNorm (both versions: rounding noise of a sum over V voxels)
            ref_scale = float(g.abs().sum()) / Cn * 1e-2 if i_ == 3 else max(float(a_.abs().max()), 1e-6)
            assert float((a_ - b_).abs().max()) <= 2e-5 * ref_scale, i_
    finally:
        ops.set_option('conv_wino', 1)


@pytest.mark.parametrize("Cin,Cout,shape,lazy,bias", [(30, 47, (4, 8, 16), True, True), (32, 64, (2, 8, 32), False, False),
                                                       (30, 33, (3, 8, 12), True, False), (20, 47, (4, 4, 10), False, True),
                                                       (30, 2, (5, 9, 21), True, True), (32, 3, (3, 7, 11), False, True),
                                                       (30, 4, (2, 8, 16), True, False), (30, 1, (3, 5, 7), True, True)])
def test_dense_head_kernel(dev, Cin, Cout, shape, lazy, bias):
    """pw_head_kernel: the 1x1x1 head with 33..64 output channels written as one dense [V][Cout] run (both channel tiles from one
    read of the input, transposed through LDS) vs F.conv3d; the (4,4,10) case has V % 32 != 0 and must fall back to pw_fast_kernel.
    Cout <= 4 from 30 / 32 dense channels: pw_narrow_kernel (one thread per voxel, vector ALU)."""
    ops = _ops()
    g = torch.Generator().manual_seed(15)
    N = 2
    x = torch.randn((N, Cin) + shape, generator=g)
    lz = [(torch.rand((N, Cin), generator=g) + 0.5, torch.randn((N, Cin), generator=g), 0.01)] if lazy else None
    xb = to_ndhwc(x).to(dev)
    xa = ops.Act(xb, scale=lz[0][0].to(dev).contiguous(), shift=lz[0][1].to(dev).contiguous(), slope=0.01) if lazy else ops.Act(xb)
    w = torch.randn((Cout, Cin, 1, 1, 1), generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g) if bias else None
    wd = w.to(dev).contiguous()
    wp = ops.pack_conv_weights(wd, Cin, 0, Cout, (1, 1, 1), ops.conv_weight_strides(wd), False, ops.POINTWISE_CK)
    out = torch.full((N,) + shape + (Cout,), float('nan'), device=dev)
    bd = b.to(dev) if bias else None
    p = ops.fill_pointwise(xa, shape, shape, (1, 1, 1), (1, 1, 1), Cout, wp, bd, ops.Act(out))
    ops.pointwise_fwd(p)
    torch.cuda.synchronize()
    ref = F.conv3d(ref_inputs([x], lz), w, b)
    assert relerr(to_ncdhw(out.cpu()), ref) < 1e-5
