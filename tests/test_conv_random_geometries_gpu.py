"""Randomised sweep over convolution geometries through the C ABI (forward, backward-data of stride-1 layers, backward-weight):
kernel sizes in {1,3}^3, strides in {1,2}^3 as the plans produce them, ragged spatial sizes, channel counts with tails, one or
two (lazily activated) sources, fp32 and bf16 matrix inputs — whatever kernel the dispatcher picks must agree with F.conv3d /
autograd on the CPU.  Seeded (reproducible); 48 cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import _bf16_round, _ops, ref_inputs, relerr, run_conv, to_ncdhw, to_ndhwc

pytestmark = pytest.mark.gpu

KERNELS = [(3, 3, 3), (1, 3, 3), (1, 1, 1), (3, 3, 3), (3, 3, 3)]
STRIDES = [(1, 1, 1), (1, 1, 1), (2, 2, 2), (1, 2, 2)]


def _case(seed):
    rs = np.random.RandomState(seed)
    k = KERNELS[rs.randint(len(KERNELS))]
    s = STRIDES[rs.randint(len(STRIDES))]
    cin = int(rs.choice([1, 8, 17, 30, 32, 48, 60]))
    cout = int(rs.choice([2, 16, 30, 33, 47, 64]))
    shape = (int(rs.randint(2, 9)), int(rs.randint(4, 21)), int(rs.randint(5, 41)))
    two = bool(rs.randint(2)) and cin > 1 and s == (1, 1, 1)
    lazy = bool(rs.randint(2))
    mma = int(rs.randint(2))
    return k, s, cin, cout, shape, two, lazy, mma


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_conv_geometry(dev, seed):
    ops = _ops()
    k, s, cin, cout, shape, two, lazy, mma = _case(seed)
    pad = tuple((kk - 1) // 2 for kk in k)
    g = torch.Generator().manual_seed(1000 + seed)
    N = 2
    srcs = [torch.randn((N, cin) + shape, generator=g)]
    lz = [(torch.rand((N, cin), generator=g) + 0.5, torch.randn((N, cin), generator=g), 0.01) if lazy else None]
    if two:
        srcs.append(torch.randn((N, cin) + shape, generator=g)); lz.append(None)
    ct = cin * len(srcs)
    w = torch.randn((cout, ct) + k, generator=g) / np.sqrt(ct * np.prod(k))
    b = torch.randn(cout, generator=g)
    ops.set_mma(mma)
    ops.set_option('conv_bf16', 2 if mma else 1)
    try:
        out, part = run_conv(dev, srcs, w, b, s, pad, lazy=lz if (lazy or two) else None, stats=True)
        xin = ref_inputs(srcs, lz if (lazy or two) else None)
        ref = F.conv3d(xin, w, b, stride=s, padding=pad)
        tol = 3e-2 if mma else 2e-5
        assert relerr(to_ncdhw(out.cpu()), ref) < tol, (k, s, cin, cout, shape, two, lazy, mma)
        st = part.cpu().double().sum(1)
        got_out = to_ncdhw(out.cpu()).double()
        assert np.allclose(st[..., 0].numpy(), got_out.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3 * np.sqrt(ref[0, 0].numel()))
        assert np.allclose(st[..., 1].numpy(), (got_out ** 2).sum((2, 3, 4)).numpy(), rtol=1e-4)
        # backward-weight of the same layer (single source)
        if not two:
            x = xin.clone()
            wr = w.clone().requires_grad_(True)
            y = F.conv3d(x, wr, None, stride=s, padding=pad)
            dy = torch.randn(y.shape, generator=g)
            y.backward(dy)
            xb = to_ndhwc(srcs[0]).to(dev)
            xa = ops.Act(xb, scale=lz[0][0].to(dev).contiguous(), shift=lz[0][1].to(dev).contiguous(), slope=0.01) if lazy else ops.Act(xb)
            ya = ops.Act(to_ndhwc(dy).to(dev))
            p = ops.fill_conv([xa], ops.ConvGeom(shape, k, s, pad), cout)
            ws = torch.empty(max(ops.conv3d_bwd_weight_workspace(p) // 4, 1), device=dev)
            dw = torch.full(w.shape, float('nan'), device=dev)
            ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
            torch.cuda.synchronize()
            assert relerr(dw.cpu(), wr.grad) < (2e-2 if mma else 5e-5), (k, s, cin, cout, shape, lazy, mma)
    finally:
        ops.set_mma(0)
        ops.set_option('conv_bf16', 1)
