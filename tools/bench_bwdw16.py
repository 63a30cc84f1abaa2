"""Mixed-precision backward-weight launches (fp16 lazy activations x bf16 gradients) of the benchmark networks' stride-1 3x3x3 / 1x3x3
layers, one by one: the direct transpose-read kernel (conv_bwdw_tr16_kernel, default) against the bf16 Winograd marching kernel it
replaces (mt_set_option bwdw_tr16 0).  Times include the ordered reduction of the partials (bwdw_reduce_kernel)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multitalent_amd import ops

LAYERS = [  # (cins, cout, shape, k)  at batch 2
    ((30,), 30, (48, 192, 192), (3, 3, 3)),
    ((30, 30), 30, (48, 192, 192), (3, 3, 3)),
    ((60,), 60, (24, 96, 96), (3, 3, 3)),
    ((60, 60), 60, (24, 96, 96), (3, 3, 3)),
    ((120,), 120, (12, 48, 48), (3, 3, 3)),
    ((120, 120), 120, (12, 48, 48), (3, 3, 3)),
    ((240,), 240, (6, 24, 24), (3, 3, 3)),
    ((30,), 30, (48, 192, 192), (1, 3, 3)),
    ((60,), 60, (48, 96, 96), (3, 3, 3)),
]
ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--only', type=int, default=-1)
ap.add_argument('--modes', type=int, nargs='+', default=[1, 0])
a = ap.parse_args()
dev = torch.device('cuda:0')
N = 2
g = torch.Generator().manual_seed(1)
for li, (cins, cout, shape, k) in enumerate(LAYERS):
    if a.only >= 0 and li != a.only:
        continue
    pad = tuple((kk - 1) // 2 for kk in k)
    geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
    srcs = []
    for ci in cins:
        x = torch.randn((N,) + shape + (ci,), device=dev).to(torch.float16)
        srcs.append(ops.Act(x, scale=torch.rand(N, ci, device=dev) + 0.5, shift=torch.randn(N, ci, device=dev), slope=0.01))
    y = ops.Act(torch.randn((N,) + shape + (cout,), device=dev).to(torch.bfloat16))
    C = sum(cins)
    flops = 2.0 * N * shape[0] * shape[1] * shape[2] * C * cout * k[0] * k[1] * k[2]
    byts = 2.0 * N * shape[0] * shape[1] * shape[2] * (C + cout)
    res = {}
    for mode in a.modes:
        ops.set_option('bwdw_tr16', mode)
        p = ops.fill_conv(srcs, geom, cout, mma=1)
        name = ops.conv_bwd_weight_kernel_name(p, y)
        dw = torch.empty((cout, C) + k, device=dev)
        ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
        run = lambda: ops.conv3d_bwd_weight(p, y, dw, ops.conv_weight_strides(dw), False, ws)
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        res[mode] = dw.clone()
        print("%-22s %s->%d %s k%s: %7.1f us  %6.0f TFLOP/s (%.2f of 2500)  %.2f TB/s algorithmic" % (
            name[:22], '+'.join(map(str, cins)), cout, 'x'.join(map(str, shape)), ''.join(map(str, k)), ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 2500, byts / ms / 1e9))
    if len(res) == 2:
        d = (res[0] - res[1]).abs().max().item() / res[0].abs().max().item()
        print("   max |tr16 - winograd| / max|dW| = %.2e" % d)
ops.set_option('bwdw_tr16', 1)
