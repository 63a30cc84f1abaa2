"""InstanceNorm + LeakyReLU backward of the low-resolution tensors, one launch (inorm_bwd_small_kernel) against three
(inorm_bwd_fast_kernel x 2 + inorm_bwd_finalize_kernel, the form large tensors take): python tools/bench_norm_small.py [--mixed 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multitalent_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--mixed', type=int, default=1)
ap.add_argument('--reps', type=int, default=50)
a = ap.parse_args()
dev = torch.device('cuda:0')
N = 2
for shape, C in [((12, 24, 24), 256), ((6, 12, 12), 320), ((3, 6, 6), 320), ((6, 24, 24), 240), ((3, 12, 12), 320)]:
    V = shape[0] * shape[1] * shape[2]
    yt = torch.randn((N,) + shape + (C,), device=dev)
    gt = torch.randn((N,) + shape + (C,), device=dev)
    if a.mixed:
        yt, gt = yt.to(torch.float16), gt.to(torch.bfloat16)
    mean, rstd = torch.randn(N, C, device=dev) * 0.1, torch.rand(N, C, device=dev) + 0.5
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    y = ops.Act(yt, scale=rstd * gamma, shift=beta - mean * rstd * gamma, slope=0.01)
    y.mean, y.rstd = mean, rstd
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.empty(ops.inorm_bwd_workspace(N, V, C) // 4 + 16, device=dev)
    g = ops.Act(gt.clone())
    run = lambda: ops.inorm_lrelu_bwd(g, y, gamma, beta, dg, db, None, ws)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    nbytes = N * V * C * (2 if a.mixed else 4) * 3
    print("N=2 %s C=%d %s: %6.1f us  (%.2f TB/s of g read, y read, dy written once)" % ('x'.join(map(str, shape)), C, 'mixed' if a.mixed else 'fp32', us, nbytes / us / 1e6))
