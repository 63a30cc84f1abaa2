"""Micro-benchmark of single convolution launches (forward kernel / backward-weight kernel) for profiling."""
import argparse, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multitalent_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=30); ap.add_argument('--cout', type=int, default=30)
ap.add_argument('--shape', type=int, nargs=3, default=[48, 192, 192]); ap.add_argument('--n', type=int, default=2)
ap.add_argument('--k', type=int, nargs=3, default=[3, 3, 3]); ap.add_argument('--stride', type=int, nargs=3, default=[1, 1, 1])
ap.add_argument('--reps', type=int, default=5); ap.add_argument('--mode', default='fwd', choices=['fwd', 'bwdw', 'bwdd'])
ap.add_argument('--acc', type=int, default=-1, help='accumulate into the output (bwdd: default 1 — the skip connection already wrote dX; fwd: default 0)')
ap.add_argument('--lazy', type=int, default=1); ap.add_argument('--mma', type=int, default=0)
ap.add_argument('--ts', type=int, default=0, help='library built with -DWN_TS=1: print the per-phase cycle totals of the persistent Winograd kernel')
a = ap.parse_args()
ops.set_mma(a.mma)
dev = torch.device('cuda:0')
N, Cin, Cout = a.n, a.cin, a.cout
x = torch.randn((N,) + tuple(a.shape) + (Cin,), device=dev)
sc = torch.rand(N, Cin, device=dev) + 0.5; sh = torch.randn(N, Cin, device=dev)
xa = ops.Act(x, scale=sc, shift=sh, slope=0.01) if a.lazy else ops.Act(x)
w = torch.randn((Cout, Cin) + tuple(a.k), device=dev) * 0.05
b = torch.randn(Cout, device=dev)
geom = ops.ConvGeom(a.shape, a.k, a.stride)
out = torch.empty((N,) + geom.out + (Cout,), device=dev)
flops = 2.0 * N * geom.out[0] * geom.out[1] * geom.out[2] * Cin * Cout * a.k[0] * a.k[1] * a.k[2]
if a.mode == 'fwd':
    p = ops.fill_conv([xa], geom, Cout, bias=b, out0=ops.Act(out), accumulate=(a.acc == 1))
    ck = ops.conv_ck(p)
    wp = ops.pack_conv_weights(w, Cin, 0, Cout, a.k, ops.conv_weight_strides(w), False, ck, layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev)
    p.stats_part = part.data_ptr()
    run = lambda: ops.conv3d_fwd(p)
    import os
    if a.ts:
        import numpy as np
        ts = torch.zeros((256, 8, 16), dtype=torch.int64, device=dev)
        p.out1 = ts.data_ptr()
        run(); torch.cuda.synchronize(); ts.zero_()
        run(); torch.cuda.synchronize()
        t = ts.cpu().numpy().astype(float)
        names = ['phase1 work', 'barrier A', 'MFMA phase', 'barrier B', 'epi stage1', 'epi bar1', 'epi stage2', 'epi bar2', 'epi stats/tail']
        used = t[:, :, :9].sum(2) > 0
        print('cycles per wave over the whole kernel (mean over workgroups): transformer waves 0-3 | stager waves 4-7')
        for q, nm in enumerate(names):
            tr = t[:, 0:4, q][used[:, 0:4]].mean(); sg = t[:, 4:8, q][used[:, 4:8]].mean()
            print('  %-15s %10.0f | %10.0f' % (nm, tr, sg))
        print('  %-15s %10.0f | %10.0f' % ('total', t[:, 0:4, :10].sum(2)[used[:, 0:4]].mean(), t[:, 4:8, :10].sum(2)[used[:, 4:8]].mean()))
        print('  stagers: wait for the patch in flight at the top of phase 1: %.0f ; store_patch: %.0f ; (phase1 work row = issue_patch)' % (t[:, 4:8, 9][used[:, 4:8]].mean(), t[:, 4:8, 10][used[:, 4:8]].mean()))
        print('  stagers, slot 11 (DMA kernel: activation in place; slot 9 = fragment issue + wait, slot 10 = scale/shift + request): %.0f' % t[:, 4:8, 11][used[:, 4:8]].mean())
elif a.mode == 'bwdd':          # backward-data of the strided conv in one launch (mt_conv3d_bwd_data_strided): dY [Cout] -> dX [Cin]
    dy = torch.randn_like(out)
    dx = torch.zeros((N,) + tuple(a.shape) + (Cin,), device=dev)
    p = ops.fill_conv([ops.Act(dy)], geom, Cout, out0=ops.Act(dx), accumulate=(a.acc != 0))
    p.Cin = Cin
    assert ops.conv3d_bwd_data_strided_supported(p)
    wp = ops.pack_conv_weights(w, Cout, 0, Cin, a.k, ops.conv_weight_strides(w, as_bwd_data=True), False, 16,
                               layout=ops.conv_bwd_data_strided_pack_layout(p))
    p.wpack = wp.data_ptr()
    run = lambda: ops.conv3d_bwd_data_strided(p)
else:
    dy = torch.randn_like(out)
    p = ops.fill_conv([xa], geom, Cout)
    ws = torch.empty(ops.conv3d_bwd_weight_workspace(p) // 4 + 16, device=dev)
    dw = torch.empty_like(w)
    ya = ops.Act(dy)
    run = lambda: ops.conv3d_bwd_weight(p, ya, dw, ops.conv_weight_strides(dw), False, ws)
    if a.mma:
        p0 = ops.fill_conv([xa], geom, Cout, mma=0)
        ws0 = torch.empty(ops.conv3d_bwd_weight_workspace(p0) // 4 + 16, device=dev)
        dw0 = torch.empty_like(w)
        ops.conv3d_bwd_weight(p0, ya, dw0, ops.conv_weight_strides(dw0), False, ws0)
        run(); torch.cuda.synchronize()
        print('bf16 vs fp32 dW: max abs err %.4g, rms err %.4g, rms ref %.4g' % ((dw - dw0).abs().max().item(),
              (dw - dw0).pow(2).mean().sqrt().item(), dw0.pow(2).mean().sqrt().item()))
run(); torch.cuda.synchronize()
if a.mode == 'fwd':
    print('kernel:', ops.conv_kernel_name(p))
if a.mode == 'fwd' and a.mma:          # error against the fp32 kernel on the same inputs
    p0 = ops.fill_conv([xa], geom, Cout, bias=b, out0=ops.Act(torch.empty_like(out)), mma=0)
    o0 = torch.empty_like(out); p0.out0 = o0.data_ptr()
    wp0 = ops.pack_conv_weights(w, Cin, 0, Cout, a.k, ops.conv_weight_strides(w), False, ops.conv_ck(p0), layout=ops.conv_pack_layout(p0))
    p0.wpack = wp0.data_ptr()
    part0 = torch.zeros((N, ops.conv_stats_blocks(p0), Cout, 2), device=dev); p0.stats_part = part0.data_ptr()
    ops.conv3d_fwd(p0); torch.cuda.synchronize()
    print('bf16 vs fp32: max abs err %.4g, rms err %.4g, rms ref %.4g; stats sum rel err %.3g' % (
        (out - o0).abs().max().item(), (out - o0).pow(2).mean().sqrt().item(), o0.pow(2).mean().sqrt().item(),
        ((part.sum(1) - part0.sum(1)).abs().max() / part0.sum(1).abs().max()).item()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
kn = ops.conv_kernel_name(p) if a.mode == 'fwd' else ''
print("%s %s cin=%d cout=%d shape=%s k=%s s=%s: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (a.mode, kn, Cin, Cout, a.shape, a.k, a.stride, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573))
