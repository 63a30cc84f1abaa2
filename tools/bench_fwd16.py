"""Mixed-precision 3x3x3 / 1x3x3 stride-1 convolution launches (16-bit storage on all operands) of the benchmark networks, one by one:
conv_x16_kernel (default; mt_set_option conv_x16) against conv_bf16_kernel (conv_x16 = 0).  fp16 = forward over
activations (lazy sources), bf16 = backward-data over gradients (plain source)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from multitalent_amd import ops

LAYERS = [  # (cins, cout, shape, k)
    ((30,), 30, (48, 192, 192), (3, 3, 3)),
    ((30, 30), 30, (48, 192, 192), (3, 3, 3)),
    ((60,), 60, (24, 96, 96), (3, 3, 3)),
    ((30,), 30, (48, 192, 192), (1, 3, 3)),
    ((60,), 60, (48, 96, 96), (3, 3, 3)),
    ((32,), 64, (48, 96, 96), (3, 3, 3)),
    ((120,), 120, (24, 48, 48), (3, 3, 3)),
    ((60, 60), 60, (48, 96, 96), (3, 3, 3)),
    ((240,), 240, (12, 24, 24), (3, 3, 3)),
    ((32,), 32, (48, 192, 192), (3, 3, 3)),
    ((64,), 64, (48, 96, 96), (3, 3, 3)),
]
ap_acc = None
ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--only', type=int, default=-1)
ap.add_argument('--modes', type=int, nargs='+', default=[1, 0])
ap.add_argument('--dtypes', nargs='+', default=['fp16', 'bf16'])
ap.add_argument('--ts', type=int, default=0, help='library built with -DX16_TS=1: per-phase cycle totals of conv_x16_kernel')
ap.add_argument('--acc', type=int, default=0, help='bf16 (backward-data) launches accumulate into the destination')
a = ap.parse_args()
dev = torch.device('cuda:0')
N = 2
for li, (cins, cout, shape, k) in enumerate(LAYERS):
    if a.only >= 0 and li != a.only:
        continue
    for dtn in a.dtypes:
        H = torch.float16 if dtn == 'fp16' else torch.bfloat16
        pad = tuple((kk - 1) // 2 for kk in k)
        geom = ops.ConvGeom(shape, k, (1, 1, 1), pad)
        srcs = []
        for ci in cins:
            x = torch.randn((N,) + shape + (ci,), device=dev).to(H)
            srcs.append(ops.Act(x, scale=torch.rand(N, ci, device=dev) + 0.5, shift=torch.randn(N, ci, device=dev), slope=0.01) if dtn == 'fp16' else ops.Act(x))
        C = sum(cins)
        w = (torch.randn((cout, C) + k, device=dev) / np.sqrt(C * np.prod(k))).contiguous()
        b = torch.randn(cout, device=dev)
        flops = 2.0 * N * shape[0] * shape[1] * shape[2] * C * cout * k[0] * k[1] * k[2]
        byts = 2.0 * N * shape[0] * shape[1] * shape[2] * (C + cout)
        res = {}
        for mode in a.modes:
            ops.set_option('conv_x16', 4096 if mode == 1 else 0)   # 4096: conv_x16_kernel wherever eligible (1 = the dispatcher's rule)
            out = torch.zeros((N,) + tuple(geom.out) + (cout,), device=dev, dtype=H)
            p = ops.fill_conv(srcs, geom, cout, out0=ops.Act(out), bias=b, mma=1, accumulate=bool(a.acc and dtn == 'bf16'))
            name = ops.conv_kernel_name(p)
            lay = ops.conv_pack_layout(p)
            wp = ops.pack_conv_weights(w, srcs[0].C, srcs[1].C if len(srcs) > 1 else 0, cout, k, ops.conv_weight_strides(w), False, ops.conv_ck(p), layout=lay)
            p.wpack = wp.data_ptr()
            part = torch.zeros((N, ops.conv_stats_blocks(p), cout, 2), device=dev)
            p.stats_part = part.data_ptr()
            run = lambda: ops.conv3d_fwd(p)
            run(); run(); torch.cuda.synchronize()
            if a.ts and mode:
                ts = torch.zeros((1024, 4, 8), dtype=torch.int64, device=dev)
                p.out1 = ts.data_ptr()
                run(); torch.cuda.synchronize()
                p.out1 = None
                t = ts.cpu().numpy().astype(float)
                used = t.sum(2) > 0
                names = ['convert', 'wait weights + barrier', 'issue patch loads', 'MFMA phase', 'epilogue', 'trailing barrier', 'weight DMA issue + stats', 'loop top']
                tot = t.sum(2)[used].mean()
                print('   cycles per wave over the whole kernel (mean over %d waves); 100 MHz s_memtime ticks' % used.sum())
                for kk, nm in enumerate(names):
                    print('     %-26s %9.0f  (%4.1f %%)' % (nm, t[:, :, kk][used].mean(), 100 * t[:, :, kk][used].mean() / tot))
                print('     %-26s %9.0f = %.1f us' % ('total', tot, tot / 100.0))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            res[mode] = (out.float(), part.sum(1))
            print("%-34s %s %s->%d %s k%s: %7.1f us  %6.0f TFLOP/s (%.2f of 2500)  %.2f TB/s algorithmic" % (
                name[:34], dtn, '+'.join(map(str, cins)), cout, 'x'.join(map(str, shape)), ''.join(map(str, k)), ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 2500, byts / ms / 1e9))
        for m_ in [m for m in res if m != 0 and 0 in res]:
            d = (res[0][0] - res[m_][0]).abs().max().item() / res[0][0].abs().max().item()
            ds = ((res[0][1] - res[m_][1]).abs().max() / res[0][1].abs().max()).item()
            print("   max |x16 - conv_bf16| / max|y| = %.2e, statistics %.2e" % (d, ds))
ops.set_option('conv_x16', 1)
