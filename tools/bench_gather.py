"""conv_gather_kernel (kernel == stride, pad 0: the backward-data of ConvTranspose3d(k = s)) on the transposed-conv levels of the benchmark
networks, against F.conv3d: python tools/bench_gather.py [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, torch.nn.functional as F
from multitalent_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=20); ap.add_argument('--check', type=int, default=1)
a = ap.parse_args()
dev = torch.device('cuda:0'); N = 2
for Cin, Cout, shape, k in [(30, 60, (48, 192, 192), (2, 2, 2)), (60, 120, (24, 96, 96), (2, 2, 2)), (120, 240, (12, 48, 48), (2, 2, 2)),
                            (240, 320, (6, 24, 24), (2, 2, 2)), (320, 320, (3, 12, 12), (1, 2, 2)), (30, 70, (8, 24, 40), (2, 2, 2))]:
    x = torch.randn((N,) + shape + (Cin,), device=dev)
    w = torch.randn((Cout, Cin) + k, device=dev) / np.sqrt(Cin * np.prod(k))
    b = torch.randn(Cout, device=dev)
    geom = ops.ConvGeom(shape, k, k, (0, 0, 0))
    out = torch.full((N,) + geom.out + (Cout,), float('nan'), device=dev)
    p = ops.fill_conv([ops.Act(x)], geom, Cout, bias=b, out0=ops.Act(out))
    wp = ops.pack_conv_weights(w, Cin, 0, Cout, k, ops.conv_weight_strides(w), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
    p.wpack = wp.data_ptr()
    name = ops.conv_kernel_name(p)
    for _ in range(3): ops.conv3d_fwd(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): ops.conv3d_fwd(p)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    flops = 2.0 * N * np.prod(geom.out) * Cin * Cout * np.prod(k); nbytes = 4.0 * N * (np.prod(shape) * Cin + np.prod(geom.out) * Cout)
    err = ''
    if a.check:
        ref = F.conv3d(x.permute(0, 4, 1, 2, 3), w, b, stride=k).permute(0, 2, 3, 4, 1)
        err = '  max|d|/max|ref| %.2e' % (float((out - ref).abs().max()) / float(ref.abs().max()))
    print('%-44s %3d->%3d %-12s k%s: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s%s' % (name, Cin, Cout, 'x'.join(map(str, shape)), ''.join(map(str, k)), us, flops / us / 1e6, nbytes / us / 1e6, err))
