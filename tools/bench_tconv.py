"""One transposed-convolution launch (pw_fast_kernel<taps>): python tools/bench_tconv.py --cin 60 --cout 30 --base 24 96 96 --k 2 2 2"""
import argparse, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multitalent_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=60); ap.add_argument('--cout', type=int, default=30)
ap.add_argument('--base', type=int, nargs=3, default=[24, 96, 96]); ap.add_argument('--k', type=int, nargs=3, default=[2, 2, 2])
ap.add_argument('--n', type=int, default=2); ap.add_argument('--ocs', type=int, default=0); ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
N, Cin, Cout = a.n, a.cin, a.cout
base, k = tuple(a.base), tuple(a.k)
osp = tuple(b * s for b, s in zip(base, k))
ocs = a.ocs or 2 * Cout                      # written into the first Cout channels of the concat buffer
x = torch.randn((N,) + base + (Cin,), device=dev)
sc = torch.rand(N, Cin, device=dev) + 0.5; sh = torch.randn(N, Cin, device=dev)
xa = ops.Act(x, scale=sc, shift=sh, slope=0.01)
w = torch.randn((Cin, Cout) + k, device=dev) * 0.05
st = ops.conv_weight_strides(w, transposed_layout=True)
wp = ops.pack_conv_weights(w, Cin, 0, Cout, k, st, False, ops.POINTWISE_CK)
out = torch.empty((N,) + osp + (ocs,), device=dev)
p = ops.fill_pointwise(xa, base, base, (1, 1, 1), k, Cout, wp, None, ops.Act(out, 0, Cout))
run = lambda: ops.pointwise_fwd(p)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
taps = k[0] * k[1] * k[2]
V = N * base[0] * base[1] * base[2]
fl = 2.0 * V * Cin * Cout * taps; nb = 4.0 * V * (Cin + Cout * taps)
print('tconv %d->%d base %s k %s: %.1f us  %.1f TF/s  %.2f TB/s  (lib %s)' % (Cin, Cout, base, k, ms * 1e3, fl / ms / 1e9, nb / ms / 1e9, os.environ.get('MT_LIB_VARIANT', 'default')))
