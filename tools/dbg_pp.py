import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from multitalent_amd import ops
dev = torch.device('cuda:0')
ops.set_option('conv_tapsplit', 0)
N, Cin, Cout, shape = 2, int(sys.argv[1]), int(sys.argv[2]), tuple(int(v) for v in sys.argv[3:6])
x = torch.randn((N,) + shape + (Cin,), device=dev)
xa = ops.Act(x, scale=torch.rand(N, Cin, device=dev) + 0.5, shift=torch.randn(N, Cin, device=dev), slope=0.01)
w = torch.randn((Cout, Cin, 3, 3, 3), device=dev) * 0.05
geom = ops.ConvGeom(shape, (3, 3, 3), (2, 2, 2))
out = torch.empty((N,) + geom.out + (Cout,), device=dev)
p = ops.fill_conv([xa], geom, Cout, bias=torch.randn(Cout, device=dev), out0=ops.Act(out))
wp = ops.pack_conv_weights(w, Cin, 0, Cout, (3, 3, 3), ops.conv_weight_strides(w), False, ops.conv_ck(p), layout=ops.conv_pack_layout(p))
p.wpack = wp.data_ptr()
part = torch.zeros((N, ops.conv_stats_blocks(p), Cout, 2), device=dev); p.stats_part = part.data_ptr()
ts = torch.zeros((256, 4, 8), dtype=torch.int64, device=dev); p.out1 = ts.data_ptr()
print(ops.conv_kernel_name(p))
for _ in range(3): ops.conv3d_fwd(p)
torch.cuda.synchronize(); ts.zero_(); ops.conv3d_fwd(p); torch.cuda.synchronize()
t = ts.cpu().numpy().astype(float)
names = ['tap loops (+ loop top)', 'chunk barrier', 'epilogue + stats']
tot = t.sum(2).mean()
for k, nm in enumerate(names):
    print('  %-38s %9.0f ticks (%4.1f %%)   waves 0-3: %s' % (nm, t[:, :, k].mean(), 100 * t[:, :, k].mean() / tot, ' '.join('%7.0f' % t[:, w, k].mean() for w in range(4))))
print('  total %.0f ticks of 10 ns = %.1f us' % (tot, tot / 100))
