"""Which device kernel serves each conv launch of the benchmark network (forward / backward-data), per precision mode."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from multitalent_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda:0')
net = bench.build_network('task009').to(dev); net.train()
net.engine().set_precision(prec)
from multitalent_amd.training.hot_loop import FusedTrainStep
step = FusedTrainStep(net, bench.make_loss('task009', False), lr=1e-2)
x, largs = bench.make_batch('task009', 2, dev, 0)
rec = collections.OrderedDict()
orig = ops.conv3d_fwd
def spy(p):
    k = (ops.conv_kernel_name(p), p.Cin, p.Cout, (p.Do, p.Ho, p.Wo), (p.KD, p.KH, p.KW), (p.SD, p.SH, p.SW), (p.dilD,))
    rec[k] = rec.get(k, 0) + 1
    orig(p)
ops.conv3d_fwd = spy
step(x, *largs); torch.cuda.synchronize()
for k, n in rec.items():
    print(n, k)
