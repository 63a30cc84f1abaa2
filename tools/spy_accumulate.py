import sys, os, collections
sys.path.insert(0, '/root/repo')
import torch, bench
from multitalent_amd import ops
wl = sys.argv[1]
dev = torch.device('cuda:0')
net = bench.build_network(wl).to(dev); net.train()
from multitalent_amd.training.hot_loop import FusedTrainStep
step = FusedTrainStep(net, bench.make_loss(wl, False), lr=1e-2)
x, largs = bench.make_batch(wl, 2, dev, 0)
rec = collections.OrderedDict()
orig = ops.conv3d_fwd
def spy(p):
    k = (ops.conv_kernel_name(p).split('<')[0], p.Cin, p.Cout, (p.Do, p.Ho, p.Wo), int(p.accumulate), int(p.csplit < p.Cout))
    rec[k] = rec.get(k, 0) + 1
    orig(p)
ops.conv3d_fwd = spy
step(x, *largs); torch.cuda.synchronize()
for k, n in rec.items():
    if k[4]: print(n, k)
print('total launches', sum(rec.values()), 'accumulating', sum(n for k, n in rec.items() if k[4]))
