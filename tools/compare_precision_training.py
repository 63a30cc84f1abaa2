"""Loss trajectories of the benchmark network (Task009 config, B=2, 48x192x192, synthetic batch) in fp32 and bf16 mode from the
same initial weights: evidence that the mixed-precision path trains like the exact one."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from multitalent_amd.training.hot_loop import FusedTrainStep
dev = torch.device('cuda:0')
out = {}
for prec in ('fp32', 'bf16'):
    torch.manual_seed(1234)
    net = bench.build_network('task009').to(dev); net.train()
    net.engine().set_precision(prec)
    step = FusedTrainStep(net, bench.make_loss('task009', False), lr=1e-2)
    x, largs = bench.make_batch('task009', 2, dev, 0)
    out[prec] = [float(step(x, *largs)) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60)]
f, b = out['fp32'], out['bf16']
for i in list(range(0, len(f), 10)) + [len(f) - 1]:
    print('step %3d  fp32 %.5f  bf16 %.5f  diff %+.5f' % (i, f[i], b[i], b[i] - f[i]))
print('max |diff| over %d steps: %.5f' % (len(f), max(abs(a - c) for a, c in zip(f, b))))
