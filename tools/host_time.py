"""How long does the HOST need to enqueue one training step (no synchronisation inside the step), against the device's step time?
usage: python tools/host_time.py [task009|task100|resenc] [fp32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from multitalent_amd.training.hot_loop import FusedTrainStep

wl = sys.argv[1] if len(sys.argv) > 1 else 'task009'
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
dev = torch.device('cuda:0')
net = bench.build_network(wl); net.train()
net.engine().set_precision(prec)
step = FusedTrainStep(net, bench.make_loss(wl, False), lr=1e-2)
B = {'task009': 2, 'task100': 4, 'resenc': 2}[wl]
x, largs = bench.make_batch(wl, B, dev, 0, bench.PATCH)
for _ in range(3):
    step(x, *largs)
torch.cuda.synchronize()
n = 20
host = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    step(x, *largs)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print('%s %s: device step %.2f ms | host enqueue per step: median %.2f ms, min %.2f, max %.2f | all %d steps enqueued after %.1f ms of %.1f ms'
      % (wl, prec, 1e3 * t_all / n, 1e3 * host[n // 2], 1e3 * host[0], 1e3 * host[-1], n, 1e3 * t_enq, 1e3 * t_all))
