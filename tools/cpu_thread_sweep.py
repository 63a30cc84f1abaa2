"""Thread sweep of bench.py's CPU baseline iteration (the oracle's fwd + loss + bwd + clip + SGD at B = 2, 48x192x192, fp32) on the
GPU box's host: 1 warm-up + 2 timed iterations per thread count.  Output: gpurun_out/cpu_thread_sweep.json (copied to
profiles/r02_cpu_thread_sweep.json, which bench.py:cpu_threads() reads)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

phys, logical = bench.host_cores()
counts = sorted({c for c in (16, 32, 64, phys // 2, phys, logical) if 0 < c <= logical})
workload = sys.argv[1] if len(sys.argv) > 1 else 'task009'
res = []
for n in counts:
    torch.set_num_threads(n)
    it = bench.cpu_iteration_fn(workload, 2)
    it()
    t0 = time.time()
    for _ in range(2):
        it()
    dt = (time.time() - t0) / 2
    res.append({'threads': n, 's_per_iteration': round(dt, 3), 'patches_per_s': round(2 / dt, 4)})
    print(res[-1], flush=True)
best = max(res, key=lambda r: r['patches_per_s'])
out = {'workload': workload, 'batch': 2, 'patch': list(bench.PATCH), 'physical_cores': phys, 'logical_cpus': logical,
       'best_threads': best['threads'], 'sweep': res,
       'cpu_model': next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'cpu_thread_sweep.json'), 'w'), indent=1)
print(json.dumps(out))
