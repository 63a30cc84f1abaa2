"""Every convolution launch of one training step with its own duration (HIP events, streams serialised): kernel, algorithmic
GFLOP / MB, microseconds, TFLOP/s and TB/s — to find the launches (not the kernels) that sit far below their family's rate.
usage: python tools/per_launch.py [task009|task100|resenc] [fp32|bf16]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

workload = sys.argv[1] if len(sys.argv) > 1 else 'task009'
precision = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
dev = torch.device('cuda', 0)
r = bench.time_training(workload, precision, tuple(bench.PATCH), bench.DEFAULT_BATCH[workload], 2, 2, dev, 0, 1, False)
eng = r['step'].eng
eng.bwdw_streams = 0; eng._packed_version = None; eng._pack_programs = {}
from multitalent_amd import ops as _ops
shapes = []
_names = (_ops.conv_kernel_name, _ops.conv_bwd_weight_kernel_name, _ops.conv_bwd_data_strided_kernel_name)
def _wrap(f):
    def g(p, *a):
        shapes.append('%dx%dx%d %d->%d acc%d' % (p.Do, p.Ho, p.Wo, p.Cin, p.Cout, p.accumulate))
        return f(p, *a)
    return g
_ops.conv_kernel_name, _ops.conv_bwd_weight_kernel_name, _ops.conv_bwd_data_strided_kernel_name = [_wrap(f) for f in _names]
pw_rows = []
_pw = _ops.pointwise_fwd
def _pw_timed(p):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _pw(p); e1.record()
    taps = p.soD * p.soH * p.soW
    vb = p.N * p.Db * p.Hb * p.Wb
    pw_rows.append(('pointwise base %dx%dx%d si %d%d%d so %d%d%d %d->%d acc%d' % (p.Db, p.Hb, p.Wb, p.siD, p.siH, p.siW, p.soD, p.soH, p.soW, p.Cin, p.Cout, p.accumulate),
                    2.0 * vb * p.Cin * p.Cout * taps, 4.0 * vb * (p.Cin + p.Cout * taps * (2 if p.accumulate else 1)), e0, e1))
_ops.pointwise_fwd = _pw_timed
with bench.ConvTimer() as t:
    r['step'](r['x'], *r['largs'])
    torch.cuda.synchronize()
    rows = []
    seq = []
    for name, recs in t.rec.items():
        for (e0, e1, fl, nb) in recs:
            seq.append((e0, name, fl, nb, e0.elapsed_time(e1)))
    # launch order = order of the name queries (one per launch)
    first = min(seq, key=lambda q: 0)[0]
    seq.sort(key=lambda q: first.elapsed_time(q[0]))
    for (q, sh) in zip(seq, shapes):
        rows.append((q[1] + ' ' + sh, q[2], q[3], q[4]))
for (nm, fl, nb, e0, e1) in pw_rows:
    rows.append((nm, fl, nb, e0.elapsed_time(e1)))
rows.sort(key=lambda q: -q[3])
tot = sum(q[3] for q in rows)
print('%d conv launches, %.2f ms' % (len(rows), tot))
for name, fl, nb, ms in rows[:70]:
    print('%-84s %8.2f GF %8.1f MB %8.1f us %7.1f TF/s %6.2f TB/s' % (name[:84], fl / 1e9, nb / 1e6, ms * 1e3, fl / ms / 1e9, nb / ms / 1e9))
