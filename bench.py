#!/usr/bin/env python
"""Benchmark of the hot path: CT patches/s (48x192x192) for one training iteration
(forward + loss + backward + clip_grad_norm_(12) + SGD-Nesterov step) on synthetic device-resident batches,
dummyLoad style (reference: nnUNet_variants/benchmarking/nnUNetTrainerV2_dummyLoad.py:26-64).

  python bench.py --gpus N --steps K --warmup W
Workload (every N): BASELINE.json configs[1] = Task009_Spleen Generic_UNet, bs=2 per GPU, patch 48x192x192, fp32, softmax
Dice+CE with deep supervision.  N > 1: one rank per GPU over RCCL (weak scaling; gradients all-reduced (mean) overlapped with
backward on a side stream) — launched by torch.distributed.run; when called as plain `python bench.py --gpus N` the script
re-executes itself under torch.distributed.run with N ranks.  --workload task100 | resenc | infer select BASELINE configs[2..4],
--patch 96 192 192 the plans' native patch, --precision bf16 the mixed-precision mode of configs[3].
Prints ONE JSON line on rank 0 (N = 1: with `roofline` of the dominant kernel and `cpu_baseline`).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

PATCH = (48, 192, 192)
POOLS = [[2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]]
KERNELS = [[3, 3, 3]] * 6
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: Peak BF16 MFMA, dense
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


RESENC_POOLS = [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]     # SURVEY §8a N6 (resenc_bs4 plan)
RESENC_KERNELS = [[1, 3, 3]] + [[3, 3, 3]] * 5
RESENC_BLOCKS = [1, 2, 3, 4, 4, 4]

# multiplications the kernel EXECUTES on the matrix cores per algorithmic multiplication of the direct convolution
#   conv_wino*: F(2x2x2, 3x3x3), 64 products per 2x2x2 outputs instead of 216;  conv_bwdw_wino*: F(3x3, 2x2) over (h, w), 16 products
#   per 3x3 taps x 2x2 outputs instead of 36, kd direct
EXECUTED_FRACTION = (('conv_wino', 64.0 / 216.0), ('conv_bwdw_wino', 16.0 / 36.0))


def executed_fraction(kernel):
    for prefix, f in EXECUTED_FRACTION:
        if kernel.startswith(prefix):
            return f
    return 1.0


def build_network(workload):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.network_architecture.initialization import InitWeights_He
    if workload == 'resenc':        # BASELINE configs[3] architecture (FabiansUNet, MultiTalent_meets_resenc.py:72-104)
        from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
        return FabiansUNet(1, 30, RESENC_BLOCKS, 2, RESENC_POOLS, RESENC_KERNELS, get_default_network_config(3, None, norm_type="in"),
                           47, [1] * (len(RESENC_POOLS) - 1), True, False, 320, InitWeights_He(1e-2))
    nc = 2 if workload == 'task009' else 47
    return Generic_UNet(1, 30, nc, len(POOLS), 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                        nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                        True, False, lambda x: x, InitWeights_He(1e-2), POOLS, KERNELS, False, True, True)


def make_batch(workload, B, dev, rank, patch=PATCH):
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
    scales = ds_scales(RESENC_POOLS, skip_first=True) if workload == 'resenc' else ds_scales(POOLS)
    x = synthetic_ct(B, patch, 1234 + rank, dev)
    if workload == 'task009':
        t = synthetic_targets(B, patch, scales, [[1]] * B, 1234 + rank, dev)
        return x, (t,)
    names = list(MultiTalent_valid_regions.keys())
    valid = [MultiTalent_valid_regions[names[(rank * B + b) % len(names)]] for b in range(B)]
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    t = synthetic_targets(B, patch, scales, label_sets, 1234 + rank, dev)
    return x, (t, valid)


def make_loss(workload, ddp):
    from multitalent_amd.training.ds_weights import ds_loss_weights
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss, MultiTalentLoss
    # MultiTalent_meets_resenc.py:157-170: net_numpool = len(net_num_pool_op_kernel_sizes) = 6 (the stem's [1,1,1] included), so all
    # five outputs of the residual-encoder network carry a weight (1, 1/2, 1/4, 1/8, 1/16) / 1.9375
    w = ds_loss_weights(len(RESENC_POOLS) if workload == 'resenc' else len(POOLS))
    if workload == 'task009':
        return DC_and_CE_DS_loss(w, batch_dice=False, ddp=ddp)
    return MultiTalentLoss(w, batch_dice=True)


def conv_flops(engine):
    """Algorithmic FLOPs of one fwd+bwd of the whole batch: fwd = sum_conv 2*|out|*Cin*k^3, fwd+bwd = 3x."""
    from multitalent_amd.engine import ConvNormOp, TConvOp
    f = 0.0
    for op in engine.ops:
        if isinstance(op, TConvOp):
            a = op.src.act
            f += 2.0 * a.N * a.V * op.tu.in_channels * op.tu.out_channels * int(np.prod(op.k))
        elif isinstance(op, ConvNormOp):
            a = op.out.act
            cin = sum(s.C for s in op.srcs)
            f += 2.0 * a.N * a.V * cin * a.C * int(np.prod(op.kernel))
    return f


def conv_bytes(engine, esize):
    """Algorithmic HBM bytes of one fwd+bwd of the whole batch in the fused ideal of SURVEY §8d: every conv / transposed conv reads its
    input once and writes its output once (norm + LeakyReLU folded into the consumer's load); backward reads dY and X and writes dX:
    sum_conv (3 |in| + 2 |out|) * esize."""
    from multitalent_amd.engine import ConvNormOp, TConvOp
    b = 0.0
    for op in engine.ops:
        if isinstance(op, TConvOp):
            a, o = op.src.act, op.out.act
            b += (3.0 * a.N * a.V * op.tu.in_channels + 2.0 * o.N * o.V * op.tu.out_channels) * esize
        elif isinstance(op, ConvNormOp):
            o = op.out.act
            b += (3.0 * o.N * sum(float(np.prod(s.spatial)) * s.C for s in op.srcs) + 2.0 * o.N * o.V * o.C) * esize
    return b


class ConvTimer:
    """HIP-event timing (on the stream the kernels are launched on = torch's current stream) of EVERY convolution launch of the
    C ABI — mt_conv3d_fwd (forward and backward-data), mt_conv3d_bwd_weight, mt_conv3d_bwd_data_strided — grouped by the device
    kernel that runs (the names rocprofv3 --kernel-trace prints), with the algorithmic work of each launch."""

    def __init__(self):
        self.rec = {}

    def __enter__(self):
        from multitalent_amd import ops
        self.ops = ops
        self.orig = (ops.conv3d_fwd, ops.conv3d_bwd_weight, ops.conv3d_bwd_data_strided)
        eb = lambda dt: 4.0 if dt == 0 else 2.0          # bytes per stored element (MT_F32 | MT_BF16 / MT_F16: mixed-precision storage)

        def timed(name, flops, nbytes, call):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            self.rec.setdefault(name, []).append((e0, e1, flops, nbytes))

        def fwd(p):
            # algorithmic work: 2 * |out| * Cin * k^3; a zero-inserted input (backward-data of a strided conv run as a
            # dilated-input conv) only carries 1/prod(dil) non-structural-zero taps.  Bytes: the input read once,
            # the output written once (read + written when accumulating)
            fl = 2.0 * p.N * p.Do * p.Ho * p.Wo * p.Cin * p.Cout * p.KD * p.KH * p.KW / (p.dilD * p.dilH * p.dilW)
            nb = p.N * (eb(p.src[0].dtype) * p.Di * p.Hi * p.Wi * p.Cin / (p.dilD * p.dilH * p.dilW) + eb(p.odtype) * p.Do * p.Ho * p.Wo * p.Cout * (2 if p.accumulate else 1))
            timed(ops.conv_kernel_name(p), fl, nb, lambda: self.orig[0](p))

        def bwdw(p, y, *a):
            # dW = X (*) dY: both activations read once; the weight gradient itself is negligible
            fl = 2.0 * p.N * p.Do * p.Ho * p.Wo * p.Cin * p.Cout * p.KD * p.KH * p.KW
            nb = p.N * (eb(p.src[0].dtype) * p.Di * p.Hi * p.Wi * p.Cin + eb(y.dt) * p.Do * p.Ho * p.Wo * p.Cout)
            timed(ops.conv_bwd_weight_kernel_name(p, y), fl, nb, lambda: self.orig[1](p, y, *a))

        def bwdd(p):
            # p = FORWARD geometry with src = dY [Do..] x Cout and out = dX [Di..] x Cin
            fl = 2.0 * p.N * p.Do * p.Ho * p.Wo * p.Cin * p.Cout * p.KD * p.KH * p.KW
            nb = p.N * (eb(p.src[0].dtype) * p.Do * p.Ho * p.Wo * p.Cout + eb(p.odtype) * p.Di * p.Hi * p.Wi * p.Cin * (2 if p.accumulate else 1))
            timed(ops.conv_bwd_data_strided_kernel_name(p), fl, nb, lambda: self.orig[2](p))

        ops.conv3d_fwd, ops.conv3d_bwd_weight, ops.conv3d_bwd_data_strided = fwd, bwdw, bwdd
        return self

    def __exit__(self, *exc):
        self.ops.conv3d_fwd, self.ops.conv3d_bwd_weight, self.ops.conv3d_bwd_data_strided = self.orig
        return False

    def groups(self):
        torch.cuda.synchronize()
        return {k: {"ms": sum(a.elapsed_time(b) for a, b, _, _ in v), "flops": sum(f for _, _, f, _ in v), "n": len(v),
                    "bytes": sum(nb for _, _, _, nb in v)} for k, v in self.rec.items()}


def roofline_from_groups(groups, nrep, precision):
    """`roofline` object for the kernel with the largest total time.  Bound:
      * fp32 matrix kernels: MFMA; `achieved` = EXECUTED matrix FLOP/s (Winograd kernels execute 64/216 resp. 16/36 of the
        direct convolution's multiplications — the algorithmic-equivalent rate is reported beside it under its own key);
      * bf16 matrix kernels: the tensors they move bound them (AI of a 32-channel 3x3x3 layer = 216 FLOP/B vs a machine balance
        of 312 for dense bf16): HBM, `achieved` = algorithmic bytes / duration."""
    name = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[name]
    sec = g["ms"] * 1e-3
    alg_tflops = g["flops"] / sec / 1e12
    ex = executed_fraction(name)
    out = {"kernel": name, "launches_per_step": g["n"] // nrep, "avg_launch_ms": round(g["ms"] / g["n"], 4),
           "algorithmic_gflop_per_launch": round(g["flops"] / g["n"] / 1e9, 2),
           "algorithmic_bytes_per_launch": int(g["bytes"] / g["n"]), "traffic": None}
    if 'tr16' in name:
        # 16-bit direct backward-weight: 654 FLOP/B algorithmic at 60 -> 60 channels (K = 27 taps x both channel
        # counts against two 2-byte tensors), above the dense 16-bit machine balance of 312: priced against the 16-bit MFMA peak
        out.update({"bound": "mfma", "achieved": round(alg_tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(alg_tflops / MFMA_BF16_PEAK_TFLOPS, 4), "hbm_gbs_algorithmic": round(g["bytes"] / sec / 1e9, 1),
                    "hbm_frac": round(g["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4)})
    elif 'bf16' in name or name.rstrip('>').endswith('true'):
        gbs = g["bytes"] / sec / 1e9
        out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_tflops": round(alg_tflops, 1), "mfma_frac_of_bf16_peak": round(alg_tflops / MFMA_BF16_PEAK_TFLOPS, 4)})
    else:
        out.update({"bound": "mfma", "achieved": round(alg_tflops * ex, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(alg_tflops * ex / MFMA_F32_PEAK_TFLOPS, 4)})
        if ex != 1.0:
            out["executed_fraction_of_direct_multiplications"] = round(ex, 4)
            out["algorithmic_equivalent"] = {"achieved": round(alg_tflops, 2), "frac_of_fp32_mfma_peak": round(alg_tflops / MFMA_F32_PEAK_TFLOPS, 4),
                                             "note": "direct-convolution FLOPs / time: may exceed 1 because Winograd executes fewer multiplications"}
    tot_ms = sum(v["ms"] for v in groups.values())
    out["all_conv_launches"] = {"ms_per_step": round(tot_ms / nrep, 3),
                                "algorithmic_tflops": round(sum(v["flops"] for v in groups.values()) / (tot_ms * 1e-3) / 1e12, 2),
                                "by_kernel_ms_per_step": {k: round(v["ms"] / nrep, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
    return out


def measure_roofline(step, x, largs, precision, nrep=3):
    """Per-launch HIP-event timing of every convolution launch over `nrep` more steps.  The engine issues the backward-weight launches
    on a side stream that overlaps the backward-data chain (Engine.weight_stream); under that overlap an event pair around a launch
    also counts the time the launch waits for CUs held by the other stream's kernels (a 0.6-ms kernel reads as 1.1 ms), which says
    nothing about the kernel.  This pass therefore serialises the two streams (engine.bwdw_streams = 0: same kernels, same order of
    launches per stream, one after the other) — `launches: "isolated"` — and the timed region above keeps the overlap."""
    eng = getattr(step, 'eng', None)
    saved = getattr(eng, 'bwdw_streams', None)
    if saved:
        eng.bwdw_streams = 0
        eng._packed_version = None
        eng._pack_programs = {}
    try:
        with ConvTimer() as t:
            for _ in range(nrep):
                step(x, *largs)
            groups = t.groups()
    finally:
        if saved:
            eng.bwdw_streams = saved
            eng._packed_version = None
            eng._pack_programs = {}
    out = roofline_from_groups(groups, nrep, precision)
    out["launches"] = "isolated (weight-gradient stream serialised for this pass)" if saved else "in step"
    return out


def measure_traffic(kernel, argv):
    """HBM bytes per launch of `kernel` from the PMC counters, measured NOW on this box: two child runs of this very command under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC block cannot hold both,
    MI355X_MICROARCH.md §rocprofv3 PMC slots; no other trace domain).  gfx950 correction (same guide, §HBM): FETCH_SIZE tallies the
    128-B requests of wide streaming reads at 64 B -> doubled (an upper bound for narrower accesses; the as-counted figure is kept);
    WRITE_SIZE as counted.  Both counters are in KB."""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return {"traffic": None, "traffic_error": "rocprofv3 not found"}
    res = {}
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='mt_pmc_', dir='/tmp')
        cmd = [rocprof, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '-o', 'b', '--', sys.executable,
               os.path.join(ROOT, 'bench.py')] + argv + ['--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-roofline', '--gpus', '1']
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420, check=True)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            tot, n = 0.0, 0
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name', ctr) != ctr:
                        continue
                    if r['Kernel_Name'].replace('void ', '').split('(')[0].strip() == kernel:
                        tot += float(r['Counter_Value'])
                        n += 1
            if n == 0:
                return {"traffic": None, "traffic_error": "kernel %s not found in the %s pass" % (kernel, ctr)}
            res[ctr] = tot / n
        except Exception as e:     # noqa: BLE001 — the bench line must still be printed
            return {"traffic": None, "traffic_error": "%s pass failed: %s" % (ctr, str(e)[:200])}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"traffic": int((2 * res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024),
            "traffic_as_counted": int((res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024),
            "traffic_unit": "bytes per launch, live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE x2 per MI355X_MICROARCH.md)"}


def host_cores():
    """(physical cores, logical cpus available to this process)."""
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    phys = None
    try:
        out = subprocess.check_output(['lscpu', '-p=CORE,SOCKET'], text=True)
        phys = len({l for l in out.splitlines() if l and not l.startswith('#')})
    except Exception:
        pass
    if not phys:
        phys = max(logical // 2, 1)
    return min(phys, logical), logical


def cpu_threads():
    """BASELINE.md §3 times the CPU path on the host's physical cores.  A thread sweep of this very iteration on the GPU box's
    host (tools/cpu_thread_sweep.py -> profiles/r02_cpu_thread_sweep.json) is committed; when it found a FASTER thread count for
    this host's core count, that one is used — the baseline is the best the host does, and `cores` says what was used."""
    phys, logical = host_cores()
    try:
        sw = json.load(open(os.path.join(ROOT, 'profiles', 'r02_cpu_thread_sweep.json')))
        if int(sw.get('physical_cores', -1)) == phys and int(sw['best_threads']) <= logical:
            return int(sw['best_threads']), phys
    except (OSError, ValueError, KeyError):
        pass
    return phys, phys


def cpu_iteration_fn(workload, B, patch=PATCH):
    """One full training iteration of the oracle (CPU restatement of the reference path) -> callable."""
    from oracle import reference_ops as R
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    torch.manual_seed(0)
    net = build_network(workload)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = synthetic_ct(B, patch, 99, 'cpu')
    params = list(sd.values())
    opt = torch.optim.SGD(params, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    if workload == 'task009':
        t = synthetic_targets(B, patch, ds_scales(POOLS), [[1]] * B, 99, 'cpu')
        w = R.ds_loss_weights(len(POOLS))
        loss = lambda: R.multiple_output_loss(R.generic_unet_forward(sd, x, POOLS, KERNELS), t, w)
    else:
        from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_regions, MultiTalent_region_output_idx_mapping,
                                                                            MultiTalent_valid_regions)
        names = list(MultiTalent_valid_regions.keys())
        valid = [MultiTalent_valid_regions[names[b % len(names)]] for b in range(B)]
        label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
        if workload == 'resenc':
            t = synthetic_targets(B, patch, ds_scales(RESENC_POOLS, skip_first=True), label_sets, 99, 'cpu')
            w = R.ds_loss_weights(len(RESENC_POOLS))
            fwd = lambda: R.fabians_unet_forward(sd, x, RESENC_POOLS, RESENC_KERNELS, RESENC_BLOCKS)
        else:
            t = synthetic_targets(B, patch, ds_scales(POOLS), label_sets, 99, 'cpu')
            w = R.ds_loss_weights(len(POOLS))
            fwd = lambda: R.generic_unet_forward(sd, x, POOLS, KERNELS)
        loss = lambda: R.multitalent_loss(list(fwd()), t, valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)[0]

    def iteration():
        opt.zero_grad()
        loss().backward()
        torch.nn.utils.clip_grad_norm_(params, 12)
        opt.step()
    return iteration


def cpu_baseline(workload, patch=PATCH):
    """the oracle's full training iteration (fwd + loss + bwd + clip + SGD) at B = 2, fp32, 3 warm-up + 5 timed
    iterations on the host cores (`cores` = threads used)."""
    threads, phys = cpu_threads()
    torch.set_num_threads(threads)
    B, WARM, NIT = 2, 3, 5
    it = cpu_iteration_fn(workload, B, patch)
    for _ in range(WARM):         # oneDNN primitive creation, page faults of ~20 GB of autograd buffers
        it()
    t0 = time.time()
    for _ in range(NIT):
        it()
    dt = time.time() - t0
    return {"value": round(NIT * B / dt, 4), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": "%d timed training iterations after %d warm-up (fwd+loss+bwd+clip+SGD), batch %d, patch %s, fp32, torch-CPU oracle "
                      "(oracle/reference_ops.py), %d threads on %d physical cores, %.1f s" % (NIT, WARM, B, 'x'.join(str(i) for i in patch), threads, phys, dt)}


def cpu_baseline_infer(patch, tiles_total, mirror):
    """Sliding-window CPU baseline on a reduced tile subset, extrapolated: the oracle's forward + sigmoid of ONE tile
    (x8 flips when mirroring) timed 1 + 2 times; volumes/min = 60 / (tiles_total * t_tile).  The host-side overlap-add of the
    reference (numpy, 333 MB per tile) is NOT included, which flatters the CPU."""
    from oracle import reference_ops as R
    threads, phys = cpu_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = build_network('task100')
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    x = torch.randn((1, 1) + tuple(patch))
    nflip = 8 if mirror else 1

    def tile():
        with torch.no_grad():
            for _ in range(nflip):
                torch.sigmoid(R.generic_unet_forward(sd, x, POOLS, KERNELS, deep_supervision=False))
    tile()
    t0 = time.time()
    for _ in range(2):
        tile()
    dt = (time.time() - t0) / 2
    return {"value": round(60.0 / (tiles_total * dt), 5), "unit": "volumes/min", "cores": threads, "kind": "port",
            "sample": "network forward + sigmoid of ONE %s tile (%d mirrored passes) timed twice after 1 warm-up = %.2f s per tile, extrapolated to "
                      "%d tiles; overlap-add excluded; torch-CPU oracle, %d threads on %d physical cores" % ('x'.join(str(i) for i in patch), nflip, dt, tiles_total, threads, phys)}


def bench_infer(args, dev, rank, world, ddp, emit=True):
    """BASELINE.json configs[4]: predict_MultiTalent-style sliding-window inference of ONE synthetic CT volume, tiles sharded
    over the ranks (strong scaling), Gaussian weighting, step 0.5, optional 8-fold mirroring; a 'step' is one whole volume.
    Metric: volumes per minute, volume already resident in host memory, result left on the device."""
    from multitalent_amd.inference.sliding_window import compute_steps_for_sliding_window, predict_3D
    patch = tuple(args.patch)
    torch.manual_seed(1234)
    net = build_network('task100').to(dev)
    net.eval()
    net.engine().set_precision(args.precision)
    net.inference_apply_nonlin = nn.Sigmoid()
    vol = np.random.RandomState(7).randn(1, *args.volume).astype(np.float32)
    shard = (rank, world) if world > 1 else None
    mixed = args.precision == 'bf16'          # predict_3D(mixed_precision=...): the reference's autocast switch
    mk = lambda mode: (lambda v=vol: predict_3D(net, v, bool(args.mirror), (0, 1, 2), True, 0.5, patch, None, True, 'constant', None, True,
                                                False, mixed, tile_shard=shard, return_device_tensors=mode))

    def timed(fn):
        for _ in range(max(1, min(args.warmup, 1))):
            fn()
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        dt = time.perf_counter() - t0
        if ddp:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce_dev(tmax, dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt
    run = mk(True)
    dt_sharded = timed(run)             # result left sharded on the devices (slab per rank)
    variants, check = None, None
    if world > 1:
        # the reference returns a WHOLE (seg, probabilities) per case (predict_MultiTalent.py:222-266): the gathered variant ends with the
        # whole mask on every rank (one uint8 all_gather: 134 MB at 512^3; the 25 GB of probabilities stay sharded) — that is `value`
        dt = timed(mk('mask'))
        variants = {"sharded_volumes_per_min": round(60.0 * args.steps / dt_sharded, 3), "mask_gathered_volumes_per_min": round(60.0 * args.steps / dt, 3),
                    "value_is": "mask_gathered"}
        # correctness inside the run: tile-sharded == unsharded on a smaller volume (own slab of the probabilities, whole gathered mask)
        small = np.random.RandomState(11).randn(1, 160, 256, 256).astype(np.float32)
        net._sliding_window_cache = None
        seg1, p1 = predict_3D(net, small, False, (0, 1, 2), True, 0.5, patch, None, True, 'constant', None, True, False, mixed, return_device_tensors=True)
        seg1, p1 = seg1.clone(), p1.clone()
        net._sliding_window_cache = None
        segf, pslab, (x0, x1) = predict_3D(net, small, False, (0, 1, 2), True, 0.5, patch, None, True, 'constant', None, True, False, mixed,
                                           tile_shard=shard, return_device_tensors='mask')
        dp = (pslab - p1[:, x0:x1]).abs().max() if x1 > x0 else torch.zeros((), device=dev)
        top2 = p1.topk(2, dim=0).values
        stable = (top2[0] - top2[1]) > 1e-4                           # voxels away from a tie of the argmax (regions_class_order=None)
        del top2
        nm = ((segf != seg1) & stable).sum().double()
        t = torch.stack([dp.double(), nm])
        all_reduce_dev(t, dist.ReduceOp.MAX)
        check = {"volume": [160, 256, 256], "max_abs_probability_difference_sharded_vs_unsharded": float(t[0]),
                 "mask_mismatches_away_from_ties": int(t[1]), "ok": bool(float(t[0]) < 1e-5 and int(t[1]) == 0)}
        net._sliding_window_cache = None
        del seg1, p1, segf, pslab
        torch.cuda.empty_cache()
    else:
        dt = dt_sharded
    groups = None
    if not args.no_roofline:          # every rank repeats one volume with per-launch events (the sharded run exchanges slabs)
        with ConvTimer() as t:
            run()
            groups = t.groups()
    if rank == 0:
        steps = compute_steps_for_sliding_window(patch, tuple(args.volume), 0.5)
        ntiles = len(steps[0]) * len(steps[1]) * len(steps[2])
        line = {
            "metric": "sliding-window vols/min", "value": round(60.0 * args.steps / dt, 3), "unit": "volumes/min",
            "n_gpus": world, "steps": args.steps, "warmup": 1, "ms_per_step": round(dt / args.steps * 1e3, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.precision == 'fp32' else "bf16", "data": "synthetic",
            "config": {"workload": "predict_MultiTalent sliding window, Generic_UNet nc=47 sigmoid", "volume": list(args.volume),
                       "patch": list(patch), "tiles": ntiles, "step_size": 0.5, "gaussian": True, "mirror_tta": bool(args.mirror),
                       "parallelism": "tile-shard%d" % world}}
        if groups is not None:
            line["roofline"] = roofline_from_groups(groups, 1, args.precision)
            line["roofline"]["launches"] = "in step (forward only: one stream)"
            if world == 1 and not args.no_traffic:
                line["roofline"].update(measure_traffic(line["roofline"]["kernel"], child_argv(args)))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_infer(patch, ntiles, bool(args.mirror))
        stats = getattr(net, '_slab_exchange_stats', None)
        if stats is not None:
            line["comm"] = stats
        if variants is not None:
            line["variants"] = variants
            line["sharded_equals_unsharded"] = check
    else:
        line = None
    del net
    if emit:
        if ddp:
            flush_c_stdio()
            dist.barrier()
        if rank == 0:
            print(json.dumps(line), flush=True)
        if ddp:
            dist.barrier()
            dist.destroy_process_group()
    return line


def child_argv(args):
    a = ['--workload', args.workload or 'task009', '--precision', args.precision, '--patch'] + [str(i) for i in args.patch]
    if args.batch:
        a += ['--batch', str(args.batch)]
    if (args.workload or '') == 'infer':
        a += ['--mirror', str(args.mirror), '--volume'] + [str(i) for i in args.volume]
    return a


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run and hand over."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get('MT_BENCH_ONE_GPU', '0') != '1':
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s); a data-parallel run needs one rank per GPU" % (n, have))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    raise SystemExit(subprocess.call(cmd, env=env))


DEFAULT_BATCH = {'task009': 2, 'task100': 4, 'resenc': 2}
WORKLOAD_NAMES = {"task009": "Task009_Spleen Generic_UNet nc=2 softmax Dice+CE",
                  "task100": "Task100_MultiTalent Generic_UNet nc=47 MultiTalent BCE+Dice loss",
                  "resenc": "Task100_MultiTalent FabiansUNet (residual encoder) nc=47 MultiTalent BCE+Dice loss"}


def all_reduce_dev(t, op):
    """dist.all_reduce of a small device tensor; through the host when the backend is gloo (MT_BENCH_ONE_GPU test mode)"""
    if dist.get_backend() == 'gloo' and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def time_training(workload, precision, patch, B, steps, warmup, dev, rank, world, ddp):
    """W untimed + K timed training iterations of `workload`, bracketed by barrier + synchronize, max over ranks."""
    from multitalent_amd.training.hot_loop import FusedTrainStep
    torch.manual_seed(1234)           # identical initial weights on all ranks (DDP broadcast semantics)
    net = build_network(workload)
    net.train()
    net.engine().set_precision(precision)
    step = FusedTrainStep(net, make_loss(workload, ddp), lr=1e-2, ddp=ddp)
    x, largs = make_batch(workload, B, dev, rank, patch)
    for _ in range(warmup):
        step(x, *largs)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    if step.reducer is not None:
        step.reducer.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step(x, *largs)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0           # this rank's own K steps, before it waits for the others
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rank_ms = None
    if ddp:
        tmax = torch.tensor([dt, dt_own, -dt_own], dtype=torch.float64, device=dev)
        all_reduce_dev(tmax, dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        rank_ms = {'max': round(float(tmax[1]) / steps * 1e3, 3), 'min': round(-float(tmax[2]) / steps * 1e3, 3)}
    loss = res[0] if isinstance(res, tuple) else res
    comm = step.reducer.stats() if step.reducer is not None else None
    if ddp:
        # the ranks must hold bit-identical parameters after the timed steps (same initial weights, all-reduced gradients, the same
        # optimizer arithmetic): a checksum of the flat parameter buffer, max - min over the ranks, must be exactly 0
        flat = step.eng.flat
        cs = torch.stack([flat.double().sum(), flat.double().abs().sum(), (flat.double() * torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.float64)).sum()])
        hi, lo = cs.clone(), cs.clone()
        all_reduce_dev(hi, dist.ReduceOp.MAX)
        all_reduce_dev(lo, dist.ReduceOp.MIN)
        spread = float((hi - lo).abs().max())
        comm = dict(comm or {})
        comm['param_checksum_spread_over_ranks'] = spread
        comm['params_identical_on_all_ranks'] = spread == 0.0
        comm['loss_finite'] = bool(np.isfinite(float(loss)))
        comm['ms_per_step_over_ranks_before_barrier'] = rank_ms
    return {'dt': dt, 'ms': dt / steps * 1e3, 'value': world * B * steps / dt, 'loss': float(loss), 'step': step, 'x': x, 'largs': largs,
            'net': net, 'comm': comm}


def training_line(r, workload, precision, patch, B, steps, warmup, world):
    fl = conv_flops(r['step'].eng) * 3.0
    pname = 'x'.join(str(i) for i in patch)
    line = {
        "metric": "CT patches/s (%s) train fwd+bwd" % pname, "value": round(r['value'], 3), "unit": "patches/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(r['ms'], 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == 'fp32' else "bf16", "data": "synthetic",
        "config": {"workload": WORKLOAD_NAMES[workload], "patch": list(patch), "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": "dp%d" % world, "step": "fwd+loss+bwd+clip12+SGD-nesterov",
                   "precision": "fp32" if precision == 'fp32' else
                   "mixed: fp16 activations (storage + forward products), bf16 gradients (storage + backward products), fp32 accumulation, master weights, norm statistics, loss, optimizer",
                   "final_loss": round(r['loss'], 5)},
        "algorithmic_tflop_per_step": round(fl / 1e12, 3),
        "step_frac_of_fp32_mfma_roofline": round(fl / (r['ms'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
    }
    # roofline of the step := max(t_HBM_alg, t_MFMA_alg) / t_measured with the figures of the run's dtype
    esz = 4 if precision == 'fp32' else 2
    nb = conv_bytes(r['step'].eng, esz)
    t_hbm = nb / (HBM_PEAK_GBS * 1e9)
    t_mfma = fl / ((MFMA_F32_PEAK_TFLOPS if precision == 'fp32' else MFMA_BF16_PEAK_TFLOPS) * 1e12)
    line["algorithmic_gb_per_step"] = round(nb / 1e9, 3)
    line["step_bound"] = {"t_hbm_ms": round(t_hbm * 1e3, 3), "t_mfma_ms": round(t_mfma * 1e3, 3), "bound": "hbm" if t_hbm > t_mfma else "mfma",
                          "mfma_peak_tflops": MFMA_F32_PEAK_TFLOPS if precision == 'fp32' else MFMA_BF16_PEAK_TFLOPS,
                          "frac": round(max(t_hbm, t_mfma) / (r['ms'] * 1e-3), 4)}
    if precision != 'fp32':
        line["frac_of_bf16_bound"] = line["step_bound"]["frac"]
        del line["step_frac_of_fp32_mfma_roofline"]      # a 16-bit run against the fp32 matrix peak says nothing
    if r.get('comm') is not None:
        line["comm"] = r['comm']
    return line


def measure_also(args, dev, rank, world, ddp):
    """The other BASELINE configs in the same process, after the headline workload (a few seconds each): configs[2] Task100 B = 4
    (the workload north_star's scaling target names), configs[3] residual encoder in fp32 and bf16, configs[4] sliding window of a
    512^3 volume without mirroring.  At N > 1 the training workloads run data-parallel over the same process group and the
    sliding window shards its tiles; each entry is whole-job throughput like `value`."""
    also = {}
    patch = tuple(args.patch)
    todo = [('task100', 'fp32'), ('resenc', 'fp32'), ('resenc', 'bf16')] if world == 1 else [('task100', 'fp32'), ('resenc', 'bf16')]
    fail = os.environ.get('MT_BENCH_INJECT_ALSO_FAILURE', '')      # tests: names of entries that raise, or 'all' (first-contact robustness)

    def guarded(name, fn):
        # an `also` entry must never take the headline line with it: whatever it raises becomes {"error": ...} under its name
        try:
            if fail == 'all' or name in fail.split(','):
                raise RuntimeError("injected failure in the also leg %r" % name)
            e = fn()
            if rank == 0:
                also[name] = e
        except Exception as ex:        # noqa: BLE001 - reported, not swallowed
            import traceback
            sys.stderr.write("bench.py: also[%s] failed on rank %d:\n%s\n" % (name, rank, traceback.format_exc()))
            also[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        net_cache_clear()

    def train_entry(workload, precision):
        B = DEFAULT_BATCH[workload]
        r = time_training(workload, precision, patch, B, args.also_steps, 3, dev, rank, world, ddp)
        rf = None if args.no_roofline else measure_roofline(r['step'], r['x'], r['largs'], precision, nrep=2)     # all ranks (collectives inside)
        e = training_line(r, workload, precision, patch, B, args.also_steps, 3, world)
        e = {k: e[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "algorithmic_tflop_per_step",
                               "step_frac_of_fp32_mfma_roofline", "frac_of_bf16_bound", "algorithmic_gb_per_step", "step_bound", "comm") if k in e}
        if rf is not None:
            e["roofline"] = rf
            if world == 1 and rank == 0 and not args.no_traffic:
                ca = ['--workload', workload, '--precision', precision, '--patch'] + [str(i) for i in patch]
                e["roofline"].update(measure_traffic(rf["kernel"], ca))
        return e

    for workload, precision in todo:
        guarded(workload + ('' if precision == 'fp32' else '_bf16'), lambda: train_entry(workload, precision))

    def infer_entry():
        ia = argparse.Namespace(**vars(args))
        ia.workload, ia.mirror, ia.steps, ia.warmup, ia.no_cpu_baseline, ia.no_traffic, ia.precision = 'infer', 0, 2, 1, True, True, 'fp32'
        e = bench_infer(ia, dev, rank, world, ddp, emit=False)
        return None if e is None else {k: e[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "scaling", "comm",
                                                         "variants", "sharded_equals_unsharded") if k in e}
    guarded('infer_512_nomirror', infer_entry)
    if world == 1:
        # the reference's DEFAULT inference mode — 8-fold mirror TTA (neural_network.py:502-591) — on a smaller volume (45 tiles x 8
        # mirrored passes; the 512^3 volume takes 17 s per pass in this mode), fp32
        ia = argparse.Namespace(**vars(args))
        ia.workload, ia.mirror, ia.steps, ia.warmup, ia.no_cpu_baseline, ia.no_traffic, ia.no_roofline, ia.precision = 'infer', 1, 1, 1, True, True, True, 'fp32'
        ia.volume = [128, 384, 384]

        def tta_entry():
            e = bench_infer(ia, dev, rank, world, ddp, emit=False)
            o = {k: e[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config") if k in e}
            o['network_passes_per_s'] = round(e['config']['tiles'] * 8 / (e['ms_per_step'] * 1e-3), 1)
            return o
        guarded('infer_128x384x384_mirror_tta', tta_entry)
    return also


def net_cache_clear():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default=None, choices=[None, 'task009', 'task100', 'resenc', 'infer'])
    ap.add_argument('--patch', type=int, nargs=3, default=list(PATCH), help='48 192 192 (BASELINE metric) or 96 192 192 (the plans\' native patch)')
    ap.add_argument('--volume', type=int, nargs=3, default=[512, 512, 512], help='--workload infer: synthetic CT volume')
    ap.add_argument('--mirror', type=int, default=1, help='--workload infer: 8-fold mirror TTA (reference default)')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help='bf16 = mixed precision (BASELINE configs[3]): bf16 matrix inputs, fp32 accumulation; the headline metric is fp32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='headline workload only (default run: + the other BASELINE configs under "also")')
    ap.add_argument('--also-steps', type=int, default=5)
    ap.add_argument('--no-traffic', action='store_true', help='skip the two rocprofv3 --pmc child passes (roofline.traffic = null)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # MT_BENCH_ONE_GPU=1 (tests on a one-GPU box): every rank on cuda:0 with gloo as the transport (device tensors of the collectives
    # travel through the host) — exercises the N > 1 code path, its checks and its JSON; the numbers mean nothing
    one_gpu = os.environ.get('MT_BENCH_ONE_GPU', '0') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ddp = world > 1 or ('RANK' in os.environ and int(os.environ.get('MT_FORCE_REDUCER', '0')))
    if ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo' if one_gpu else 'nccl', init_method='env://')
    workload = args.workload or 'task009'      # same per-GPU workload at every N (weak scaling on BASELINE configs[1])
    if workload == 'infer':
        return bench_infer(args, dev, rank, world, ddp)
    patch = tuple(args.patch)
    B = args.batch or DEFAULT_BATCH[workload]
    r = time_training(workload, args.precision, patch, B, args.steps, args.warmup, dev, rank, world, ddp)
    line = None
    if rank == 0:
        line = training_line(r, workload, args.precision, patch, B, args.steps, args.warmup, world)
    if ddp:
        # N > 1: the headline leaves the process as soon as it is measured — BEFORE the per-launch roofline pass (every rank, collectives
        # inside) and before the other configs (first RCCL contact of the sharded sliding window and of two more networks): a hang or a
        # crash in either must not cost the headline.  The complete line follows as the LAST line of the job with the same headline fields.
        flush_c_stdio()
        if rank == 0:
            print(json.dumps(dict(line, roofline="pending: the complete line follows", also="pending: the complete line follows")), flush=True)
    if not args.no_roofline:
        # every rank runs the per-launch pass (the step contains the gradient all-reduce); rank 0 reports its own launches
        rf = measure_roofline(r['step'], r['x'], r['largs'], args.precision)
        if rank == 0:
            line["roofline"] = rf
            if world == 1 and not args.no_traffic:
                line["roofline"].update(measure_traffic(line["roofline"]["kernel"], child_argv(args)))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(workload, patch)
    del r
    torch.cuda.empty_cache()
    if args.workload is None and not args.no_also:
        if ddp and not args.no_roofline:
            # second early line: now with the roofline object, still before the other configs run
            flush_c_stdio()
            dist.barrier()
            if rank == 0:
                print(json.dumps(dict(line, also="pending: the complete line follows")), flush=True)
        also = measure_also(args, dev, rank, world, ddp)
        if rank == 0:
            line["also"] = also
    if ddp:
        flush_c_stdio()
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


def flush_c_stdio():
    """RCCL writes its version banner to stdout through C stdio, which a pipe buffers until exit — it would land AFTER the JSON line.
    Every rank flushes C stdio before the barrier that precedes rank 0's print, so the JSON line is the last line of the job."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


if __name__ == '__main__':
    main()
