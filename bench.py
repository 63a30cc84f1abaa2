#!/usr/bin/env python
"""Benchmark of the hot path: CT patches/s (48x192x192) for one training iteration
(forward + loss + backward + clip_grad_norm_(12) + SGD-Nesterov step) on synthetic device-resident batches,
dummyLoad style (reference: nnUNet_variants/benchmarking/nnUNetTrainerV2_dummyLoad.py:26-64).

  python bench.py --gpus N --steps K --warmup W
Workload (every N): BASELINE.json configs[1] = Task009_Spleen Generic_UNet, bs=2 per GPU, patch 48x192x192, fp32, softmax
Dice+CE with deep supervision.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL), weak scaling; gradients
all-reduced (mean) overlapped with backward on a side stream.  --workload task100 selects the nc=47 MultiTalent loss, bs=4.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
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
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


RESENC_POOLS = [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]     # SURVEY §8a N6 (resenc_bs4 plan)
RESENC_KERNELS = [[1, 3, 3]] + [[3, 3, 3]] * 5
RESENC_BLOCKS = [1, 2, 3, 4, 4, 4]


def build_network(workload):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.network_architecture.initialization import InitWeights_He
    if workload == 'resenc':        # BASELINE configs[3] architecture (FabiansUNet, MultiTalent_meets_resenc.py:72-104), here in fp32
        from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
        return FabiansUNet(1, 30, RESENC_BLOCKS, 2, RESENC_POOLS, RESENC_KERNELS, get_default_network_config(3, None, norm_type="in"),
                           47, [1] * (len(RESENC_POOLS) - 1), True, False, 320, InitWeights_He(1e-2))
    nc = 2 if workload == 'task009' else 47
    return Generic_UNet(1, 30, nc, len(POOLS), 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                        nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                        True, False, lambda x: x, InitWeights_He(1e-2), POOLS, KERNELS, False, True, True)


def make_batch(workload, B, dev, rank):
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
    scales = ds_scales(RESENC_POOLS, skip_first=True) if workload == 'resenc' else ds_scales(POOLS)
    x = synthetic_ct(B, PATCH, 1234 + rank, dev)
    if workload == 'task009':
        t = synthetic_targets(B, PATCH, scales, [[1]] * B, 1234 + rank, dev)
        return x, (t,)
    names = list(MultiTalent_valid_regions.keys())
    valid = [MultiTalent_valid_regions[names[(rank * B + b) % len(names)]] for b in range(B)]
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    t = synthetic_targets(B, PATCH, scales, label_sets, 1234 + rank, dev)
    return x, (t, valid)


def make_loss(workload, ddp):
    from multitalent_amd.training.ds_weights import ds_loss_weights
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss, MultiTalentLoss
    w = ds_loss_weights(len(RESENC_POOLS) - 1 if workload == 'resenc' else len(POOLS))
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


def measure_roofline(step, x, largs, nrep=3):
    """Per-launch HIP-event timing (on the stream the kernels are launched on = torch's current stream) of every
    mt_conv3d_fwd launch, grouped by the device kernel that runs (same names as rocprofv3 --kernel-trace).  Reports the
    dominant kernel: achieved = mean algorithmic FLOPs per launch / mean launch duration."""
    from multitalent_amd import ops
    rec = {}
    orig = ops.conv3d_fwd

    def timed(p):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(p)
        e1.record()
        # algorithmic work: 2 * |out| * Cin * k^3; a zero-inserted input (backward-data of a strided conv) only carries
        # 1/prod(dil) non-structural-zero taps
        flops = 2.0 * p.N * p.Do * p.Ho * p.Wo * p.Cin * p.Cout * p.KD * p.KH * p.KW / (p.dilD * p.dilH * p.dilW)
        # algorithmic bytes: the input read once, the output written once (read+written when accumulating), fp32
        nbytes = 4.0 * p.N * (p.Di * p.Hi * p.Wi * p.Cin + p.Do * p.Ho * p.Wo * p.Cout * (2 if p.accumulate else 1))
        rec.setdefault(ops.conv_kernel_name(p), []).append((e0, e1, flops, nbytes))

    ops.conv3d_fwd = timed
    try:
        for _ in range(nrep):
            step(x, *largs)
        torch.cuda.synchronize()
    finally:
        ops.conv3d_fwd = orig
    groups = {k: (sum(a.elapsed_time(b) for a, b, _, _ in v), sum(f for _, _, f, _ in v), len(v), sum(nb for _, _, _, nb in v))
              for k, v in rec.items()}
    name = max(groups, key=lambda k: groups[k][0])
    ms, fl, n, nby = groups[name]
    ach = fl / (ms * 1e-3) / 1e12
    all_ms = sum(g[0] for g in groups.values()); all_fl = sum(g[1] for g in groups.values())
    out = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "launches_per_step": n // nrep,
            "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
            "all_conv_fwd_launches": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2), "ms_per_step": round(all_ms / nrep, 3)}}
    # HBM bytes per launch of the dominant kernel from the PMC counters: rocprofv3 --pmc cannot run inside this process, so the
    # table is produced by tools/profile_pmc_bench.sh (separate FETCH_SIZE / WRITE_SIZE passes over THIS command, averaged over
    # the kernel's launches of a step) and committed as profiles/r01_pmc_per_kernel.json; gfx950 correction per
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (upper bound), WRITE_SIZE as counted
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_per_kernel.json')) as f:
            pmc = json.load(f)
        for prec in pmc.values():
            e = next((v for k, v in prec.items() if k.replace('void ', '').split('(')[0] == name), None)   # rocprofv3 prints "void name<...>(Params)"
            if e is not None:
                out["traffic"] = int((2 * e['fetch_kb_per_launch'] + e['write_kb_per_launch']) * 1024)
                out["traffic_as_counted"] = int((e['fetch_kb_per_launch'] + e['write_kb_per_launch']) * 1024)
                out["traffic_unit"] = "bytes per launch (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, profiles/r01_pmc_per_kernel.json)"
                out["algorithmic_bytes_per_launch"] = int(nby / n)
                break
    except (OSError, ValueError, KeyError):
        pass
    if name.startswith('conv_bf16'):
        # bf16 matrix inputs: 27 MFMAs per 16-channel chunk instead of 216 — the kernel is bound by moving the fp32 activations
        gbs = nby / (ms * 1e-3) / 1e9
        out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "algorithmic_mbytes_per_launch": round(nby / n / 1e6, 1), "algorithmic_tflops": round(ach, 1)})
    if name.startswith('conv_wino'):
        # `achieved` counts the ALGORITHMIC FLOPs of the direct convolution; the Winograd F(2x2x2,3x3x3) kernel executes
        # 64/216 of them on the matrix cores (plus the transforms on the vector ALU), so frac may exceed what a direct kernel can
        out["algorithm"] = "winograd F(2x2x2,3x3x3): executed MFMA FLOPs = algorithmic / 3.375"
        out["executed_mfma_tflops"] = round(ach / 3.375, 2)
    return out


def cpu_baseline(workload):
    """Oracle (CPU restatement of the reference path) timed on the host cores: one full training iteration at B=1."""
    from oracle import reference_ops as R
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    torch.manual_seed(0)
    net = build_network(workload)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 32)     # oneDNN conv3d stops scaling (and 256 SMT threads thrash) well before a whole 2-socket host
    torch.set_num_threads(cores)
    B = 1
    x = synthetic_ct(B, PATCH, 99, 'cpu')
    t = synthetic_targets(B, PATCH, ds_scales(POOLS), [[1]] * B, 99, 'cpu')
    w = R.ds_loss_weights(len(POOLS))
    params = list(sd.values())
    opt = torch.optim.SGD(params, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    def iteration():
        opt.zero_grad()
        out = R.generic_unet_forward(sd, x, POOLS, KERNELS)
        if workload == 'task009':
            loss = R.multiple_output_loss(out, t, w)
        else:
            from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_regions, MultiTalent_region_output_idx_mapping,
                                                                                MultiTalent_valid_regions)
            loss = R.multitalent_loss(list(out), t, [MultiTalent_valid_regions['Task009_Spleen']], MultiTalent_regions,
                                      MultiTalent_region_output_idx_mapping, w)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 12)
        opt.step()

    iteration()                       # warm-up: oneDNN primitive creation, page faults of ~10 GB of autograd buffers
    NIT = 3
    t0 = time.time()
    for _ in range(NIT):
        iteration()
    dt = time.time() - t0
    return {"value": round(NIT * B / dt, 4), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": "%d training iterations after 1 warm-up (fwd+loss+bwd+clip+SGD), batch 1, patch 48x192x192, fp32, torch CPU oracle, %.1f s" % (NIT, dt)}


def bench_infer(args, dev, rank, world, ddp):
    """BASELINE.json configs[4]: predict_MultiTalent-style sliding-window inference of ONE synthetic CT volume, tiles sharded
    over the ranks (strong scaling), Gaussian weighting, step 0.5, optional 8-fold mirroring; a 'step' is one whole volume.
    Metric: volumes per minute, volume already resident in host memory, result left on the device."""
    from multitalent_amd.inference.sliding_window import predict_3D
    torch.manual_seed(1234)
    net = build_network('task100').to(dev)
    net.eval()
    net.engine().set_precision(args.precision)
    net.inference_apply_nonlin = nn.Sigmoid()
    vol = np.random.RandomState(7).randn(1, *args.volume).astype(np.float32)
    shard = (rank, world) if world > 1 else None
    run = lambda: predict_3D(net, vol, bool(args.mirror), (0, 1, 2), True, 0.5, PATCH, None, True, 'constant', None, True,
                             False, True, tile_shard=shard, return_device_tensors=True)
    for _ in range(max(1, min(args.warmup, 1))):
        run()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "sliding-window vols/min", "value": round(60.0 * args.steps / dt, 3), "unit": "volumes/min",
            "n_gpus": world, "steps": args.steps, "warmup": 1, "ms_per_step": round(dt / args.steps * 1e3, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.precision == 'fp32' else "bf16", "data": "synthetic",
            "config": {"workload": "predict_MultiTalent sliding window, Generic_UNet nc=47 sigmoid", "volume": list(args.volume),
                       "patch": list(PATCH), "step_size": 0.5, "gaussian": True, "mirror_tta": bool(args.mirror),
                       "parallelism": "tile-shard%d" % world}}), flush=True)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default=None, choices=[None, 'task009', 'task100', 'resenc', 'infer'])
    ap.add_argument('--volume', type=int, nargs=3, default=[512, 512, 512], help='--workload infer: synthetic CT volume')
    ap.add_argument('--mirror', type=int, default=1, help='--workload infer: 8-fold mirror TTA (reference default)')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help='bf16 = mixed precision (BASELINE configs[3]): bf16 matrix inputs, fp32 accumulation/storage; the headline metric is fp32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ddp = world > 1 or ('RANK' in os.environ and int(os.environ.get('MT_FORCE_REDUCER', '0')))
    if ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', init_method='env://')
    workload = args.workload or 'task009'      # same per-GPU workload at every N (weak scaling on BASELINE configs[1])
    if workload == 'infer':
        return bench_infer(args, dev, rank, world, ddp)
    B = args.batch or {'task009': 2, 'task100': 4, 'resenc': 2}[workload]

    from multitalent_amd.training.hot_loop import FusedTrainStep
    torch.manual_seed(1234)           # identical initial weights on all ranks (DDP broadcast semantics)
    net = build_network(workload)
    net.train()
    net.engine().set_precision(args.precision)
    step = FusedTrainStep(net, make_loss(workload, ddp), lr=1e-2, ddp=ddp)
    x, largs = make_batch(workload, B, dev, rank)

    for _ in range(args.warmup):
        step(x, *largs)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(x, *largs)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ddp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = res[0] if isinstance(res, tuple) else res
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    line = None
    if rank == 0:
        fl = conv_flops(step.eng) * 3.0
        line = {
            "metric": "CT patches/s (48x192x192) train fwd+bwd", "value": round(value, 3), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 'fp32' else "bf16", "data": "synthetic",
            "config": {"workload": {"task009": "Task009_Spleen Generic_UNet nc=2 softmax Dice+CE",
                                    "task100": "Task100_MultiTalent Generic_UNet nc=47 MultiTalent BCE+Dice loss",
                                    "resenc": "Task100_MultiTalent FabiansUNet (residual encoder) nc=47 MultiTalent BCE+Dice loss"}[workload],
                       "patch": list(PATCH), "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world, "step": "fwd+loss+bwd+clip12+SGD-nesterov",
                       "precision": "fp32" if args.precision == 'fp32' else
                       "bf16 matrix inputs + fp32 accumulation in the 3x3x3 stride-1 convs (fwd, bwd-data, bwd-weight); fp32 storage, norm, loss, optimizer, other layers",
                       "final_loss": round(float(loss), 5)},
            "algorithmic_tflop_per_step": round(fl / 1e12, 3),
            "step_frac_of_mfma_roofline": round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
        }
    if rank == 0 and world == 1:
        if not args.no_roofline:
            line["roofline"] = measure_roofline(step, x, largs)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(workload)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
