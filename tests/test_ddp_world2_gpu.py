"""World-size-2 parity of the HIP path against goldens produced by TWO gloo processes around the REAL reference
(tests/golden/ddp_w2.npz, tools/oracle_gen/make_golden_ddp.py): cross-rank batch Dice (L1/L3/L5), online evaluation with its
rank gather (L4) and two full DDP training iterations — gradient mean over ranks through GradAllReducer, clip 12, SGD-Nesterov
(T1/T3).  Two processes: one per GPU over RCCL ('nccl') when the box has >= 2 GPUs, otherwise both on cuda:0 with gloo as the
transport (a few KB of statistics and the 117 KB gradient of the toy network travel through the host; every kernel is the HIP one)."""
import json
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    multi = torch.cuda.device_count() >= world
    torch.cuda.set_device(rank if multi else 0)
    dist.init_process_group('nccl' if multi else 'gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world, torch.device('cuda', rank if multi else 0))
    finally:
        dist.destroy_process_group()


def run_world_gpu(fn, world=2):
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "rank process failed (exit code %s)" % p.exitcode
    return [ret[r] for r in range(world)]


def _losses(rank, world, dev):
    from multitalent_amd.training.distributed_utils import sum_over_ranks
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss, MultiTalentLoss
    from multitalent_amd.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
    z = np.load(os.path.join(G, 'ddp_w2.npz'))
    meta = json.load(open(os.path.join(G, 'ddp_w2.json')))
    p = 'r%d/' % rank
    valid = meta['valid_regions'][rank]
    # L3: gather + sum over the rank axis == one all_reduce each way
    x = torch.from_numpy(z[p + 'ag/x']).to(dev).requires_grad_(True)
    y = sum_over_ranks(x)
    coef = torch.from_numpy(z[p + 'ag/coef']).to(dev)
    (y[None] * coef).sum().backward()              # the reference's consumer sums the gathered [W,B,C] over W
    assert np.allclose(y.detach().cpu().numpy(), z[p + 'ag/y'].sum(0), atol=1e-6)
    # d/dx of sum_w coef[w] * y: the reference's backward all-reduces the gathered gradient and selects the own slice; with the
    # `.sum(0)` consumer every slice of that gradient is sum_ranks(sum_w coef_r[w])
    other = 'r%d/' % (1 - rank)
    assert np.allclose(x.grad.cpu().numpy(), z[p + 'ag/coef'].sum(0) + z[other + 'ag/coef'].sum(0), atol=1e-5)
    # L1: MultiTalent loss, batch Dice summed over ranks
    for bd in (1, 0):
        logits = [torch.from_numpy(z[p + 'mt/logits%d' % i]).to(dev).requires_grad_(True) for i in range(2)]
        tg = [torch.from_numpy(z[p + 'mt/target%d' % i]).to(dev) for i in range(2)]
        l, ce, dc = MultiTalentLoss(z[p + 'mt/weights'], batch_dice=bool(bd))(logits, tg, valid)
        l.backward()
        got = np.array([float(l.detach()), float(ce.detach()), float(dc.detach())])
        assert np.allclose(got, z[p + 'mt/bd%d/loss' % bd], rtol=1e-4, atol=1e-4), (bd, got, z[p + 'mt/bd%d/loss' % bd])
        for i in range(2):
            ref = z[p + 'mt/bd%d/dlogits%d' % (bd, i)]
            assert np.abs(logits[i].grad.cpu().numpy() - ref).max() < 1e-6 + 1e-4 * np.abs(ref).max(), (bd, i)
    # L5 DDP flavour
    for bd in (1, 0):
        sl = [torch.from_numpy(z[p + 'sm/logits%d' % i]).to(dev).requires_grad_(True) for i in range(2)]
        stg = [torch.from_numpy(z[p + 'sm/target%d' % i]).to(dev) for i in range(2)]
        l = DC_and_CE_DS_loss(z[p + 'mt/weights'], batch_dice=bool(bd), ddp=True)(sl, stg)
        l.backward()
        assert abs(float(l.detach()) - float(z[p + 'sm/bd%d/loss' % bd])) < 1e-5
        for i in range(2):
            ref = z[p + 'sm/bd%d/dlogits%d' % (bd, i)]
            assert np.abs(sl[i].grad.cpu().numpy() - ref).max() < 1e-7 + 1e-4 * np.abs(ref).max(), (bd, i)
    # L4: online evaluation incl. the gather over ranks and the epoch metric
    logs = []
    t = SimpleNamespace(train_step=SimpleNamespace(loss_fn=MultiTalentLoss([1.0])), online_eval_foreground_dc=[], online_eval_tp=[],
                        online_eval_fp=[], online_eval_fn=[], all_val_eval_metrics=[], print_to_log_file=lambda *a, **k: logs.append(a))
    for it in range(2):
        MultiTalent_trainer_ddp.run_online_evaluation(t, [torch.from_numpy(z[p + 'oe/out%d' % it]).to(dev)],
                                                      [torch.from_numpy(z[p + 'oe/target%d' % it]).to(dev)], valid)
    for k, lst in (('tp', t.online_eval_tp), ('fp', t.online_eval_fp), ('fn', t.online_eval_fn)):
        assert np.array_equal(np.array(lst, dtype=np.float64), z[p + 'oe/' + k]), k          # exact integer counts
    assert np.allclose(np.array(t.online_eval_foreground_dc, dtype=np.float64), z[p + 'oe/foreground_dc'], atol=1e-7)
    MultiTalent_trainer_ddp.finish_online_evaluation(t)
    assert abs(float(t.all_val_eval_metrics[0]) - float(z[p + 'oe/all_val_eval_metrics'][0])) < 1e-9
    assert t.online_eval_tp == []
    return dist.get_backend()


def test_world2_losses_and_online_evaluation_match_real_reference(dev):
    res = run_world_gpu(_losses)
    assert res[0] == res[1]


def _train(rank, world, dev):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    z = np.load(os.path.join(G, 'ddp_w2.npz'))
    meta = json.load(open(os.path.join(G, 'ddp_w2.json')))
    p = 'r%d/train/' % rank
    valid = meta['valid_regions'][rank]
    pools, kernels = z[p + 'pools'].tolist(), z[p + 'kernels'].tolist()
    net = Generic_UNet(1, 6, 47, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    net.load_state_dict({k[len(p) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p + 'sd0/')})
    net.train()
    step = FusedTrainStep(net, MultiTalentLoss(z[p + 'weights'], batch_dice=True), lr=1e-2, ddp=True)
    assert step.reducer is not None and step.reducer.world == 2
    x = torch.from_numpy(z[p + 'x']).to(dev)
    tg = [torch.from_numpy(z[p + 'target%d' % i]).to(dev) for i in range(2)]
    for it in range(2):
        l, ce, dc = step(x, tg, valid)
        got = np.array([float(l), float(ce), float(dc)])
        assert np.allclose(got, z[p + 'losses'][it], rtol=2e-4, atol=2e-4), (it, got, z[p + 'losses'][it])
        if it == 0:
            eng = net.engine()
            for n, prm in net.named_parameters():
                ref = z[p + 'grad0/' + n]               # DDP: mean over the two ranks
                g = eng.grad_of(prm).cpu().numpy()
                assert np.abs(g - ref).max() < 2e-3 * max(np.abs(ref).max(), 1e-3), n
    torch.cuda.synchronize()
    worst = 0.0
    for k, v in net.state_dict().items():
        worst = max(worst, float(np.abs(v.cpu().numpy() - z[p + 'sd2/' + k]).max()))
    assert worst < 1e-4, worst
    return worst, step.reducer.via_host


def test_world2_two_ddp_training_iterations_match_real_reference(dev):
    """T3: the reference wraps the network in torch DDP (MultiTalent_Trainer_DDP.py:121); here GradAllReducer averages the flat
    gradient buffer bucket by bucket.  Both ranks must end with the reference's parameters (and therefore identical ones)."""
    res = run_world_gpu(_train)
    assert all(r[0] < 1e-4 for r in res)


def _reducer_rccl(rank, world, dev):
    """bucketed side-stream all-reduce on DEVICE tensors: uneven completion slices, more buckets than slices."""
    from multitalent_amd.training.hot_loop import GradAllReducer
    n = 3_000_000
    eng = SimpleNamespace(flat_grad=torch.zeros(n, device=dev))
    red = GradAllReducer(eng, bucket_bytes=4 * 500_000)
    full = torch.randn(n, generator=torch.Generator().manual_seed(rank)).to(dev)
    other = torch.randn(n, generator=torch.Generator().manual_seed(1 - rank)).to(dev)
    for rep in range(3):                                   # the streams / events are re-used across steps
        eng.flat_grad.zero_()
        red.begin()
        cuts = [0, 5000, 1_410_000, 1_410_010, 2_777_777, 2_900_000, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eng.flat_grad[lo:hi] = full[lo:hi] * (rep + 1)
            red.ready(lo, hi)
        red.ready(n, n)
        red.finish()
        assert red.sent == n
        torch.cuda.synchronize()
        assert torch.allclose(eng.flat_grad, (full + other) * (rep + 1) / 2, atol=1e-6)
    return True


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
def test_grad_allreducer_on_rccl(dev):
    assert all(run_world_gpu(_reducer_rccl))


def test_grad_allreducer_device_tensors_world2(dev):
    """same, on whatever transport the box offers (gloo through the host on a one-GPU box)."""
    assert all(run_world_gpu(_reducer_rccl))


def _sharded_inference(rank, world, dev):
    """predict_3D with tile_shard=(rank, world): slab ownership + boundary exchange == the unsharded run of the same process
    (probabilities to fp32 rounding of a re-associated sum, masks identical away from ties) == the real reference's output."""
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    z = dict(np.load(os.path.join(G, 'sliding_window.npz')))
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    net = Generic_UNet(1, 6, 5, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    net.load_state_dict({k[len('mt/sd/'):]: torch.from_numpy(v) for k, v in z.items() if k.startswith('mt/sd/')})
    net.to(dev)
    net.inference_apply_nonlin = nn.Sigmoid()
    net.eval(); net.do_ds = False
    worst = 0.0
    for mirror in (True, False):
        kw = dict(do_mirroring=mirror, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5, patch_size=(8, 16, 16),
                  regions_class_order=[3, 1, 4, 2, 5], use_gaussian=True, pad_border_mode='constant', pad_kwargs={'constant_values': 0},
                  all_in_gpu=False, verbose=False, mixed_precision=False)
        seg1, p1 = net.predict_3D(z['mt/vol'], **kw)
        net._sliding_window_cache = None
        segs, ps = net.predict_3D(z['mt/vol'], tile_shard=(rank, world), **kw)
        net._sliding_window_cache = None
        assert ps.shape == p1.shape and segs.shape == seg1.shape
        d = float(np.abs(ps - p1).max())
        worst = max(worst, d)
        assert d < 2e-6, d
        safe = (np.abs(p1 - 0.5) > 1e-5).all(0)
        assert np.array_equal(segs[safe], seg1[safe])
        ref_p, ref_s = z['mt/probs_m%d' % int(mirror)], z['mt/seg_m%d' % int(mirror)]
        assert np.abs(ps - ref_p).max() < 1e-4
        from mask_check import check_masks
        check_masks(segs, ref_s, ps, ref_p, [3, 1, 4, 2, 5], 1e-4, 'tile-sharded predict_3D rank %d/%d mirror=%d' % (rank, world, int(mirror)),
                    key='tile-sharded predict_3D world %d mirror=%d' % (world, int(mirror)))
    return worst


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_inference_matches_unsharded_and_reference(dev, world):
    if torch.cuda.device_count() >= 2 and torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs or exactly one" % world)
    res = run_world_gpu(_sharded_inference, world)
    assert all(r < 2e-6 for r in res)


def _rccl_world1(rank, world, dev):
    """One rank, backend nccl (= RCCL): the bucketed side-stream all-reduce runs real RCCL kernels (MT_FORCE_REDUCER=1 keeps the
    world-size-1 short cut off) between backward and the optimizer; the result must equal the step without a reducer bit for bit."""
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    os.environ['MT_FORCE_REDUCER'] = '1'
    pools, kernels = [[2, 2, 2], [2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 4
    res = []
    for ddp in (False, True):
        torch.manual_seed(3)
        net = Generic_UNet(1, 16, 3, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, None, pools, kernels, False, True, True)
        net.train()
        step = FusedTrainStep(net, DC_and_CE_DS_loss([4 / 7, 2 / 7, 1 / 7], batch_dice=False, ddp=True), lr=1e-2, ddp=ddp)
        if ddp:
            step.reducer.bucket = 20000               # many buckets: several all-reduces in flight on the side stream
            assert step.reducer.force and not step.reducer.__dict__.get('via_host', False)
        g = torch.Generator().manual_seed(4)
        x = torch.randn((2, 1, 16, 32, 32), generator=g).to(dev)
        t0 = torch.randint(0, 3, (2, 1, 16, 32, 32), generator=g).float()
        tg = [t0.to(dev), t0[:, :, ::2, ::2, ::2].contiguous().to(dev), t0[:, :, ::4, ::4, ::4].contiguous().to(dev)]
        losses = [float(step(x, tg)) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()))
        if ddp:
            assert len(step.reducer.handles) > 3 and step.reducer.stream is not None
            st = step.reducer.stats()
            assert st['allreduce_bytes_per_step'] == 4 * net.engine().flat_grad.numel() and st['exposed_wait_ms_per_step'] >= 0 and st['side_stream_busy_ms_per_step'] > 0
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    return True


def _worker_nccl1(fn, ret, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        ret[0] = fn(0, 1, torch.device('cuda', 0))
    finally:
        dist.destroy_process_group()


def test_grad_allreducer_side_stream_on_rccl_world1(dev):
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker_nccl1, args=(_rccl_world1, ret, _free_port()))
    p.start()
    p.join(600)
    assert p.exitcode == 0 and ret[0] is True


def _launcher_free_env():
    """os.environ without the rendezvous variables other tests of this process set for their single-rank process groups (test_drivers_gpu,
    test_trainer_gpu: RANK / WORLD_SIZE): bench.py --gpus N must start its own ranks."""
    return {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'MT_FORCE_REDUCER')}


@pytest.mark.parametrize("extra", [['--batch', '1'], ['--workload', 'infer', '--mirror', '0', '--volume', '96', '256', '256']])
def test_bench_two_ranks_self_validation_on_one_gpu(extra):
    """The N > 1 code path of bench.py with both ranks on cuda:0 and gloo as the transport (MT_BENCH_ONE_GPU=1; a box with one GPU
    cannot run RCCL with two ranks): the line must carry the correctness signals of a multi-GPU run — bit-identical parameters on all
    ranks after the timed steps; the tile-sharded sliding window equal to the unsharded one, the mask-gathered variant as `value`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-also', '--no-roofline'] + extra,
                         capture_output=True, text=True, timeout=1200, env=dict(_launcher_free_env(), MT_BENCH_ONE_GPU='1'))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and 'comm' in line, line
    if 'infer' in extra:
        assert line['sharded_equals_unsharded']['ok'], line['sharded_equals_unsharded']
        assert line['variants']['value_is'] == 'mask_gathered' and line['variants']['sharded_volumes_per_min'] >= line['value'] * 0.8    # (separately timed passes: noise, not an ordering guarantee)
    else:
        assert line['comm']['params_identical_on_all_ranks'] and line['comm']['param_checksum_spread_over_ranks'] == 0.0, line['comm']
        assert line['comm']['loss_finite']


@pytest.mark.parametrize("extra", [['--patch', '16', '64', '64', '--batch', '1'],
                                   ['--workload', 'infer', '--mirror', '0', '--volume', '96', '384', '384']])
def test_bench_eight_ranks_on_one_gpu(extra):
    """World size 8 end to end (VERDICT r5 #8c) with all ranks on cuda:0 and gloo as the transport: the gradient buckets of 8 ranks
    (bucket boundaries, mean over 8, bit-identical parameters afterwards) on a reduced patch, and the sliding window's shard_plan /
    exchange_slabs / gather_slabs with 27 tiles over 8 ranks (uneven shares, padded slabs) equal to the unsharded result."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--no-also', '--no-roofline'] + extra,
                         capture_output=True, text=True, timeout=1800, env=dict(_launcher_free_env(), MT_BENCH_ONE_GPU='1'))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['value'] > 0 and 'comm' in line, line
    if 'infer' in extra:
        assert line['sharded_equals_unsharded']['ok'], line['sharded_equals_unsharded']
        assert line['config']['tiles'] == 27
    else:
        assert line['comm']['world'] == 8 and line['comm']['params_identical_on_all_ranks'] and line['comm']['param_checksum_spread_over_ranks'] == 0.0, line['comm']
        assert line['comm']['loss_finite']


def test_bench_headline_survives_a_failing_also_leg():
    """First contact of the N > 1 default run: the headline line is printed BEFORE the other configs run, and an
    exception inside any `also` entry becomes {"error": ...} under its name in the complete line instead of losing the job's output."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-roofline'],
                         capture_output=True, text=True, timeout=1200,
                         env=dict(_launcher_free_env(), MT_BENCH_ONE_GPU='1', MT_BENCH_INJECT_ALSO_FAILURE='all'))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 2, out.stdout[-2000:]
    first, last = lines
    assert isinstance(first['also'], str) and isinstance(first['roofline'], str) and first['value'] == last['value'] and first['n_gpus'] == 2
    assert last['comm']['params_identical_on_all_ranks'] and last['comm']['ms_per_step_over_ranks_before_barrier']['max'] >= \
        last['comm']['ms_per_step_over_ranks_before_barrier']['min'] > 0
    assert set(last['also']) >= {'task100', 'resenc_bf16', 'infer_512_nomirror'}
    assert all('injected failure' in e['error'] for e in last['also'].values()), last['also']


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
@pytest.mark.parametrize("extra", [[], ['--workload', 'task100'], ['--workload', 'infer', '--mirror', '0', '--volume', '160', '256', '256']])
def test_bench_two_ranks_smoke(extra):
    """`bench.py --gpus 2` (self-spawning under torch.distributed.run) for the headline workload, Task100 and the tile-sharded sliding
    window: one JSON line with n_gpus 2 and the communication accounting (`comm`) of the gradient all-reduce / slab exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-also'] + extra,
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and 'comm' in line, line
    if 'infer' in extra:
        assert line['comm']['bytes_sent'] > 0 and line['scaling'] == 'strong'
        assert line['sharded_equals_unsharded']['ok'] and line['variants']['value_is'] == 'mask_gathered'
    else:
        assert line['comm']['allreduce_bytes_per_step'] > 1e8 and line['comm']['world'] == 2 and line['scaling'] == 'weak'
        assert line['comm']['params_identical_on_all_ranks']
