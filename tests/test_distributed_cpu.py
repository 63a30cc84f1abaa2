"""Multi-process CPU tests (gloo, world size 2) of everything the N>1 path adds on top of the single-GPU path:
the Dice-statistics exchange, the bucketed gradient all-reduce, and the per-rank batch / oversampling split."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import reference_ops as R


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run_world(fn, world=2):
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [ret[r] for r in range(world)]


def _dice_exchange(rank, world):
    """sum_over_ranks (one all_reduce each way) == awesome_allgather_function(x).sum(0) of the reference, forward and
    backward (distributed.py:28-73, MultiTalent_Trainer_DDP.py:598-606)."""
    from multitalent_amd.training.distributed_utils import sum_over_ranks
    g = torch.Generator().manual_seed(100 + rank)
    out = {}
    for name, f in (('ours', lambda t: sum_over_ranks(t)), ('ref', lambda t: R.AwesomeAllgather.apply(t).sum(0))):
        tp = (torch.rand((3, 47), generator=torch.Generator().manual_seed(100 + rank)) * 50).requires_grad_(True)
        fp = (torch.rand((3, 47), generator=torch.Generator().manual_seed(200 + rank)) * 50).requires_grad_(True)
        fn = (torch.rand((3, 47), generator=torch.Generator().manual_seed(300 + rank)) * 50).requires_grad_(True)
        if name == 'ours':
            s = f(torch.stack((tp, fp, fn), 0))
            TP, FP, FN = s[0], s[1], s[2]
        else:
            TP, FP, FN = f(tp), f(fp), f(fn)
        dc = (2 * TP / torch.clamp(2 * TP + FP + FN, min=1e-7)).sum() * (rank + 1)      # rank-dependent local loss
        dc.backward()
        out[name] = (float(dc), tp.grad.numpy().copy(), fp.grad.numpy().copy(), fn.grad.numpy().copy())
    return out


def test_dice_statistics_exchange_matches_reference_allgather():
    res = run_world(_dice_exchange)
    for r in res:
        assert abs(r['ours'][0] - r['ref'][0]) < 1e-4 * abs(r['ref'][0])
        for a, b in zip(r['ours'][1:], r['ref'][1:]):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-7)


class _FakeEngine:
    def __init__(self, n):
        self.flat_grad = torch.zeros(n)


def _grad_allreduce(rank, world):
    from multitalent_amd.training.hot_loop import GradAllReducer
    n = 100000
    eng = _FakeEngine(n)
    red = GradAllReducer(eng, bucket_bytes=4 * 30000)
    g = torch.Generator().manual_seed(rank)
    full = torch.randn(n, generator=g)
    red.begin()
    # gradients become final in 7 uneven slices (backward-completion order)
    cuts = [0, 5000, 41000, 41010, 77777, 90000, 99999, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        eng.flat_grad[lo:hi] = full[lo:hi]
        red.ready(lo, hi)
    red.ready(n, n)
    red.finish()
    assert red.sent == n
    return eng.flat_grad.numpy().copy(), full.numpy()


def test_bucketed_gradient_allreduce_is_the_mean():
    res = run_world(_grad_allreduce)
    mean = (res[0][1] + res[1][1]) / 2
    for r in res:
        assert np.allclose(r[0], mean, atol=1e-6)


def test_batch_size_and_oversample_split_matches_reference():
    from multitalent_amd.training.network_training.nnUNetTrainer import nnUNetTrainerV2_DDP

    class Fake:
        pass
    for world in (1, 2, 4, 8):
        for bs in (2, 4, 9):
            for dbs in (False, True):
                for rank in range(world):
                    exp_bs, exp_pct = R.set_batch_size_and_oversample(bs, world, rank, dbs)
                    if exp_bs <= 0:
                        continue
                    t = Fake(); t.batch_size = bs; t.oversample_foreground_percent = 0.33; t.distribute_batch_size = dbs
                    orig = (dist.get_world_size, dist.get_rank)
                    dist.get_world_size, dist.get_rank = (lambda: world), (lambda: rank)
                    try:
                        nnUNetTrainerV2_DDP.set_batch_size_and_oversample(t)
                    finally:
                        dist.get_world_size, dist.get_rank = orig
                    assert t.batch_size == exp_bs and abs(t.oversample_foreground_percent - exp_pct) < 1e-12, (world, bs, dbs, rank)


def test_dbs_with_more_ranks_than_samples_is_rejected():
    """8 ranks, plan batch 4, --dbs: ranks 4-7 would get no samples (nnUNetTrainerV2_DDP.py:87-98)."""
    from multitalent_amd.training.network_training.nnUNetTrainer import nnUNetTrainerV2_DDP

    class Fake:
        pass
    t = Fake(); t.batch_size = 4; t.oversample_foreground_percent = 0.33; t.distribute_batch_size = True
    orig = (dist.get_world_size, dist.get_rank)
    dist.get_world_size, dist.get_rank = (lambda: 8), (lambda: 6)
    try:
        with pytest.raises(RuntimeError):
            nnUNetTrainerV2_DDP.set_batch_size_and_oversample(t)
    finally:
        dist.get_world_size, dist.get_rank = orig


def test_tile_shard_plan_partitions_the_volume():
    """inference/sliding_window.shard_plan (pure function, identical on every rank): contiguous tile runs, owned slabs partition
    [0, X), every tile lies inside its rank's local range, and every contribution to an owned slab is either local or inside
    exactly the zones exchange_slabs ships (touched_q intersect owned_r)."""
    from multitalent_amd.inference.sliding_window import _intersect, compute_steps_for_sliding_window, shard_plan
    rng = np.random.RandomState(3)
    for _ in range(300):
        patch = tuple(int(i) for i in rng.randint(4, 40, 3))
        img = tuple(int(p + rng.randint(0, 120)) for p in patch)
        world = int(rng.randint(1, 9))
        steps = compute_steps_for_sliding_window(patch, img, float(rng.choice([0.5, 0.3, 1.0])))
        tiles = [(a, b, c) for a in steps[0] for b in steps[1] for c in steps[2]]
        plan = shard_plan(tiles, patch[0], img[0], world)
        assert sum(plan['tiles'], []) == tiles
        assert plan['owned'][0][0] == 0 and plan['owned'][-1][1] == img[0]
        for r in range(world):
            assert plan['owned'][r][0] <= plan['owned'][r][1]
            if r:
                assert plan['owned'][r][0] == plan['owned'][r - 1][1]
            for t in plan['tiles'][r]:
                assert plan['local'][r][0] <= t[0] and t[0] + patch[0] <= plan['local'][r][1]
                assert plan['touched'][r][0] <= t[0] and t[0] + patch[0] <= plan['touched'][r][1]
        # contributions: voxel row x of owner r receives from rank q iff some tile of q covers x; that row must be inside
        # touched_q intersect owned_r (shipped) or q == r (local)
        for r in range(world):
            for q in range(world):
                rows = set()
                for t in plan['tiles'][q]:
                    rows |= set(range(max(t[0], plan['owned'][r][0]), min(t[0] + patch[0], plan['owned'][r][1])))
                if not rows:
                    continue
                if q == r:
                    assert plan['local'][r][0] <= min(rows) and max(rows) < plan['local'][r][1]
                else:
                    z = _intersect(plan['touched'][q], plan['owned'][r])
                    assert z is not None and z[0] <= min(rows) and max(rows) < z[1]


def _gather(rank, world):
    """gather_slabs / slab_ranges / the growable slab cache of exchange_slabs on CPU tensors over gloo: every rank ends with the whole
    (seg, probabilities); ragged and EMPTY slabs; one cache entry per role however many volume shapes pass through."""
    from multitalent_amd.inference.sliding_window import (compute_steps_for_sliding_window, exchange_slabs, gather_slabs, shard_plan,
                                                          slab_ranges)
    rng = np.random.RandomState(11)
    cache = {}
    for X, Y, Z, px, pad in ((37, 5, 6, 8, 0), (16, 4, 4, 8, 3), (64, 3, 5, 16, 0)):
        steps = compute_steps_for_sliding_window((px, Y, Z), (X, Y, Z), 0.5)
        tiles = [(a, b, c) for a in steps[0] for b in steps[1] for c in steps[2]]
        plan = shard_plan(tiles, px, X, world)
        sx = slice(pad, X - pad)                       # the crop of pad_nd_image
        ranges = slab_ranges(plan, sx, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == X - 2 * pad and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        full_p = torch.from_numpy(rng.rand(3, X - 2 * pad, Y, Z).astype(np.float32))
        full_s = torch.from_numpy(rng.randint(0, 5, (X - 2 * pad, Y, Z)).astype(np.int32))
        a, b = ranges[rank]
        seg, probs = gather_slabs(full_s[a:b].clone(), full_p[:, a:b].clone(), ranges, world)
        assert torch.equal(seg, full_s) and seg.dtype == torch.int32 and torch.equal(probs, full_p)
        seg2, none = gather_slabs(full_s[a:b].clone(), None, ranges, world, max_label=4)        # byte wire format, decided by the job
        assert none is None and torch.equal(seg2, full_s)
        # labels >= 256 on ONE rank only — the wire type comes from max_label (rank-invariant), not from the rank's slab
        big = full_s.clone()
        big[ranges[world - 1][0]:ranges[world - 1][1]] += 300
        seg3, _ = gather_slabs(big[a:b].clone(), None, ranges, world, max_label=304)
        assert torch.equal(seg3, big)
        # exchange_slabs: partial aggregates of this rank's tiles -> the owned slab == the slab of the full sum
        C = 2
        tot_a, tot_n = torch.zeros((C, X, Y, Z)), torch.zeros((X, Y, Z))
        mine_a, mine_n = None, None
        for q in range(world):
            qa, qn = torch.zeros((C, X, Y, Z)), torch.zeros((X, Y, Z))
            for t in plan['tiles'][q]:
                w = torch.from_numpy(np.random.RandomState(hash(t) % 1000).rand(C, px, Y, Z).astype(np.float32))
                qa[:, t[0]:t[0] + px] += w
                qn[t[0]:t[0] + px] += 1
            tot_a += qa; tot_n += qn
            if q == rank:
                lo, hi = plan['local'][q]
                mine_a, mine_n = qa[:, lo:hi].contiguous(), qn[lo:hi].contiguous()
        fa, fn = exchange_slabs(mine_a, mine_n, plan, rank, world, cache=cache)
        o = plan['owned'][rank]
        assert torch.allclose(fa, tot_a[:, o[0]:o[1]], atol=1e-5) and torch.equal(fn, tot_n[o[0]:o[1]])
    # ONE flat buffer per role and device, however many shapes went through (the round-3 cache grew by an entry per shape)
    assert len(cache) <= 2 * world + 2 and all(v.dim() == 1 for v in cache.values()), sorted(cache)
    return len(cache)


@pytest.mark.parametrize("world", [2, 3])
def test_gather_slabs_and_slab_cache(world):
    run_world(_gather, world)
