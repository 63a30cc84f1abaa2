"""Pins the oracle's world-size > 1 restatements (L1 batch Dice, L3 awesome_allgather, L4 online evaluation, L5 DDP flavour, T3 batch
split) and the round-2 single-process additions (non-uniform kernel sizes, pad path, MultiTalent folds) to the REAL reference:
tests/golden/ddp_w2.* were produced by W = 2 gloo processes around the imported reference (tools/oracle_gen/make_golden_ddp.py),
plain_unet_aniso / sliding_window_pad / multitalent_splits by tools/oracle_gen/make_golden_r2.py.  CPU only."""
import json
import os
import pickle

import numpy as np
import torch

from oracle import reference_ops as R
from tests.test_distributed_cpu import run_world

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _tables():
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_region_output_idx_mapping, MultiTalent_regions
    return MultiTalent_regions, MultiTalent_region_output_idx_mapping


def _oracle_w2(rank, world):
    """every rank replays its golden inputs through the ORACLE under a real gloo group and returns what it computed."""
    z = np.load(os.path.join(G, 'ddp_w2.npz'))
    meta = json.load(open(os.path.join(G, 'ddp_w2.json')))
    regions, ridx = _tables()
    p = 'r%d/' % rank
    out = {}
    # L3
    x = torch.from_numpy(z[p + 'ag/x']).requires_grad_(True)
    y = R.AwesomeAllgather.apply(x)
    (y * torch.from_numpy(z[p + 'ag/coef'])).sum().backward()
    out['ag/y'], out['ag/dx'] = y.detach().numpy(), x.grad.numpy()
    # L1 with cross-rank batch Dice
    valid = meta['valid_regions'][rank]
    for bd in (1, 0):
        logits = [torch.from_numpy(z[p + 'mt/logits%d' % i]).requires_grad_(True) for i in range(2)]
        tg = [torch.from_numpy(z[p + 'mt/target%d' % i]) for i in range(2)]
        l, ce, dc = R.multitalent_loss(logits, tg, valid, regions, ridx, z[p + 'mt/weights'], batch_dice=bool(bd),
                                       gathered=R.gathered_over_ranks)
        l.backward()
        out['mt/bd%d/loss' % bd] = np.array([float(l), float(ce), float(dc)])
        for i in range(2):
            out['mt/bd%d/dlogits%d' % (bd, i)] = logits[i].grad.numpy()
    # L4
    tps, fps, fns, dcs = [], [], [], []
    for it in range(2):
        tp, fp, fn, fg = R.multitalent_online_evaluation(torch.from_numpy(z[p + 'oe/out%d' % it]), torch.from_numpy(z[p + 'oe/target%d' % it]),
                                                         valid, regions, ridx, gathered=R.gathered_over_ranks)
        tps.append(tp); fps.append(fp); fns.append(fn); dcs.append(fg)
    out['oe/tp'], out['oe/fp'], out['oe/fn'], out['oe/foreground_dc'] = np.array(tps), np.array(fps), np.array(fns), np.array(dcs)
    out['oe/metric'] = R.multitalent_finish_online_evaluation(tps, fps, fns)
    # L5 DDP flavour
    for bd in (1, 0):
        sl = [torch.from_numpy(z[p + 'sm/logits%d' % i]).requires_grad_(True) for i in range(2)]
        stg = [torch.from_numpy(z[p + 'sm/target%d' % i]) for i in range(2)]
        l = R.softmax_ddp_loss(sl, stg, z[p + 'mt/weights'], batch_dice=bool(bd), gathered=R.gathered_over_ranks)
        l.backward()
        out['sm/bd%d/loss' % bd] = float(l)
        for i in range(2):
            out['sm/bd%d/dlogits%d' % (bd, i)] = sl[i].grad.numpy()
    return out


def test_oracle_world2_matches_real_reference():
    res = run_world(_oracle_w2)
    z = np.load(os.path.join(G, 'ddp_w2.npz'))
    for r, o in enumerate(res):
        p = 'r%d/' % r
        assert np.array_equal(o['ag/y'], z[p + 'ag/y'])
        assert np.allclose(o['ag/dx'], z[p + 'ag/dx'], atol=1e-6)
        for bd in (1, 0):
            assert np.allclose(o['mt/bd%d/loss' % bd], z[p + 'mt/bd%d/loss' % bd], rtol=1e-5, atol=1e-5)
            for i in range(2):
                assert np.allclose(o['mt/bd%d/dlogits%d' % (bd, i)], z[p + 'mt/bd%d/dlogits%d' % (bd, i)], atol=1e-7)
            assert abs(o['sm/bd%d/loss' % bd] - float(z[p + 'sm/bd%d/loss' % bd])) < 1e-6
            for i in range(2):
                assert np.allclose(o['sm/bd%d/dlogits%d' % (bd, i)], z[p + 'sm/bd%d/dlogits%d' % (bd, i)], atol=1e-7)
        for k in ('tp', 'fp', 'fn'):
            assert np.array_equal(o['oe/' + k], z[p + 'oe/' + k])            # exact integer counts
        assert np.allclose(o['oe/foreground_dc'], z[p + 'oe/foreground_dc'], atol=1e-7)
        assert abs(o['oe/metric'] - float(z[p + 'oe/all_val_eval_metrics'][0])) < 1e-9
    # the Dice part is identical on both ranks (summed over the rank axis), the BCE part is local
    assert abs(res[0]['mt/bd1/loss'][2] - res[1]['mt/bd1/loss'][2]) < 1e-6


def test_batch_split_matches_real_reference_table():
    """set_batch_size_and_oversample of the real class on 2, 4 and 8 gloo ranks (nnUNetTrainerV2_DDP.py:75-117) vs the oracle AND vs
    the product's trainer method."""
    import torch.distributed as dist
    from multitalent_amd.training.network_training.nnUNetTrainer import nnUNetTrainerV2_DDP
    table = json.load(open(os.path.join(G, 'ddp_batch_split.json')))
    n = 0
    for world_s, per_rank in table.items():
        world = int(world_s)
        for rank, rows in enumerate(per_rank):
            for row in rows:
                bs, pct = R.set_batch_size_and_oversample(row['plan_batch'], world, rank, row['dbs'], row['fg'])
                assert bs == row['batch_size'] and abs(pct - row['oversample']) < 1e-12, (world, rank, row)

                class T:
                    pass
                t = T(); t.batch_size = row['plan_batch']; t.oversample_foreground_percent = row['fg']; t.distribute_batch_size = row['dbs']
                orig = (dist.get_world_size, dist.get_rank)
                dist.get_world_size, dist.get_rank = (lambda: world), (lambda: rank)
                try:
                    nnUNetTrainerV2_DDP.set_batch_size_and_oversample(t)
                finally:
                    dist.get_world_size, dist.get_rank = orig
                assert t.batch_size == row['batch_size'] and abs(t.oversample_foreground_percent - row['oversample']) < 1e-12
                assert t.global_batch_size == row['global_batch_size']
                n += 1
    assert n > 60


def test_plain_unet_nonuniform_kernels():
    """Decoder stage u takes conv_kernel_sizes[-(u+1)] (generic_UNet.py:338-339): oracle forward/loss/gradients AND the product
    module's state_dict shapes vs the real reference."""
    from torch import nn
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    z = dict(np.load(os.path.join(G, 'plain_unet_aniso.npz')))
    pools, kernels = z['pools'].tolist(), z['kernels'].tolist()
    sd = {k[4:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in z.items() if k.startswith('sd0/')}
    x = torch.from_numpy(z['x'])
    out = R.generic_unet_forward(sd, x, pools, kernels)
    for i, o in enumerate(out):
        assert np.allclose(o.detach().numpy(), z['out%d' % i], atol=1e-5)
    tg = [torch.from_numpy(z['target%d' % i]) for i in range(3)]
    l = R.multiple_output_loss(out, tg, z['weights'])
    assert abs(float(l) - float(z['loss'])) < 1e-5
    l.backward()
    for k, v in sd.items():
        if 'grad0/' + k in z:
            assert np.allclose(v.grad.numpy(), z['grad0/' + k], atol=2e-5), k
    net = Generic_UNet(1, 6, 3, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in sd.items()}
    assert mine == ref
    assert mine['conv_blocks_localization.0.0.blocks.0.conv.weight'][2:] == (1, 3, 3)
    net.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)


def test_sliding_window_smaller_than_patch():
    """pad_nd_image path (neural_network.py:301): volume smaller than the patch along two axes, odd differences."""
    z = dict(np.load(os.path.join(G, 'sliding_window_pad.npz')))
    sd = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd/')}
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    fwd = lambda t: R.generic_unet_forward(sd, t, pools, kernels, deep_supervision=False)
    for tag in ('a', 'b'):
        for m in (1, 0):
            with torch.no_grad():
                seg, probs = R.predict_3d_tiled(fwd, z[tag + '/vol'], (8, 16, 16), 5, do_mirroring=bool(m), regions_class_order=[3, 1, 4, 2, 5])
            assert probs.shape == z['%s/probs_m%d' % (tag, m)].shape
            assert np.allclose(probs, z['%s/probs_m%d' % (tag, m)], atol=2e-6)
            ref = z['%s/seg_m%d' % (tag, m)]
            safe = (np.abs(probs - 0.5) > 1e-5).all(0)
            assert np.array_equal(seg[safe].astype(np.int16), ref[safe])


def test_multitalent_folds_match_real_reference(tmp_path, monkeypatch):
    """MultiTalent_trainer_ddp.do_split (:433-543): folds 0-4 re-use the per-dataset splits, 5-11 leave one dataset out, missing
    cases are skipped with a warning; the product's trainer method on the same folder layout."""
    from collections import OrderedDict
    from multitalent_amd.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
    ref = json.load(open(os.path.join(G, 'multitalent_splits.json')))
    prep = tmp_path / 'preprocessed'
    for name, splits in ref['per_task_splits'].items():
        (prep / name).mkdir(parents=True)
        sp = [OrderedDict(train=np.array(s['train']), val=np.array(s['val'])) for s in splits]
        pickle.dump(sp, open(prep / name / 'splits_final.pkl', 'wb'))
    ddir = prep / 'Task100_MultiTalent'
    ddir.mkdir()
    monkeypatch.setenv('nnUNet_preprocessed', str(prep))

    class T:
        pass
    for fold_s, exp in ref['folds'].items():
        logs = []
        t = T()
        t.dataset = OrderedDict((k, {'f': k}) for k in ref['keys'])
        t.fold = fold_s if fold_s == 'all' else int(fold_s)
        t.dataset_directory, t.local_rank = str(ddir), 0
        t.print_to_log_file = lambda *a, **k: logs.append(a)
        for m in ('_preprocessed_root', '_task_folder', '_build_custom_splits'):
            setattr(t, m, getattr(MultiTalent_trainer_ddp, m).__get__(t))
        MultiTalent_trainer_ddp.do_split(t)
        assert list(t.dataset_tr.keys()) == exp['train'], fold_s
        assert list(t.dataset_val.keys()) == exp['val'], fold_s
        assert len(logs) == exp['warnings']
    assert os.path.isfile(ddir / 'splits_custom.pkl') and not os.path.isfile(str(ddir / 'splits_custom.pkl') + '.tmp')
