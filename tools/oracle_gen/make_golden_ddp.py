"""World-size > 1 golden vectors from the REAL reference (build container only): W CPU processes under gloo run the reference's own
awesome_allgather_function, MultiTalent_trainer_ddp.compute_loss / run_online_evaluation / finish_online_evaluation,
nnUNetTrainerV2_DDP.compute_loss / set_batch_size_and_oversample and two full DDP training iterations (torch DDP around the
reference's Generic_UNet, clip 12, SGD-Nesterov), and record every rank's inputs and outputs.

Writes tests/golden/ddp_w2.npz + ddp_w2.json (+ ddp_batch_split.json).  Re-run:  python tools/oracle_gen/make_golden_ddp.py
"""
import json
import os
import pickle
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden'))
DATASETS = [['Task003_Liver', 'Task017_AbdominalOrganSegmentation'], ['Task064_KiTS_labelsFixed', 'Task009_Spleen']]   # [rank][b]


def blocky(shape, nlabels, B, seed, labels=None):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, nlabels, (B, 1) + tuple(max(s // 4, 1) for s in shape), generator=g)
    if labels is not None:
        lab = torch.tensor(labels)
        coarse = lab[coarse % len(lab)]
    return torch.nn.functional.interpolate(coarse.float(), size=tuple(shape), mode='nearest')


def pyramid(full, scales):
    shape = full.shape[2:]
    return [torch.nn.functional.interpolate(full, size=tuple(int(round(s * f)) for s, f in zip(shape, sc)), mode='nearest') for sc in scales]


def worker(rank, world, port, tmp):
    sys.path.insert(0, HERE)
    import ref_import
    ref_import.install()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(4)
    from nnunet.utilities.distributed import awesome_allgather_function
    from nnunet.dataset_conversion.Task100_MultiTalent import MultiTalent_valid_regions, MultiTalent_regions
    from nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
    from nnunet.training.network_training.nnUNetTrainerV2_DDP import nnUNetTrainerV2_DDP
    from nnunet.training.loss_functions.crossentropy import RobustCrossEntropyLoss
    from nnunet.network_architecture.generic_UNet import Generic_UNet
    from nnunet.network_architecture.initialization import InitWeights_He
    from torch.nn.parallel import DistributedDataParallel as DDP
    rec = {}

    # ---- (a) awesome_allgather_function (utilities/distributed.py:28-73): forward = stacked gather, backward = all-reduced slice
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn((3, 5), generator=g).requires_grad_(True)
    coef = torch.randn((world, 3, 5), generator=torch.Generator().manual_seed(200 + rank))     # each rank has its own downstream loss
    y = awesome_allgather_function.apply(x)
    (y * coef).sum().backward()
    rec['ag/x'], rec['ag/coef'], rec['ag/y'], rec['ag/dx'] = x.detach().numpy(), coef.numpy(), y.detach().numpy(), x.grad.numpy().copy()

    # ---- (b) MultiTalent compute_loss with cross-rank batch Dice (MultiTalent_Trainer_DDP.py:544-623)
    B, C = 2, 47
    shapes = [(6, 12, 12), (3, 6, 6)]
    g = torch.Generator().manual_seed(300 + rank)
    logits = [(2.0 * torch.randn((B, C) + s, generator=g)).requires_grad_(True) for s in shapes]
    valid = [list(MultiTalent_valid_regions[n]) for n in DATASETS[rank]]
    tg = pyramid(blocky(shapes[0], 48, B, 400 + rank), [[1, 1, 1], [.5, .5, .5]])
    w = np.array([0.75, 0.25])
    for bd in (True, False):
        for t in logits:
            t.grad = None
        so = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=bd, ds_loss_weights=w)
        l, ce, dc = MultiTalent_trainer_ddp.compute_loss(so, logits, tg, valid)
        l.backward()
        key = 'mt/bd%d' % int(bd)
        rec[key + '/loss'] = np.array([float(l), float(ce), float(dc)])
        for i, t in enumerate(logits):
            rec[key + '/dlogits%d' % i] = t.grad.numpy().copy()
    for i, t in enumerate(logits):
        rec['mt/logits%d' % i], rec['mt/target%d' % i] = t.detach().numpy(), tg[i].numpy()
    rec['mt/weights'] = w

    # ---- (c) run_online_evaluation x2 + finish_online_evaluation (:372-431), incl. the gather over ranks
    logs = []
    so = SimpleNamespace(online_eval_foreground_dc=[], online_eval_tp=[], online_eval_fp=[], online_eval_fn=[], all_val_eval_metrics=[],
                         print_to_log_file=lambda *a, **k: logs.append(a))
    for it in range(2):
        g = torch.Generator().manual_seed(500 + 10 * it + rank)
        out = [2.0 * torch.randn((B, C) + shapes[0], generator=g)]
        t_it = [blocky(shapes[0], 48, B, 600 + 10 * it + rank)]
        MultiTalent_trainer_ddp.run_online_evaluation(so, out, t_it, valid)
        rec['oe/out%d' % it], rec['oe/target%d' % it] = out[0].numpy(), t_it[0].numpy()
    rec['oe/tp'] = np.array(so.online_eval_tp, dtype=np.float64)          # [iterations, B, C]  (sum over the RANK axis only)
    rec['oe/fp'] = np.array(so.online_eval_fp, dtype=np.float64)
    rec['oe/fn'] = np.array(so.online_eval_fn, dtype=np.float64)
    rec['oe/foreground_dc'] = np.array(so.online_eval_foreground_dc, dtype=np.float64)     # [iterations, W, B, C]
    MultiTalent_trainer_ddp.finish_online_evaluation(so)
    rec['oe/all_val_eval_metrics'] = np.array(so.all_val_eval_metrics, dtype=np.float64)

    # ---- (d) nnUNetTrainerV2_DDP.compute_loss (softmax Dice + CE, :249-282) with and without cross-rank batch Dice
    g = torch.Generator().manual_seed(700 + rank)
    sl = [(1.5 * torch.randn((B, 4) + s, generator=g)).requires_grad_(True) for s in shapes]
    stg = pyramid(blocky(shapes[0], 4, B, 800 + rank), [[1, 1, 1], [.5, .5, .5]])
    for bd in (True, False):
        for t in sl:
            t.grad = None
        so = SimpleNamespace(batch_dice=bd, ds_loss_weights=w, ce_loss=RobustCrossEntropyLoss())
        l = nnUNetTrainerV2_DDP.compute_loss(so, sl, stg)
        l.backward()
        key = 'sm/bd%d' % int(bd)
        rec[key + '/loss'] = np.array(float(l))
        for i, t in enumerate(sl):
            rec[key + '/dlogits%d' % i] = t.grad.numpy().copy()
    for i, t in enumerate(sl):
        rec['sm/logits%d' % i], rec['sm/target%d' % i] = t.detach().numpy(), stg[i].numpy()

    # ---- (e) two full DDP training iterations (MultiTalent run_iteration :324-370, fp32): torch DDP (gradient mean over ranks)
    #          around the reference network, batch Dice gathered over ranks, clip 12, SGD-Nesterov
    torch.manual_seed(11)
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    net = Generic_UNet(1, 6, 47, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    gp = torch.Generator().manual_seed(12)
    for n, p in net.named_parameters():
        if p.dim() == 1 and 'norm' in n and n.endswith('weight'):
            p.data = 0.5 + torch.rand(p.shape, generator=gp)
        elif n.endswith('bias'):
            p.data = 0.2 * torch.randn(p.shape, generator=gp)
    for k, v in net.state_dict().items():
        rec['train/sd0/' + k] = v.detach().numpy().copy()
    ddp = DDP(net)
    opt = torch.optim.SGD(ddp.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    dsw = np.array([1.0, 0.5]); dsw[-1] = 0; dsw = dsw / dsw.sum()
    dsw = np.array([2 / 3, 1 / 3])          # both levels active so every head receives gradient
    so = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=True, ds_loss_weights=dsw)
    gx = torch.Generator().manual_seed(900 + rank)
    data = torch.randn((B, 1, 8, 16, 16), generator=gx)
    labels = sorted({l for r in valid for reg in r for l in MultiTalent_regions[reg]} | {0})
    ttg = pyramid(blocky((8, 16, 16), 48, B, 950 + rank, labels=labels), [[1, 1, 1], [.5, .5, .5]])
    losses = []
    ddp.train()
    for it in range(2):
        opt.zero_grad()
        out = ddp(data)
        l, ce, dc = MultiTalent_trainer_ddp.compute_loss(so, out, ttg, valid)
        l.backward()
        if it == 0:
            for n, p in net.named_parameters():
                rec['train/grad0/' + n] = p.grad.detach().numpy().copy()            # averaged over ranks by DDP
        torch.nn.utils.clip_grad_norm_(ddp.parameters(), 12)
        opt.step()
        losses.append([float(l), float(ce), float(dc)])
    rec['train/losses'] = np.array(losses)
    rec['train/x'] = data.numpy()
    for i, t in enumerate(ttg):
        rec['train/target%d' % i] = t.numpy()
    rec['train/weights'] = dsw
    rec['train/pools'], rec['train/kernels'] = np.array(pools), np.array(kernels)
    for k, v in net.state_dict().items():
        rec['train/sd2/' + k] = v.detach().numpy().copy()

    with open(os.path.join(tmp, 'rank%d.pkl' % rank), 'wb') as f:
        pickle.dump({'rec': rec, 'valid': valid}, f)
    dist.barrier()
    dist.destroy_process_group()


def split_worker(rank, world, port, tmp):
    """set_batch_size_and_oversample (nnUNetTrainerV2_DDP.py:75-117) of the real class on `world` real gloo ranks."""
    sys.path.insert(0, HERE)
    import ref_import
    ref_import.install()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nnunet.training.network_training.nnUNetTrainerV2_DDP import nnUNetTrainerV2_DDP
    res = []
    for bs, dbs, fg in ((2, False, 0.33), (4, False, 0.33), (4, True, 0.33), (8, True, 0.33), (9, True, 0.33), (16, True, 0.5),
                        (12, True, 0.1), (8, False, 1.0), (8, True, 0.0)):
        if dbs and int(np.ceil(bs / world)) * (world - 1) >= bs:
            continue                          # the reference would hand a rank a batch size <= 0
        so = SimpleNamespace(batch_size=bs, distribute_batch_size=dbs, oversample_foreground_percent=fg, global_batch_size=None)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            nnUNetTrainerV2_DDP.set_batch_size_and_oversample(so)
        res.append({'plan_batch': bs, 'dbs': dbs, 'fg': fg, 'batch_size': int(so.batch_size),
                    'oversample': float(so.oversample_foreground_percent), 'global_batch_size': int(so.global_batch_size)})
    with open(os.path.join(tmp, 'split_w%d_r%d.json' % (world, rank)), 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        W = 2
        mp.spawn(worker, args=(W, 29541, tmp), nprocs=W, join=True)
        merged, valid = {}, []
        for r in range(W):
            d = pickle.load(open(os.path.join(tmp, 'rank%d.pkl' % r), 'rb'))
            valid.append(d['valid'])
            for k, v in d['rec'].items():
                merged['r%d/%s' % (r, k)] = v
        np.savez_compressed(os.path.join(OUT, 'ddp_w2.npz'), **merged)
        json.dump({'world': W, 'datasets': DATASETS, 'valid_regions': valid}, open(os.path.join(OUT, 'ddp_w2.json'), 'w'))
        table = {}
        for i, world in enumerate((2, 4, 8)):
            mp.spawn(split_worker, args=(world, 29551 + i, tmp), nprocs=world, join=True)
            table[str(world)] = [json.load(open(os.path.join(tmp, 'split_w%d_r%d.json' % (world, r)))) for r in range(world)]
        json.dump(table, open(os.path.join(OUT, 'ddp_batch_split.json'), 'w'))
    for f in ('ddp_w2.npz', 'ddp_w2.json', 'ddp_batch_split.json'):
        print(f, os.path.getsize(os.path.join(OUT, f)))
    z = np.load(os.path.join(OUT, 'ddp_w2.npz'))
    print('train losses r0', z['r0/train/losses'], 'r1', z['r1/train/losses'])
    print('mt bd1 r0', z['r0/mt/bd1/loss'], 'r1', z['r1/mt/bd1/loss'], 'oe metric', z['r0/oe/all_val_eval_metrics'])
