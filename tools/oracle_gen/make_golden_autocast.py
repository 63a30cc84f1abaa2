"""Pin the MIXED-PRECISION mode to the reference: run the IMPORTED reference networks + losses on CPU under
`torch.autocast('cpu', dtype=torch.float16)` — what `run_iteration` does with `fp16=True` (nnUNetTrainerV2.py:249-262,
MultiTalent_Trainer_DDP.py:340-352: forward AND compute_loss inside autocast, `amp_grad_scaler.scale(l).backward()`) — and, as a second
variant, under bfloat16 autocast.  Build container only; nothing here travels except the fixtures it writes.

  python tools/oracle_gen/make_golden_autocast.py                 -> tests/golden/autocast.npz   (toy networks of plain_unet.npz / resenc_unet.npz:
                                                                      logits, loss, gradients per variant + the fp64 gradient)
  python tools/oracle_gen/make_golden_autocast.py --fullsize resenc|task009|task100
                                                                   -> tests/golden/autocast_fullsize_<net>.json (the reference-autocast's OWN
                                                                      deviation from its fp32 logits / the fp64 gradient at 48x192x192, B = 1,
                                                                      same seeds as tests/test_fullsize_oracle_gpu.py) + a copy under profiles/

The loss-scaling of the reference's GradScaler (initial scale 65536, halved when a gradient overflows; network_trainer.py:400-402) is
reproduced by hand: scale the loss, backward, unscale, halve and retry on inf / nan — the scale that succeeded is recorded."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
import ref_import
ref_import.install()
sys.path.insert(1, ROOT)

OUT = os.path.join(ROOT, 'tests', 'golden')

from nnunet.network_architecture.generic_UNet import Generic_UNet
from nnunet.network_architecture.generic_modular_residual_UNet import FabiansUNet
from nnunet.network_architecture.generic_modular_UNet import get_default_network_config
from nnunet.network_architecture.initialization import InitWeights_He
from nnunet.training.loss_functions.deep_supervision import MultipleOutputLoss2
from nnunet.training.loss_functions.dice_loss import DC_and_CE_loss
from nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp

VARIANTS = {'fp16': torch.float16, 'bf16': torch.bfloat16}


def run_variant(net, x, loss_of_outputs, variant):
    """variant: 'fp32' | 'fp64' | 'fp16' | 'bf16'.  Returns (logits [float32 numpy], loss values, {name: grad float64 numpy}, scale)."""
    for p in net.parameters():
        p.grad = None
    scale = 1.0
    if variant in ('fp32', 'fp64'):
        dt = torch.float32 if variant == 'fp32' else torch.float64
        net.to(dt)
        out = net(x.to(dt))
        res = loss_of_outputs(out)
        (res[0] if isinstance(res, (tuple, list)) else res).backward()
        grads = {n: (p.grad.double().numpy().copy() if p.grad is not None else None) for n, p in net.named_parameters()}
        net.to(torch.float32)
    else:
        scale = 65536.0
        while True:
            for p in net.parameters():
                p.grad = None
            with torch.autocast('cpu', dtype=VARIANTS[variant]):
                out = net(x)
                res = loss_of_outputs(out)
            l = res[0] if isinstance(res, (tuple, list)) else res
            (l * scale).backward()
            ok = all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
            if ok or scale < 1.0:
                break
            scale /= 2.0                   # GradScaler: skip the step, backoff_factor 0.5
        grads = {n: (p.grad.double().numpy().copy() / scale if p.grad is not None else None) for n, p in net.named_parameters()}
    vals = [float(r.detach().float()) for r in res] if isinstance(res, (tuple, list)) else [float(res.detach().float())]
    return [o.detach().float().numpy() for o in out], vals, grads, scale


def flat(g, names):
    return np.concatenate([(g[n] if g[n] is not None else np.zeros(1)).reshape(-1) for n in names])


def deviation(logits, loss, grads, l32, loss32, g64, names):
    ga, gt = flat(grads, names), flat(g64, names)
    big = [n for n in names if n.endswith('.weight') and g64[n] is not None and g64[n].ndim == 5 and g64[n].size > 50000]
    worst = min([float((grads[n] * g64[n]).sum() / (np.linalg.norm(grads[n]) * np.linalg.norm(g64[n]) + 1e-300)) for n in big], default=None)
    return {'logits_rel_l2_vs_fp32': [float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)) for a, b in zip(logits, l32)],
            'logits_max_err_over_max_vs_fp32': [float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(logits, l32)],
            'loss': loss, 'loss_abs_err_vs_fp32': [abs(a - b) for a, b in zip(loss, loss32)],
            'grad_rel_l2_vs_fp64': float(np.linalg.norm(ga - gt) / np.linalg.norm(gt)),
            'grad_cos_vs_fp64': float((ga * gt).sum() / (np.linalg.norm(ga) * np.linalg.norm(gt))),
            'worst_large_conv_weight_cos_vs_fp64': worst}


def toy():
    rec = {}
    # ---- plain Generic_UNet, softmax Dice + CE (nnUNetTrainerV2.run_iteration) on the inputs of plain_unet.npz
    z = np.load(os.path.join(OUT, 'plain_unet.npz'))
    pools, kernels = z['pools'].tolist(), z['kernels'].tolist()
    net = Generic_UNet(1, 6, 4, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')})
    net.train()
    x = torch.from_numpy(z['x'])
    tg = [torch.from_numpy(z['target%d' % i]) for i in range(3)]
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), z['weights'])
    summary = {}
    summary['plain'] = record(rec, 'plain', net, x, lambda out: loss_fn(out, tg))
    # ---- residual encoder FabiansUNet, MultiTalent BCE + Dice (MultiTalent_trainer_ddp.compute_loss) on the inputs of resenc_unet.npz
    z = np.load(os.path.join(OUT, 'resenc_unet.npz'))
    valid = json.load(open(os.path.join(OUT, 'resenc_unet_valid.json')))['valid_regions']
    net = FabiansUNet(1, 6, z['blocks'].tolist(), 2, z['pools'].tolist(), z['kernels'].tolist(), get_default_network_config(3, None, norm_type="in"),
                      47, [1, 1, 1], True, False, 16, InitWeights_He(1e-2))
    net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')})
    net.train()
    x = torch.from_numpy(z['x'])
    tg = [torch.from_numpy(z['target%d' % i]) for i in range(3)]
    selfobj = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=True, ds_loss_weights=z['weights'])
    summary['resenc'] = record(rec, 'resenc', net, x, lambda out: MultiTalent_trainer_ddp.compute_loss(selfobj, out, tg, valid))
    np.savez_compressed(os.path.join(OUT, 'autocast.npz'), **rec)
    json.dump(summary, open(os.path.join(OUT, 'autocast_summary.json'), 'w'), indent=1)
    print(json.dumps(summary, indent=1))


def record(rec, tag, net, x, loss_of_outputs):
    names = [n for n, _ in net.named_parameters()]
    l32, loss32, g32, _ = run_variant(net, x, loss_of_outputs, 'fp32')
    _, _, g64, _ = run_variant(net, x, loss_of_outputs, 'fp64')
    for n in names:
        if g64[n] is not None:
            rec['%s/fp64/grad/%s' % (tag, n)] = g64[n].astype(np.float32)      # 6e-8 relative: far below any deviation measured against it
    for i, o in enumerate(l32):
        rec['%s/fp32/out%d' % (tag, i)] = o
    rec['%s/fp32/loss' % tag] = np.array(loss32)
    summ = {'fp32': deviation(l32, loss32, g32, l32, loss32, g64, names)}
    for v in VARIANTS:
        lg, ls, g, scale = run_variant(net, x, loss_of_outputs, v)
        for i, o in enumerate(lg):
            rec['%s/%s/out%d' % (tag, v, i)] = o
        rec['%s/%s/loss' % (tag, v)] = np.array(ls)
        rec['%s/%s/scale' % (tag, v)] = np.array(scale)
        for n in names:
            if g[n] is not None:
                rec['%s/%s/grad/%s' % (tag, v, n)] = g[n].astype(np.float32)
        summ[v] = deviation(lg, ls, g, l32, loss32, g64, names)
        summ[v]['grad_scale'] = scale
    return summ


def fullsize(which, variants):
    import bench
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.ds_weights import ds_loss_weights
    PATCH = (48, 192, 192)
    dev = torch.device('cpu')
    # the same seeds / inputs as tests/test_fullsize_oracle_gpu.py (the product's parameter holders give the initial weights: identical
    # state_dict keys, loaded strict into the REFERENCE's modules below)
    if which == 'resenc':
        torch.manual_seed(99)
        sd0 = bench.build_network('resenc').state_dict()
        net = FabiansUNet(1, 30, bench.RESENC_BLOCKS, 2, bench.RESENC_POOLS, bench.RESENC_KERNELS, get_default_network_config(3, None, norm_type="in"),
                          47, [1] * (len(bench.RESENC_POOLS) - 1), True, False, 320, InitWeights_He(1e-2))
        valid = [MultiTalent_valid_regions['Task064_KiTS_labelsFixed']]
        label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
        x = synthetic_ct(1, PATCH, 79, dev)
        tg = synthetic_targets(1, PATCH, ds_scales(bench.RESENC_POOLS, skip_first=True), label_sets, 79, dev)
        w = ds_loss_weights(len(bench.RESENC_POOLS))
        selfobj = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=True, ds_loss_weights=w)
        lo = lambda out: MultiTalent_trainer_ddp.compute_loss(selfobj, out, tg, valid)
    else:
        nc = 2 if which == 'task009' else 47
        torch.manual_seed(1234 if which == 'task009' else 4321)
        sd0 = bench.build_network(which).state_dict()
        net = Generic_UNet(1, 30, nc, len(bench.POOLS), 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, InitWeights_He(1e-2), bench.POOLS, bench.KERNELS, False, True, True)
        w = ds_loss_weights(len(bench.POOLS))
        if which == 'task009':
            x = synthetic_ct(1, PATCH, 77, dev)
            tg = synthetic_targets(1, PATCH, ds_scales(bench.POOLS), [[1]], 77, dev)
            loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)
            lo = lambda out: loss_fn(out, tg)
        else:
            valid = [MultiTalent_valid_regions['Task046_AbdOrgSegm2']]
            label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
            x = synthetic_ct(1, PATCH, 78, dev)
            tg = synthetic_targets(1, PATCH, ds_scales(bench.POOLS), label_sets, 78, dev)
            selfobj = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=True, ds_loss_weights=w)
            lo = lambda out: MultiTalent_trainer_ddp.compute_loss(selfobj, out, tg, valid)
    net.load_state_dict(sd0, strict=True)
    net.train()
    names = [n for n, _ in net.named_parameters()]
    res = {'network': which, 'patch': list(PATCH), 'batch': 1, 'torch': torch.__version__, 'threads': torch.get_num_threads(),
           'what': "the imported reference network + loss on CPU; deviation of each arithmetic from the fp32 logits / loss and the fp64 gradient"}
    path = os.path.join(OUT, 'autocast_fullsize_%s.json' % which)
    t = time.time()
    l32, loss32, g32, _ = run_variant(net, x, lo, 'fp32')
    print('fp32 done %.0f s' % (time.time() - t), flush=True); t = time.time()
    _, _, g64, _ = run_variant(net, x, lo, 'fp64')
    print('fp64 done %.0f s' % (time.time() - t), flush=True)
    res['fp32'] = deviation(l32, loss32, g32, l32, loss32, g64, names)
    json.dump(res, open(path, 'w'), indent=1)
    for v in variants:
        t = time.time()
        lg, ls, g, scale = run_variant(net, x, lo, v)
        res[v] = deviation(lg, ls, g, l32, loss32, g64, names)
        res[v]['grad_scale'] = scale
        res[v]['seconds'] = round(time.time() - t)
        print(v, json.dumps(res[v]), flush=True)
        json.dump(res, open(path, 'w'), indent=1)
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'r05_reference_autocast_fullsize_%s.json' % which), 'w'), indent=1)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--fullsize', default=None, choices=['resenc', 'task009', 'task100'])
    ap.add_argument('--variants', nargs='+', default=['bf16', 'fp16'])
    a = ap.parse_args()
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541' if a.fullsize is None else str(29542 + len(a.fullsize)))
    dist.init_process_group('gloo', rank=0, world_size=1)        # compute_loss all_gathers its Dice statistics (world size 1)
    if a.fullsize:
        fullsize(a.fullsize, a.variants)
    else:
        toy()
