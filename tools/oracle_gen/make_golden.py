"""Generate golden vectors by IMPORTING AND RUNNING THE REAL REFERENCE on CPU (build container only).

Writes small .npz / .json fixtures to tests/golden/.  The fixtures are data (inputs + expected outputs); no
reference source travels.  Re-run:  python tools/oracle_gen/make_golden.py
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import
ref_import.install()

OUT = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden'))
os.makedirs(OUT, exist_ok=True)

from nnunet.network_architecture.generic_UNet import Generic_UNet
from nnunet.network_architecture.generic_modular_residual_UNet import FabiansUNet
from nnunet.network_architecture.generic_modular_UNet import get_default_network_config
from nnunet.network_architecture.initialization import InitWeights_He
from nnunet.training.loss_functions.deep_supervision import MultipleOutputLoss2
from nnunet.training.loss_functions.dice_loss import DC_and_CE_loss
from nnunet.dataset_conversion.Task100_MultiTalent import MultiTalent_valid_regions
from nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
from nnunet.training.network_training.nnUNetTrainerV2_DDP import nnUNetTrainerV2_DDP
from nnunet.network_architecture.neural_network import SegmentationNetwork


def sd_np(net):
    return {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}


def randomize(net, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in net.named_parameters():
        if p.dim() == 1 and ('norm' in n) and n.endswith('weight'):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        elif n.endswith('bias'):
            p.data = 0.2 * torch.randn(p.shape, generator=g)


def blocky_targets(shape, scales, nlabels, B, seed):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, nlabels, (B, 1) + tuple(max(s // 4, 1) for s in shape), generator=g).float()
    full = torch.nn.functional.interpolate(coarse, size=tuple(shape), mode='nearest')
    return [torch.nn.functional.interpolate(full, size=tuple(int(round(s * f)) for s, f in zip(shape, sc)), mode='nearest')
            for sc in scales]


def plain_unet():
    torch.manual_seed(11)
    pools, kernels = [[2, 2, 2], [2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 4
    net = Generic_UNet(1, 6, 4, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    randomize(net, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 1, 8, 16, 16), generator=g)
    scales = [[1, 1, 1], [.5, .5, .5], [.25, .25, .25]]
    tg = blocky_targets((8, 16, 16), scales, 4, 2, 3)
    w = np.array([1 / (2 ** i) for i in range(3)]); w[-1] = 0; w = w / w.sum()
    sd0 = sd_np(net)
    # --- single-GPU trainer loss (MultipleOutputLoss2(DC_and_CE_loss)) + 2 SGD steps exactly like nnUNetTrainerV2.run_iteration
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    rec = {'x': x.numpy(), 'weights': w, 'pools': np.array(pools), 'kernels': np.array(kernels)}
    for i, t in enumerate(tg):
        rec['target%d' % i] = t.numpy()
    for k, v in sd0.items():
        rec['sd0/' + k] = v
    net.train()
    out = net(x)
    for i, o in enumerate(out):
        rec['out%d' % i] = o.detach().numpy()
    losses = []
    for step in range(2):
        opt.zero_grad()
        out = net(x)
        l = loss_fn(out, tg)
        l.backward()
        if step == 0:
            for n, p in net.named_parameters():
                if p.grad is not None:     # head of the zero-weight level gets no gradient (deep_supervision.py:41)
                    rec['grad0/' + n] = p.grad.detach().numpy().copy()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        losses.append(float(l))
    rec['losses'] = np.array(losses)
    for k, v in sd_np(net).items():
        rec['sd2/' + k] = v
    # --- DDP flavour of the softmax loss (nnUNetTrainerV2_DDP.compute_loss), world size 1
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    out = net(x)
    selfobj = SimpleNamespace(batch_dice=True, ds_loss_weights=w, ce_loss=__import__('nnunet.training.loss_functions.crossentropy', fromlist=['x']).RobustCrossEntropyLoss())
    rec['loss_ddp_batchdice'] = np.array(float(nnUNetTrainerV2_DDP.compute_loss(selfobj, out, tg)))
    selfobj.batch_dice = False
    rec['loss_ddp_nobatchdice'] = np.array(float(nnUNetTrainerV2_DDP.compute_loss(selfobj, out, tg)))
    # inference output (do_ds False)
    net.eval(); net.do_ds = False
    with torch.no_grad():
        rec['out_infer'] = net(x).numpy()
    np.savez_compressed(os.path.join(OUT, 'plain_unet.npz'), **rec)
    print('plain_unet: losses', losses)


def resenc_unet():
    torch.manual_seed(12)
    pools = [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    kernels = [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]
    blocks = [1, 2, 2, 2]
    net = FabiansUNet(1, 6, blocks, 2, pools, kernels, get_default_network_config(3, None, norm_type="in"), 47, [1, 1, 1],
                      True, False, 16, InitWeights_He(1e-2))
    randomize(net, 4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 1, 8, 16, 16), generator=g)
    rec = {'x': x.numpy(), 'pools': np.array(pools), 'kernels': np.array(kernels), 'blocks': np.array(blocks)}
    for k, v in sd_np(net).items():
        rec['sd0/' + k] = v
    net.train()
    out = net(x)
    for i, o in enumerate(out):
        rec['out%d' % i] = o.detach().numpy()
    valid = [list(MultiTalent_valid_regions['Task046_AbdOrgSegm2']), list(MultiTalent_valid_regions['Task064_KiTS_labelsFixed'])]
    scales = [[1, 1, 1], [1, .5, .5], [.5, .25, .25]]
    tg = blocky_targets((8, 16, 16), scales, 44, 2, 6)
    w = np.array([0.5, 0.3, 0.2])
    selfobj = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=True, ds_loss_weights=w)
    l, ce, dc = MultiTalent_trainer_ddp.compute_loss(selfobj, out, tg, valid)
    l.backward()
    for i, t in enumerate(tg):
        rec['target%d' % i] = t.numpy()
    rec['weights'] = w
    rec['loss'] = np.array([float(l), float(ce), float(dc)])
    for n, p in net.named_parameters():
        rec['grad0/' + n] = p.grad.detach().numpy().copy()
    json.dump({'valid_regions': valid}, open(os.path.join(OUT, 'resenc_unet_valid.json'), 'w'))
    np.savez_compressed(os.path.join(OUT, 'resenc_unet.npz'), **rec)
    print('resenc_unet: loss', float(l), float(ce), float(dc))


def multitalent_loss():
    g = torch.Generator().manual_seed(21)
    B, C = 3, 47
    shapes = [(6, 12, 12), (3, 6, 6)]
    logits = [(2.0 * torch.randn((B, C) + s, generator=g)).requires_grad_(True) for s in shapes]
    tg = blocky_targets(shapes[0], [[1, 1, 1], [.5, .5, .5]], 48, B, 22)
    names = ['Task003_Liver', 'Task017_AbdominalOrganSegmentation', 'Task018_PelvicOrganSegmentation']
    valid = [list(MultiTalent_valid_regions[n]) for n in names]
    w = np.array([0.75, 0.25])
    rec = {'weights': w}
    for bd in (True, False):
        for t in logits:
            t.grad = None
        selfobj = SimpleNamespace(ce_loss=nn.BCEWithLogitsLoss(), batch_dice=bd, ds_loss_weights=w)
        l, ce, dc = MultiTalent_trainer_ddp.compute_loss(selfobj, logits, tg, valid)
        l.backward()
        key = 'bd1' if bd else 'bd0'
        rec[key + '/loss'] = np.array([float(l), float(ce), float(dc)])
        for i, t in enumerate(logits):
            rec[key + '/dlogits%d' % i] = t.grad.numpy().copy()
    for i, t in enumerate(logits):
        rec['logits%d' % i] = t.detach().numpy()
        rec['target%d' % i] = tg[i].numpy()
    json.dump({'valid_regions': valid, 'datasets': names}, open(os.path.join(OUT, 'multitalent_loss_valid.json'), 'w'))
    np.savez_compressed(os.path.join(OUT, 'multitalent_loss.npz'), **rec)
    print('multitalent_loss:', rec['bd1/loss'], rec['bd0/loss'])


def sliding_window():
    # (1) the reference's own manually verified known answers (tests/test_steps_for_sliding_window_prediction.py:96-163)
    cases = [((128, 128, 128), (146, 176, 148), 0.5), ((128, 128, 128), (424, 456, 456), 0.5), ((64, 192, 192), (94, 308, 308), 0.5),
             ((128, 128, 128), (128, 128, 128), 0.5), ((48, 192, 192), (512, 512, 512), 0.5), ((96, 192, 192), (512, 512, 512), 0.5),
             ((30, 224, 224), (30, 224, 224), 1), ((30, 224, 224), (37, 251, 291), 0.125), ((16, 32, 32), (40, 72, 72), 0.5)]
    table = []
    for patch, img, step in cases:
        table.append({'patch': patch, 'image': img, 'step': step,
                      'steps': SegmentationNetwork._compute_steps_for_sliding_window(patch, img, step)})
    json.dump(table, open(os.path.join(OUT, 'sliding_window_steps.json'), 'w'))
    rec = {'gaussian_16_32_32': SegmentationNetwork._get_gaussian((16, 32, 32), 1. / 8),
           'gaussian_48_192_192_slice': SegmentationNetwork._get_gaussian((48, 192, 192), 1. / 8)[24, 96, :].copy(),
           'gaussian_48_192_192_minmax': np.array([SegmentationNetwork._get_gaussian((48, 192, 192), 1. / 8).min(), 1.0])}
    # (2) predict_3D of the real reference on a small volume with a small network (sigmoid, regions_class_order) and softmax/argmax
    torch.manual_seed(31)
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    for tag, nc, nonlin, order in (('mt', 5, nn.Sigmoid(), [3, 1, 4, 2, 5]), ('sm', 3, lambda x: torch.softmax(x, 1), None)):
        net = Generic_UNet(1, 6, nc, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
        randomize(net, 32)
        net.inference_apply_nonlin = nonlin
        net.eval(); net.do_ds = False
        g = torch.Generator().manual_seed(33)
        vol = torch.randn((1, 20, 40, 44), generator=g).numpy()
        for mirror in (True, False):
            seg, probs = net.predict_3D(vol, do_mirroring=mirror, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5,
                                        patch_size=(8, 16, 16), regions_class_order=order, use_gaussian=True,
                                        pad_border_mode='constant', pad_kwargs={'constant_values': 0}, all_in_gpu=False,
                                        verbose=False, mixed_precision=False)
            rec['%s/seg_m%d' % (tag, int(mirror))] = seg.astype(np.int16)
            rec['%s/probs_m%d' % (tag, int(mirror))] = probs.astype(np.float32)
        rec[tag + '/vol'] = vol
        for k, v in sd_np(net).items():
            rec[tag + '/sd/' + k] = v
    # volume smaller than the patch in one axis (exercises pad_nd_image; parity unpinned at that third-party boundary)
    np.savez_compressed(os.path.join(OUT, 'sliding_window.npz'), **rec)
    print('sliding_window done')


if __name__ == '__main__':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('gloo', rank=0, world_size=1)
    torch.set_num_threads(8)
    plain_unet(); resenc_unet(); multitalent_loss(); sliding_window()
    dist.destroy_process_group()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
