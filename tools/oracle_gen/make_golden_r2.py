"""Round-2 golden vectors from the REAL reference (build container only):
  * plain_unet_aniso.npz   Generic_UNet with NON-UNIFORM conv_kernel_sizes ([1,3,3] first stage): the decoder's
                           conv_kernel_sizes[-(u+1)] indexing (generic_UNet.py:338-339) fixes the weight shapes
  * sliding_window_pad.npz predict_3D of a volume SMALLER than the patch in two axes (pad_nd_image path, neural_network.py:301)
  * multitalent_splits.json MultiTalent_trainer_ddp.do_split (:433-543): folds 0-11 from per-dataset splits_final.pkl files
Re-run:  python tools/oracle_gen/make_golden_r2.py
"""
import json
import os
import pickle
import sys
import tempfile
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import
ref_import.install()

OUT = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden'))

from nnunet.network_architecture.generic_UNet import Generic_UNet
from nnunet.network_architecture.initialization import InitWeights_He
from nnunet.training.loss_functions.deep_supervision import MultipleOutputLoss2
from nnunet.training.loss_functions.dice_loss import DC_and_CE_loss
from make_golden import blocky_targets, randomize, sd_np


def plain_unet_aniso():
    torch.manual_seed(41)
    pools = [[1, 2, 2], [2, 2, 2], [2, 2, 2]]
    kernels = [[1, 3, 3], [3, 3, 3], [3, 3, 3], [1, 3, 3]]       # decoder stage u uses kernels[-(u+1)]: [1,3,3], [3,3,3], [3,3,3]
    net = Generic_UNet(1, 6, 3, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    randomize(net, 42)
    g = torch.Generator().manual_seed(43)
    x = torch.randn((2, 1, 8, 16, 16), generator=g)
    scales = [[1, 1, 1], [1, .5, .5], [.5, .25, .25]]
    tg = blocky_targets((8, 16, 16), scales, 3, 2, 44)
    w = np.array([4 / 7, 2 / 7, 1 / 7])
    rec = {'x': x.numpy(), 'weights': w, 'pools': np.array(pools), 'kernels': np.array(kernels)}
    for k, v in sd_np(net).items():
        rec['sd0/' + k] = v
    for i, t in enumerate(tg):
        rec['target%d' % i] = t.numpy()
    net.train()
    out = net(x)
    for i, o in enumerate(out):
        rec['out%d' % i] = o.detach().numpy()
    l = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)(out, tg)
    l.backward()
    rec['loss'] = np.array(float(l))
    for n, p in net.named_parameters():
        rec['grad0/' + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'plain_unet_aniso.npz'), **rec)
    print('plain_unet_aniso: loss', float(l), 'loc0 conv weight', tuple(net.conv_blocks_localization[0][0].blocks[0].conv.weight.shape),
          'loc2', tuple(net.conv_blocks_localization[2][0].blocks[0].conv.weight.shape))


def sliding_window_pad():
    torch.manual_seed(51)
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    rec = {}
    net = Generic_UNet(1, 6, 5, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    randomize(net, 52)
    net.inference_apply_nonlin = nn.Sigmoid()
    net.eval(); net.do_ds = False
    for k, v in sd_np(net).items():
        rec['sd/' + k] = v
    # (5, 40, 13): smaller than the patch (8, 16, 16) along axes 0 (odd difference 3 -> 1 below, 2 above) and 2 (3 -> 1, 2)
    for tag, shape in (('a', (5, 40, 13)), ('b', (8, 10, 33))):
        g = torch.Generator().manual_seed(53)
        vol = torch.randn((1,) + shape, generator=g).numpy()
        for mirror in (True, False):
            seg, probs = net.predict_3D(vol, do_mirroring=mirror, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5,
                                        patch_size=(8, 16, 16), regions_class_order=[3, 1, 4, 2, 5], use_gaussian=True,
                                        pad_border_mode='constant', pad_kwargs={'constant_values': 0}, all_in_gpu=False,
                                        verbose=False, mixed_precision=False)
            assert seg.shape == shape and probs.shape == (5,) + shape
            rec['%s/seg_m%d' % (tag, int(mirror))] = seg.astype(np.int16)
            rec['%s/probs_m%d' % (tag, int(mirror))] = probs.astype(np.float32)
        rec[tag + '/vol'] = vol
    np.savez_compressed(os.path.join(OUT, 'sliding_window_pad.npz'), **rec)
    print('sliding_window_pad done')


def multitalent_splits():
    import nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP as M
    rs = np.random.RandomState(7)
    tasks = {3: 'Task003_Liver', 7: 'Task007_Pancreas', 8: 'Task008_HepaticVessel', 10: 'Task010_Colon', 17: 'Task017_AbdominalOrganSegmentation',
             55: 'Task055_SegTHOR', 64: 'Task064_KiTS_labelsFixed'}
    with tempfile.TemporaryDirectory() as tmp:
        prep = os.path.join(tmp, 'preprocessed')
        case_ids = {}
        for tid, name in tasks.items():
            os.makedirs(os.path.join(prep, name))
            ids = ['img%04d' % i if tid == 17 else '%s_%03d' % (name[8:11].lower(), i) for i in range(1, 12)]
            case_ids[tid] = ids
            perm = rs.permutation(len(ids))
            splits = []
            for f in range(5):
                val = [ids[i] for i in perm[f::5]]
                splits.append(OrderedDict(train=np.array([i for i in ids if i not in val]), val=np.array(val)))
            pickle.dump(splits, open(os.path.join(prep, name, 'splits_final.pkl'), 'wb'))
        keys = []
        for tid, ids in case_ids.items():
            keys += ['%03d_%s' % (tid, i) for i in ids]
        # Task046: Task017's images (train AND "test") plus its own PAN cases (reference comment :456-461)
        keys += ['046_%s' % i for i in case_ids[17]] + ['046_img%04d' % i for i in (61, 62)] + ['046_PANCREAS_%04d' % i for i in range(1, 10)]
        keys.remove('003_liv_004')              # a case listed in a per-dataset split but absent from the preprocessed folder
        dataset = OrderedDict((k, {'f': k}) for k in sorted(keys))
        ddir = os.path.join(tmp, 'Task100_MultiTalent')
        os.makedirs(ddir)
        M.preprocessing_output_dir = prep
        M.convert_id_to_task_name = lambda t: tasks[int(t)]
        M.save_pickle = lambda obj, f, mode='wb': pickle.dump(obj, open(f, mode))          # batchgenerators' helper (third party, absent)
        res = {'keys': sorted(keys), 'tasks': {str(k): v for k, v in tasks.items()}, 'per_task_splits': {}, 'folds': {}}
        for tid, name in tasks.items():
            sp = pickle.load(open(os.path.join(prep, name, 'splits_final.pkl'), 'rb'))
            res['per_task_splits'][name] = [{'train': [str(i) for i in s['train']], 'val': [str(i) for i in s['val']]} for s in sp]
        for fold in list(range(12)) + ['all']:
            logs = []
            so = SimpleNamespace(dataset=dataset, fold=fold, dataset_directory=ddir, local_rank=0,
                                 print_to_log_file=lambda *a, **k: logs.append(' '.join(str(i) for i in a)))
            M.MultiTalent_trainer_ddp.do_split(so)
            res['folds'][str(fold)] = {'train': list(so.dataset_tr.keys()), 'val': list(so.dataset_val.keys()), 'warnings': len(logs)}
    json.dump(res, open(os.path.join(OUT, 'multitalent_splits.json'), 'w'))
    print('multitalent_splits: fold0 train/val', len(res['folds']['0']['train']), len(res['folds']['0']['val']),
          'fold5', len(res['folds']['5']['train']), 'warnings fold0', res['folds']['0']['warnings'])


if __name__ == '__main__':
    torch.set_num_threads(8)
    plain_unet_aniso(); sliding_window_pad(); multitalent_splits()
    for f in ('plain_unet_aniso.npz', 'sliding_window_pad.npz', 'multitalent_splits.json'):
        print(f, os.path.getsize(os.path.join(OUT, f)))
