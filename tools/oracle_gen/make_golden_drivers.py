"""Golden vectors for the two driver tails: the REAL reference's `MultiTalent_trainer_ddp.validate`
(custom_trainers/MultiTalent/MultiTalent/MultiTalent_Trainer_DDP.py:129-322) and `predict_cases`
(inference/predict_MultiTalent.py:127-266, with `load_model_and_checkpoint_files`, `preprocess_multithreaded`,
`GenericPreprocessor.preprocess_test_case` and `save_segmentation_nifti_from_softmax` underneath) run on CPU in the build
container on small synthetic cases.  Substitutions for what this image lacks, all at third-party seams:

  * SimpleITK -> a shim over multitalent_amd.utilities.nifti_io (real .nii.gz files on disk, read and written by it);
  * skimage.transform.resize -> its own delegate scipy.ndimage.zoom(mode='nearest', grid_mode=True) (like make_golden_export /
    make_golden_preprocess);
  * torch DDP (needs a GPU under nccl) -> a transparent wrapper with `.module`; `init_process_group('nccl')` -> the gloo group
    this script opened; the evaluation (`aggregate_scores`, SimpleITK + medpy) is not part of the golden.

Writes tests/golden/drivers.npz: raw CT volumes + geometry, preprocessed validation cases + properties, the network's
state_dict, and every mask the reference wrote.  Run: python tools/oracle_gen/make_golden_drivers.py"""
import os
import pickle
import shutil
import sys
import tempfile
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import
ref_import.install()

from multitalent_amd.utilities import nifti_io                                    # the shim's backend (file format only)

import batchgenerators.utilities.file_and_folder_operations as ffo
import json


def _save_json(obj, file, indent=4, sort_keys=True):
    with open(file, 'w') as f:
        json.dump(obj, f, sort_keys=sort_keys, indent=indent)


def _subfolders(folder, join=True, prefix=None, suffix=None, sort=True):
    r = [os.path.join(folder, i) if join else i for i in os.listdir(folder) if os.path.isdir(os.path.join(folder, i))
         and (prefix is None or i.startswith(prefix)) and (suffix is None or i.endswith(suffix))]
    return sorted(r) if sort else r


def _subfiles(folder, join=True, prefix=None, suffix=None, sort=True):
    r = [os.path.join(folder, i) if join else i for i in os.listdir(folder) if os.path.isfile(os.path.join(folder, i))
         and (prefix is None or i.startswith(prefix)) and (suffix is None or i.endswith(suffix))]
    return sorted(r) if sort else r


def _save_pickle(obj, file, mode='wb'):
    with open(file, mode) as f:
        pickle.dump(obj, f)


for n, f in (('save_json', _save_json), ('subfolders', _subfolders), ('subfiles', _subfiles), ('save_pickle', _save_pickle),
             ('subdirs', _subfolders), ('write_pickle', _save_pickle)):
    setattr(ffo, n, f)
ffo.__all__ = list(ffo.__all__) + ['save_json', 'subfolders', 'subfiles', 'save_pickle', 'subdirs', 'write_pickle']

# ---- SimpleITK shim ----------------------------------------------------------------------------------------------------------


class _SitkImage:
    def __init__(self, arr):
        self.arr = np.asarray(arr)
        self.spacing, self.origin, self.direction = (1., 1., 1.), (0., 0., 0.), tuple(np.eye(3).ravel())

    def SetSpacing(self, s): self.spacing = tuple(s)
    def SetOrigin(self, s): self.origin = tuple(s)
    def SetDirection(self, s): self.direction = tuple(s)
    def GetSpacing(self): return self.spacing
    def GetOrigin(self): return self.origin
    def GetDirection(self): return self.direction
    def GetSize(self): return tuple(int(i) for i in self.arr.shape[::-1])


def _read(fname):
    im = nifti_io._read_nifti(fname)
    o = _SitkImage(im.array)
    o.spacing, o.origin, o.direction = im.spacing, im.origin, im.direction
    return o


SITK = SimpleNamespace(GetImageFromArray=lambda a: _SitkImage(a), GetArrayFromImage=lambda im: im.arr, ReadImage=_read,
                       WriteImage=lambda im, f: nifti_io._write_nifti(nifti_io.Image(im.arr, im.spacing, im.origin, im.direction), f))


def _resize(img, shape, order, mode='edge', anti_aliasing=False, **kw):
    assert mode == 'edge' and not anti_aliasing
    img = np.asarray(img, dtype=float)
    return ndimage.zoom(img, [n / o for n, o in zip(shape, img.shape)], order=order, mode='nearest', grid_mode=True)


import nnunet.preprocessing.preprocessing as pre
import nnunet.preprocessing.cropping as crop
import nnunet.inference.segmentation_export as se


def _resize_segmentation(segmentation, new_shape, order=3):
    """batchgenerators.augmentations.utils.resize_segmentation (third party, absent): per label, resize the mask with `order`
    and keep the label where the result is >= 0.5.  Only the -1 / 0 non-zero mask passes through it at test time, and every
    caller on this path discards the result (predict_MultiTalent.py:46 `d, _, dct = preprocess_fn(l)`)."""
    if order == 0:
        return _resize(segmentation.astype(float), new_shape, 0).astype(segmentation.dtype)
    out = np.zeros(new_shape, dtype=segmentation.dtype)
    for c in np.unique(segmentation):
        out[_resize((segmentation == c).astype(float), new_shape, order) >= 0.5] = c
    return out


pre.resize = _resize
pre.resize_segmentation = _resize_segmentation
crop.sitk = SITK
se.sitk = SITK

import torch.distributed as dist
from torch import nn


class FakeDDP(nn.Module):
    def __init__(self, module, device_ids=None, **kw):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


import nnunet.training.network_training.nnUNetTrainerV2_DDP as v2ddp
import nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP as mt
import nnunet.inference.predict_MultiTalent as pm
from nnunet.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions

v2ddp.DDP = FakeDDP
mt.DDP = FakeDDP
mt.aggregate_scores = lambda *a, **k: None
for m in (mt, v2ddp, pm):
    for n in ('save_json', 'subfolders', 'subfiles', 'save_pickle', 'join', 'isfile', 'isdir', 'maybe_mkdir_p', 'load_pickle', 'write_pickle'):
        setattr(m, n, getattr(ffo, n))
import nnunet.training.model_restore as mr
import nnunet.training.network_training.nnUNetTrainer as nt
import nnunet.training.network_training.network_trainer as nwt
import nnunet.training.network_training.nnUNetTrainerV2 as v2
for m in (mr, nt, nwt, v2, se, pre, crop):
    for n in ('save_json', 'subfolders', 'subfiles', 'save_pickle', 'join', 'isfile', 'isdir', 'maybe_mkdir_p', 'load_pickle', 'write_pickle'):
        setattr(m, n, getattr(ffo, n))
# the loader's oversized patch (batchgenerators' rotate_coords_3d, absent) plays no role in validation / prediction
v2.get_patch_size = lambda final_patch_size, *a, **k: np.array(final_patch_size)
_real_init_pg = dist.init_process_group
_real_torch_load = torch.load
torch.load = lambda f, map_location=None, **k: _real_torch_load(f, map_location=map_location, weights_only=False)

OUT = os.path.join(ROOT, 'tests', 'golden')
IP = {0: {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}}
STAGE = {'batch_size': 2, 'patch_size': np.array([8, 16, 16]), 'pool_op_kernel_sizes': [[2, 2, 2], [1, 2, 2]],
         'conv_kernel_sizes': [[3, 3, 3]] * 3, 'do_dummy_2D_data_aug': False, 'current_spacing': np.array([2.0, 1.0, 1.0]),
         'num_pool_per_axis': [1, 2, 2]}


def make_plans():
    return {'num_stages': 2, 'num_modalities': 1, 'modalities': {0: 'CT'}, 'normalization_schemes': OrderedDict({0: 'CT'}),
            'num_classes': 47, 'all_classes': list(range(1, 48)), 'base_num_features': 4, 'use_mask_for_norm': OrderedDict({0: False}),
            'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2], 'data_identifier': 'MultiTalent_data',
            'conv_per_stage': 2, 'plans_per_stage': {1: dict(STAGE)}, 'preprocessor_name': 'GenericPreprocessor',
            'dataset_properties': {'intensityproperties': IP}, 'keep_only_largest_region': None, 'min_region_size_per_class': None,
            'min_size_per_class': None}


def randomize(net, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in net.named_parameters():
        if p.dim() == 1 and ('norm' in n) and n.endswith('weight'):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        elif n.endswith('bias'):
            p.data = 0.3 * torch.randn(p.shape, generator=g)
        elif 'seg_outputs' in n:
            p.data = 2.5 * torch.randn(p.shape, generator=g)       # lively heads: both sides of 0.5 occur in every region


def ct_volume(rs, shape):
    v = ndimage.gaussian_filter(rs.randn(*shape), 1.2) * 700 + 40
    v[:2] = 0; v[:, :1] = 0; v[:, :, -2:] = 0                  # a zero border: crop_to_nonzero has something to cut
    return v.astype(np.float32)


def build_trainer(plans, out_base, dataset_directory, fold):
    """the reference's own class through its own constructor (nccl init and DDP wrap redirected as described above)."""
    dist.init_process_group = lambda *a, **k: None
    tr = mt.MultiTalent_trainer_ddp(plans, fold, 0, output_folder=out_base, dataset_directory=dataset_directory, batch_dice=True,
                                    stage=1, unpack_data=False, deterministic=False, fp16=False)
    dist.init_process_group = _real_init_pg
    tr.load_plans_file = lambda: None
    tr.plans = plans
    tr.initialize(False)
    return tr


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29571')
    dist.init_process_group('gloo', rank=0, world_size=1)
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp(prefix='mt_golden_drivers_')
    rec = {}
    try:
        plans = make_plans()
        pre_root = os.path.join(tmp, 'pre', 'Task100_MultiTalent')
        os.makedirs(pre_root)
        res_base = os.path.join(tmp, 'res', 'MultiTalent_trainer_ddp__plans')
        tr = build_trainer(plans, res_base, pre_root, 'all')
        net = tr.network.module
        randomize(net, 77)
        for k, v in net.state_dict().items():
            rec['sd/' + k] = v.detach().numpy().copy()

        # ---------------- (1) validate(): preprocessed cases on disk -> masks -----------------------------------------------
        rs = np.random.RandomState(5)
        folder = os.path.join(pre_root, plans['data_identifier'] + '_stage1')
        os.makedirs(folder)
        cases = [('003_liver_7', 'Task003_Liver', (14, 30, 26), (21, 30, 26), (23, 33, 28), (1, 2, 1), (3.0, 1.0, 1.0)),
                 ('017_img0003', 'Task017_AbdominalOrganSegmentation', (12, 24, 28), (12, 31, 37), (12, 31, 37), (0, 0, 0), (5.0, 0.8, 0.8)),
                 ('064_case_00009', 'Task064_KiTS_labelsFixed', (10, 20, 22), (10, 20, 22), (12, 20, 24), (2, 0, 1), (2.0, 1.0, 1.0))]
        tr.dataset = OrderedDict()
        for key, dsname, shape, after, before, lo, sp0 in cases:
            data = np.concatenate([ndimage.gaussian_filter(rs.randn(*shape), 1.0)[None] * 2.0, np.zeros((1,) + shape)]).astype(np.float32)
            data[-1][0] = -1
            props = OrderedDict(list_of_data_files=['/raw/imagesTr/' + key[4:] + '_0000.nii.gz'], valid_labels=[1, 2],
                                valid_regions=list(MultiTalent_valid_regions[dsname]), original_spacing=np.array(sp0),
                                spacing_after_resampling=np.array([2.0, 1.0, 1.0]), size_after_cropping=np.array(after),
                                original_size_of_raw_data=np.array(before),
                                crop_bbox=[[lo[i], lo[i] + after[i]] for i in range(3)], itk_spacing=tuple(sp0[::-1]),
                                itk_origin=(0., 0., 0.), itk_direction=tuple(np.eye(3).ravel()))
            np.savez_compressed(os.path.join(folder, key + '.npz'), data=data)
            _save_pickle(props, os.path.join(folder, key + '.pkl'))
            tr.dataset[key] = {'data_file': os.path.join(folder, key + '.npz'), 'properties_file': os.path.join(folder, key + '.pkl')}
            rec['val/%s/data' % key] = data
            rec['val/%s/meta' % key] = np.array(list(after) + list(before) + list(lo), dtype=np.int64)
            rec['val/%s/spacing' % key] = np.array(sp0, dtype=np.float64)
        tr.dataset_val = tr.dataset
        tr.gt_niftis_folder = os.path.join(pre_root, 'gt_segmentations')

        class SyncPool:
            def __init__(self, n): pass
            def starmap_async(self, fn, args):
                res = [fn(*a) for a in args]
                return SimpleNamespace(get=lambda: res)
            def close(self): pass
            def join(self): pass

        mt.Pool = SyncPool
        tr.validate(do_mirroring=True, use_sliding_window=True, step_size=0.5, save_softmax=False, use_gaussian=True, overwrite=True,
                    validation_folder_name='validation_raw', all_in_gpu=False)
        vf = os.path.join(tr.output_folder, 'validation_raw')
        for key, *_ in cases:
            fname = key[4:]
            rec['val/%s/seg' % key] = nifti_io._read_nifti(os.path.join(vf, fname + '.nii.gz')).array.astype(np.uint8)
            ind = [nifti_io._read_nifti(os.path.join(vf + '_individual', fname + '__' + r + '.nii.gz')).array.astype(np.uint8)
                   for r in MultiTalent_regions.keys()]
            rec['val/%s/individual' % key] = np.packbits(np.stack(ind).astype(bool), axis=None)
            print('validate', key, rec['val/%s/seg' % key].shape, np.bincount(rec['val/%s/seg' % key].ravel()),
                  'positive fraction of the region masks %.3f' % np.stack(ind).mean())
        assert os.path.isfile(os.path.join(vf, 'validation_args.json'))

        # ---------------- (2) predict_cases(): raw .nii.gz -> preprocess -> predict -> export ------------------------------------
        tr.epoch = 4
        tr.output_folder = os.path.join(res_base, 'all')
        os.makedirs(tr.output_folder, exist_ok=True)
        tr.lr_scheduler = None
        tr.optimizer = SimpleNamespace(state_dict=lambda: {})
        tr.amp_grad_scaler = None
        tr.save_checkpoint(os.path.join(tr.output_folder, 'model_final_checkpoint.model'))
        _save_pickle(plans, os.path.join(res_base, 'plans.pkl'))
        inp = os.path.join(tmp, 'in'); outp = os.path.join(tmp, 'out')
        os.makedirs(inp)
        raw = [('caseA', (20, 34, 30), (2.5, 0.9, 0.9)), ('caseB', (9, 40, 36), (6.0, 0.8, 0.8))]
        for name, shape, sp in raw:
            v = ct_volume(rs, shape)
            origin, direction = (-12.5, 30.0, 7.25), tuple(np.eye(3).ravel())
            nifti_io._write_nifti(nifti_io.Image(v, sp[::-1], origin, direction), os.path.join(inp, name + '_0000.nii.gz'))
            rec['raw/%s/vol' % name] = v
            rec['raw/%s/spacing_zyx' % name] = np.array(sp, dtype=np.float64)
        dist.init_process_group = lambda *a, **k: None
        pm.predict_from_folder(res_base, inp, outp, ['all'], False, 1, 1, None, 0, 1, True, mixed_precision=False,
                               overwrite_existing=True, mode='normal', overwrite_all_in_gpu=None, step_size=0.5,
                               checkpoint_name='model_final_checkpoint')
        dist.init_process_group = _real_init_pg
        for name, shape, sp in raw:
            ind = [nifti_io._read_nifti(os.path.join(outp, 'individual', name + '_' + r + '.nii.gz')).array.astype(np.uint8)
                   for r in MultiTalent_regions.keys()]
            st = np.stack(ind)
            assert st.shape[1:] == shape, (st.shape, shape)
            rec['raw/%s/individual' % name] = np.packbits(st.astype(bool), axis=None)
            print('predict_cases', name, st.shape, 'positive fraction %.3f' % st.mean())
        rec['regions'] = np.array(list(MultiTalent_regions.keys()))
        dst = os.path.join(OUT, 'drivers.npz')
        np.savez_compressed(dst, **rec)
        print('wrote', dst, os.path.getsize(dst) // 1024, 'KiB')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
