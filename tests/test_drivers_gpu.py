"""The driver tails on the HIP path against the REAL reference's output (tests/golden/drivers.npz from
tools/oracle_gen/make_golden_drivers.py): `MultiTalent_trainer_ddp.validate` (…/MultiTalent_Trainer_DDP.py:129-322) and
`predict_MultiTalent.predict_from_folder -> predict_cases` (inference/predict_MultiTalent.py:127-376), plus a replay of the call
sequence of run/run_training_DDP.py:161-197 (train -> checkpoints -> validate) followed by predict_MultiTalent on the model it
wrote.  Masks must equal the reference's; a voxel may differ only where the (resampled) probability lies within `TIE` of the 0.5
decision boundary, and the number of such voxels is printed and bounded."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TIE = 2e-3        # |p - 0.5| below which a mask voxel may legitimately differ (fp32 device resampling vs the reference's fp64 scipy)


@pytest.fixture(scope='module')
def pg():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29581')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', init_method='env://')
    yield
    if created:
        dist.destroy_process_group()


def golden():
    return np.load(os.path.join(HERE, 'golden', 'drivers.npz'))


def plans():
    from multitalent_amd import plans as P
    sp = {'batch_size': 2, 'patch_size': np.array([8, 16, 16]), 'pool_op_kernel_sizes': [[2, 2, 2], [1, 2, 2]],
          'conv_kernel_sizes': [[3, 3, 3]] * 3, 'do_dummy_2D_data_aug': False, 'current_spacing': np.array([2.0, 1.0, 1.0])}
    return P.make_plans(sp, base_num_features=4, num_classes=47, stage=1)


def load_golden_weights(tr, z):
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    tr.network.load_state_dict(sd, strict=True)
    tr.network.engine().mark_params_dirty()


def dataset_of(key):
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_valid_regions
    names = [i for i in MultiTalent_valid_regions if i.startswith("Task%03d_" % int(key.split('_')[0]))]
    assert len(names) == 1
    return names[0]


def write_validation_cases(z, folder):
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_valid_regions
    keys = sorted({k.split('/')[1] for k in z.files if k.startswith('val/')})
    os.makedirs(folder, exist_ok=True)
    for key in keys:
        m = [int(i) for i in z['val/%s/meta' % key]]
        after, before, lo = m[0:3], m[3:6], m[6:9]
        sp0 = z['val/%s/spacing' % key]
        props = dict(list_of_data_files=['/raw/imagesTr/' + key[4:] + '_0000.nii.gz'], valid_labels=[1, 2],
                     valid_regions=list(MultiTalent_valid_regions[dataset_of(key)]), original_spacing=np.array(sp0),
                     spacing_after_resampling=np.array([2.0, 1.0, 1.0]), size_after_cropping=np.array(after),
                     original_size_of_raw_data=np.array(before), crop_bbox=[[lo[i], lo[i] + after[i]] for i in range(3)],
                     itk_spacing=tuple(float(i) for i in sp0[::-1]), itk_origin=(0., 0., 0.), itk_direction=tuple(np.eye(3).ravel()),
                     class_locations={})
        np.savez_compressed(os.path.join(folder, key + '.npz'), data=z['val/%s/data' % key])
        with open(os.path.join(folder, key + '.pkl'), 'wb') as f:
            pickle.dump(props, f)
    return keys


def compare_masks(got, ref, probs, what):
    """got / ref: [R, ...] binary masks, probs: [R, ...] resampled probabilities of the same voxels (or None)."""
    diff = got != ref
    n = int(diff.sum())
    if probs is not None:
        ties = int((np.abs(probs - 0.5) < TIE).sum())
        outside = int((diff & ~(np.abs(probs - 0.5) < TIE)).sum())
        print("%s: %d voxels, %d within %g of the boundary, %d differ (all of them ties: %s)" % (what, got.size, ties, TIE, n, outside == 0))
        assert outside == 0, "%s: %d mask voxels differ away from the decision boundary" % (what, outside)
        assert ties <= 0.01 * got.size
    assert n <= 0.002 * got.size, "%s: %d of %d mask voxels differ" % (what, n, got.size)
    return n


def test_validate_matches_reference(pg, tmp_path):
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                                        MultiTalent_regions_class_order, MultiTalent_valid_regions)
    from multitalent_amd.inference.segmentation_export import _separate_z_axis, resample_softmax
    from multitalent_amd.training.model_restore import find_trainer_class
    from multitalent_amd.utilities.nifti_io import read_image
    z = golden()
    P = plans()
    pre = tmp_path / 'pre' / 'Task100_MultiTalent'
    keys = write_validation_cases(z, str(pre / (P['data_identifier'] + '_stage1')))
    tr = find_trainer_class('MultiTalent_trainer_ddp')(P, 'all', 0, output_folder=str(tmp_path / 'res'), dataset_directory=str(pre),
                                                        stage=1, unpack_data=False, fp16=False)
    tr.initialize(False)
    load_golden_weights(tr, z)
    assert tr.output_folder.endswith(os.path.join('res', 'all'))
    tr.validate(do_mirroring=True, use_sliding_window=True, step_size=0.5, save_softmax=False, use_gaussian=True, overwrite=True,
                validation_folder_name='validation_raw')
    vf = os.path.join(tr.output_folder, 'validation_raw')
    args = json.load(open(os.path.join(vf, 'validation_args.json')))
    assert args['do_mirroring'] is True and args['step_size'] == 0.5 and args['validation_folder_name'] == 'validation_raw'
    regions = list(MultiTalent_regions.keys())
    assert regions == [str(r) for r in z['regions']]
    total = 0
    for key in keys:
        fname = key[4:]
        ref_ind = np.unpackbits(z['val/%s/individual' % key])
        seg_ref = z['val/%s/seg' % key]
        ref_ind = ref_ind[:47 * seg_ref.size].reshape((47,) + seg_ref.shape)
        got_ind = np.stack([np.asarray(read_image(os.path.join(vf + '_individual', fname + '__' + r + '.nii.gz')).array) for r in regions])
        assert got_ind.dtype == np.uint8 and got_ind.max() <= 1
        # resampled probabilities of the same voxels (torch-glue resampler, only used here to classify the differences)
        with open(tr.dataset[key]['properties_file'], 'rb') as f:
            props = pickle.load(f)
        probs = tr._predict_validation_case(key, True, (0, 1, 2), True, 0.5, True, False)
        after = [int(i) for i in props['size_after_cropping']]
        sep = _separate_z_axis(props, None) if tuple(probs.shape[1:]) != tuple(after) else -1
        pr = resample_softmax(probs.float(), after, sep).cpu().numpy()
        full = np.full((47,) + seg_ref.shape, 0.0, dtype=np.float32)
        bb = props['crop_bbox']
        full[:, bb[0][0]:bb[0][0] + after[0], bb[1][0]:bb[1][0] + after[1], bb[2][0]:bb[2][0] + after[2]] = pr
        ch = [MultiTalent_region_output_idx_mapping[r] for r in regions]
        total += compare_masks(got_ind, ref_ind, full[ch], 'validate %s individual' % key)
        # the dataset's own label map: valid regions painted in MultiTalent_regions_class_order
        seg = np.asarray(read_image(os.path.join(vf, fname + '.nii.gz')).array)
        assert seg.shape == seg_ref.shape and seg.dtype == np.uint8
        ds = dataset_of(key)
        idx = [MultiTalent_region_output_idx_mapping[i] for i in MultiTalent_valid_regions[ds]]
        near = (np.abs(full[idx] - 0.5) < TIE).any(0)
        d = seg != seg_ref
        print("validate %s label map: %d voxels, %d differ, %d of them away from a boundary" % (key, seg.size, int(d.sum()), int((d & ~near).sum())))
        assert int((d & ~near).sum()) == 0 and d.sum() <= 0.002 * seg.size
        assert set(np.unique(seg)) <= set([0] + [int(np.asarray(c).reshape(-1)[0]) for c in MultiTalent_regions_class_order[ds]])
    # overwrite=False: nothing is predicted again (every file exists) and the call still completes
    before = os.path.getmtime(os.path.join(vf, keys[0][4:] + '.nii.gz'))
    tr.validate(save_softmax=False, overwrite=False, validation_folder_name='validation_raw')
    assert os.path.getmtime(os.path.join(vf, keys[0][4:] + '.nii.gz')) == before
    print("validate: %d differing mask voxels in total" % total)


def test_predict_cases_matches_reference(pg, tmp_path):
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions
    from multitalent_amd.inference.predict_MultiTalent import predict_from_folder
    from multitalent_amd.training.model_restore import find_trainer_class
    from multitalent_amd.utilities.nifti_io import read_image, write_image
    z = golden()
    P = plans()
    model = tmp_path / 'res' / 'MultiTalent_trainer_ddp__plans'
    tr = find_trainer_class('MultiTalent_trainer_ddp')(P, 'all', 0, output_folder=str(model), dataset_directory=None, stage=1, fp16=False)
    tr.initialize(False)
    load_golden_weights(tr, z)
    tr.save_checkpoint(os.path.join(tr.output_folder, 'model_final_checkpoint.model'))
    with open(str(model / 'plans.pkl'), 'wb') as f:
        pickle.dump(P, f)
    inp, outp = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir()
    names = sorted({k.split('/')[1] for k in z.files if k.startswith('raw/')})
    for n in names:
        sp = z['raw/%s/spacing_zyx' % n]
        write_image(z['raw/%s/vol' % n], str(inp / (n + '_0000.nii.gz')), tuple(float(i) for i in sp[::-1]), (-12.5, 30.0, 7.25))
    predict_from_folder(str(model), str(inp), str(outp), ['all'], False, 1, 1, None, 0, 1, True, mixed_precision=False,
                        overwrite_existing=True, step_size=0.5, checkpoint_name='model_final_checkpoint')
    assert os.path.isfile(str(outp / 'plans.pkl'))
    regions = list(MultiTalent_regions.keys())
    for n in names:
        vol = z['raw/%s/vol' % n]
        ref = np.unpackbits(z['raw/%s/individual' % n])[:47 * vol.size].reshape((47,) + vol.shape)
        imgs = [read_image(str(outp / 'individual' / (n + '_' + r + '.nii.gz'))) for r in regions]
        got = np.stack([np.asarray(i.array) for i in imgs])
        assert got.shape == ref.shape
        # geometry of the input image is carried over to every output (segmentation_export.py:149-152)
        assert np.allclose(imgs[0].spacing, tuple(z['raw/%s/spacing_zyx' % n][::-1]), rtol=1e-6) and np.allclose(imgs[0].origin, (-12.5, 30.0, 7.25))
        d = int((got != ref).sum())
        print("predict_cases %s: %d of %d mask voxels differ from the reference's (%.4f %%)" % (n, d, got.size, 100.0 * d / got.size))
        # here the INPUT of the network already differs by the fp32 cubic resampling (<= 2e-4, tests/test_preprocess_gpu.py), so the
        # differing voxels are bounded by count: the masks agree on >= 99.8 % of all voxels and on every region's volume to 1 %
        assert d <= 0.002 * got.size
        vg, vr = got.reshape(47, -1).sum(1).astype(np.float64), ref.reshape(47, -1).sum(1).astype(np.float64)
        assert np.all(np.abs(vg - vr) <= 0.01 * np.maximum(vr, 100))


def test_training_driver_sequence_then_prediction(pg, tmp_path, monkeypatch):
    """run_training_DDP.py:146-197 through multitalent_amd.run.run_training_DDP.main (configuration from the environment, trainer by
    name, initialize, run_training, validate), then predict_MultiTalent on the folder it wrote, then `-val --valbest`."""
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions
    from multitalent_amd.inference.predict_MultiTalent import main as predict_main
    from multitalent_amd.run import run_training_DDP
    from multitalent_amd.training.model_restore import find_trainer_class
    from multitalent_amd.utilities.nifti_io import read_image, write_image
    z = golden()
    P = plans()
    pre_root, res_root = tmp_path / 'pre', tmp_path / 'res'
    task = pre_root / 'Task100_MultiTalent'
    keys = write_validation_cases(z, str(task / (P['data_identifier'] + '_stage1')))
    # class_locations for the foreground oversampling of the loader + ground truth for the evaluation
    (task / 'gt_segmentations').mkdir()
    for key in keys:
        f = str(task / (P['data_identifier'] + '_stage1') / (key + '.pkl'))
        props = pickle.load(open(f, 'rb'))
        data = z['val/%s/data' % key].copy()
        data[-1] = 0
        data[-1][2:6, 4:12, 4:12] = 1
        np.savez_compressed(f[:-4] + '.npz', data=data)
        props['class_locations'] = {1: np.argwhere(data[-1] == 1)}
        pickle.dump(props, open(f, 'wb'))
        gt = np.zeros([int(i) for i in props['original_size_of_raw_data']], dtype=np.uint8)
        gt[1:4, 3:9, 3:9] = 1
        write_image(gt, str(task / 'gt_segmentations' / (key[4:] + '.nii.gz')), props['itk_spacing'])
    with open(str(task / 'MultiTalent_tiny_plans_3D.pkl'), 'wb') as f:
        pickle.dump(P, f)
    monkeypatch.setenv('nnUNet_preprocessed', str(pre_root))
    monkeypatch.setenv('RESULTS_FOLDER', str(res_root))
    cls = find_trainer_class('MultiTalent_trainer_ddp')
    orig = cls.run_training

    def short(self):
        self.max_num_epochs, self.num_batches_per_epoch, self.num_val_batches_per_epoch, self.save_every = 3, 3, 2, 2
        return orig(self)

    monkeypatch.setattr(cls, 'run_training', short)
    np.random.seed(0)
    run_training_DDP.main(['3d_fullres', 'MultiTalent_trainer_ddp', '100', 'all', '-p', 'MultiTalent_tiny', '--fp32'])
    out = res_root / 'nnUNet' / '3d_fullres' / 'Task100_MultiTalent' / 'MultiTalent_trainer_ddp__MultiTalent_tiny' / 'all'
    files = set(os.listdir(str(out)))
    assert {'model_final_checkpoint.model', 'model_final_checkpoint.model.pkl', 'validation_raw', 'validation_raw_individual'} <= files
    assert 'model_latest.model' not in files                       # removed once the final checkpoint exists (network_trainer.py:497-500)
    ck = torch.load(str(out / 'model_final_checkpoint.model'), map_location='cpu', weights_only=False)
    assert ck['epoch'] == 3 and len(ck['plot_stuff'][0]) == 3 and len(ck['plot_stuff'][3]) == 3      # epochs, tr losses, val metrics
    assert ck['best_stuff'][2] is not None
    # model_best.model is written whenever the moving average of the validation metric improves (network_trainer.py:572-575)
    if 'model_best.model' in files:
        best = torch.load(str(out / 'model_best.model'), map_location='cpu', weights_only=False)
        assert 1 <= best['epoch'] <= 3
    for key in keys:
        assert os.path.isfile(str(out / 'validation_raw' / (key[4:] + '.nii.gz')))
        for r in MultiTalent_regions:
            assert os.path.isfile(str(out / 'validation_raw_individual' / (key[4:] + '__' + r + '.nii.gz')))
    summaries = [f for f in os.listdir(str(out / 'validation_raw')) if f.startswith('summary_')]
    assert len(summaries) == 3
    s = json.load(open(str(out / 'validation_raw' / summaries[0])))
    assert 'Dice' in s['results']['mean']['1'] and len(s['results']['all']) == 1
    # predict_MultiTalent on the trained model (its own CLI; -chk model_best when it exists)
    inp, outp = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir()
    write_image(z['raw/caseA/vol'], str(inp / 'caseA_0000.nii.gz'), tuple(float(i) for i in z['raw/caseA/spacing_zyx'][::-1]))
    model = str(out.parent)
    pickle.dump(P, open(os.path.join(model, 'plans.pkl'), 'wb'))
    chk = 'model_best' if 'model_best.model' in files else 'model_final_checkpoint'
    predict_main(['-i', str(inp), '-o', str(outp), '-m', model, '-f', 'all', '--tta', '0', '--disable_mixed_precision', '-chk', chk])
    for r in MultiTalent_regions:
        im = read_image(str(outp / 'individual' / ('caseA_' + r + '.nii.gz')))
        assert np.asarray(im.array).shape == z['raw/caseA/vol'].shape
    # validation only, from the best checkpoint (run_training_DDP.py:186-190)
    run_training_DDP.main(['3d_fullres', 'MultiTalent_trainer_ddp', 'Task100_MultiTalent', 'all', '-p', 'MultiTalent_tiny', '--fp32',
                           '-val', '--valbest', '--val_folder', 'validation_best'])
    assert os.path.isfile(str(out / 'validation_best' / (keys[0][4:] + '.nii.gz')))
