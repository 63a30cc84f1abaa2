"""On-disk case reader + 3D patch sampler against golden batches drawn by the REAL reference
(`tools/oracle_gen/make_golden_loader.py` -> `tests/golden/loader.npz`): same numpy seed -> bit-identical batches (the sampler
consumes the global random stream in the reference's call order), for constant/edge padding, oversized loader patches, pad_sides,
forced-foreground samples, a case without foreground, npy and npz storage; and MultiTalent's sqrt dataset balancing."""
import os
import pickle

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loader.npz')
CONFIGS = [((12, 24, 24), (12, 24, 24), 4, 0.33, 'constant', None, True),
           ((14, 28, 28), (12, 24, 24), 3, 0.5, 'edge', None, False),
           ((12, 24, 24), (12, 24, 24), 2, 0.0, 'constant', (2, 4, 4), True)]


def _write(folder, z, unpack):
    for k in z.files:
        if not k.startswith('case/'):
            continue
        name, arr = k[5:], z[k]
        np.savez_compressed(os.path.join(folder, name + '.npz'), data=arr)
        if unpack:
            np.save(os.path.join(folder, name + '.npy'), arr)
        seg = arr[-1]
        props = {'class_locations': {c: np.argwhere(seg == c) for c in (1, 2, 3, 4)},      # as the generating script built them
                 'valid_regions': ('01_spleen', '03_liver'), 'valid_labels': (1, 2)}
        with open(os.path.join(folder, name + '.pkl'), 'wb') as f:
            pickle.dump(props, f)


def test_loader_reproduces_reference_batches(tmp_path):
    from multitalent_amd.training.dataloading import dataset_loading as dl
    z = np.load(G)
    for unpack in (True, False):
        d = tmp_path / ('npy' if unpack else 'npz')
        d.mkdir()
        _write(str(d), z, unpack)
        ds = dl.load_dataset(str(d))
        assert list(ds.keys()) == [str(k) for k in z['sqrt_prob_keys']]
        p, per = dl.sqrt_sampling_probabilities(ds.keys())
        assert np.array_equal(p, z['sqrt_probabilities'])
        assert np.array_equal(np.array([per[k] for k in sorted(per)]), z['sqrt_prob_per_dataset'])
        for ci, (ps, fps, B, ov, pm, pad_sides, use_p) in enumerate(CONFIGS):
            for seed in range(4):
                np.random.seed(seed)
                loader = dl.DataLoader3D(ds, ps, fps, B, False, oversample_foreground_percent=ov, pad_mode=pm, pad_sides=pad_sides,
                                         memmap_mode='r', sampling_probabilities=p if use_p else None)
                for it in range(2):
                    b = next(loader)
                    k = 'cfg%d/seed%d/it%d/' % (ci, seed, it)
                    assert [str(x) for x in b['keys']] == [str(x) for x in z[k + 'keys']], k
                    assert b['data'].dtype == np.float32 and np.array_equal(b['data'], z[k + 'data']), k
                    assert np.array_equal(b['seg'], z[k + 'seg']), k
                    assert len(b['properties']) == B and 'valid_regions' in b['properties'][0]


def test_unpack_delete_and_target_generator(tmp_path):
    from multitalent_amd.training.dataloading import dataset_loading as dl
    z = np.load(G)
    _write(str(tmp_path), z, unpack=False)
    dl.unpack_dataset(str(tmp_path), threads=2)
    ids = sorted(dl.get_case_identifiers(str(tmp_path)))
    assert all(os.path.isfile(os.path.join(str(tmp_path), c + '.npy')) for c in ids) and len(ids) == 6
    ds = dl.load_dataset(str(tmp_path), num_cases_properties_loading_threshold=0)      # properties loaded lazily per sample
    assert 'properties' not in ds[ids[0]]
    np.random.seed(3)
    gen = dl.SegToTargetGenerator(dl.DataLoader3D(ds, (14, 28, 28), (12, 24, 24), 2, pad_mode='constant'), (12, 24, 24))
    b = next(gen)
    assert b['data'].shape == (2, 1, 12, 24, 24) and b['target'].shape == (2, 1, 12, 24, 24)
    np.random.seed(3)
    raw = next(dl.DataLoader3D(ds, (14, 28, 28), (12, 24, 24), 2, pad_mode='constant'))
    assert np.array_equal(b['data'], raw['data'][:, :, 1:13, 2:26, 2:26]) and np.array_equal(b['target'], raw['seg'][:, :, 1:13, 2:26, 2:26])
    dl.delete_npy(str(tmp_path))
    assert not any(f.endswith('.npy') for f in os.listdir(str(tmp_path)))


def test_trainer_generators_split_and_sqrt_sampling(tmp_path):
    """Trainer side of the same row: folder layout <dataset_directory>/<data_identifier>_stage<k>, the seeded 5-fold split file
    (nnUNetTrainerV2.py:276-340), MultiTalent's sqrt balancing of the TRAINING keys, batches in run_iteration's format."""
    import numpy as np
    from multitalent_amd import plans as P
    from multitalent_amd.training.model_restore import find_trainer_class
    z = np.load(G)
    sp = {'batch_size': 2, 'patch_size': np.array([12, 24, 24]), 'pool_op_kernel_sizes': [[2, 2, 2], [2, 2, 2]],
          'conv_kernel_sizes': [[3, 3, 3]] * 3, 'do_dummy_2D_data_aug': False}
    plans = P.make_plans(sp, base_num_features=8, num_classes=47, stage=1)
    folder = tmp_path / (plans['data_identifier'] + '_stage1')
    folder.mkdir()
    _write(str(folder), z, unpack=False)
    tr = find_trainer_class('nnUNetTrainerV2')(plans, 0, output_folder=None, dataset_directory=str(tmp_path), stage=1)
    tr.load_plans_file(); tr.process_plans(tr.plans)
    assert tr.folder_with_preprocessed_data == str(folder)
    np.random.seed(0)
    tr.device_augmentation = False
    dl_tr, dl_val = tr.get_basic_generators()
    assert os.path.isfile(os.path.join(str(tmp_path), 'splits_final.pkl'))
    with open(os.path.join(str(tmp_path), 'splits_final.pkl'), 'rb') as f:
        splits = pickle.load(f)
    assert len(splits) == 5 and sorted(list(splits[0]['train']) + list(splits[0]['val'])) == sorted(tr.dataset.keys())
    from sklearn.model_selection import KFold
    ks = np.sort(list(tr.dataset.keys()))
    tr_idx, va_idx = next(iter(KFold(n_splits=5, shuffle=True, random_state=12345).split(ks)))
    assert list(tr.dataset_tr.keys()) == sorted(ks[tr_idx]) and list(tr.dataset_val.keys()) == sorted(ks[va_idx])
    assert dl_tr.sampling_probabilities is None                      # stock trainers sample uniformly
    b = next(dl_tr)
    assert b['data'].shape == (2, 1, 12, 24, 24) and b['seg'].min() >= -1
    import torch.distributed as dist
    for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29631'), ('RANK', '0'), ('WORLD_SIZE', '1')):
        os.environ.setdefault(k, v)
    own_group = not dist.is_initialized()
    mt = find_trainer_class('MultiTalent_trainer_ddp')(plans, 'all', 0, output_folder=None, dataset_directory=str(tmp_path), stage=1)
    mt.load_plans_file(); mt.process_plans(mt.plans)
    mt.setup_augmentation_params()
    dl_tr, dl_val = mt.get_basic_generators()
    assert tuple(dl_tr.patch_size) == tuple(int(i) for i in mt.basic_generator_patch_size) and tuple(dl_val.patch_size) == (12, 24, 24)
    from multitalent_amd.training.dataloading.dataset_loading import sqrt_sampling_probabilities
    assert np.array_equal(dl_tr.sampling_probabilities, sqrt_sampling_probabilities(list(mt.dataset_tr.keys()))[0])
    assert abs(sum(mt.dataset_prob.values()) - 1) < 1e-12 and set(mt.dataset_prob) == {'BTCV', 'KiTS', 'LiTS'}
    if own_group:
        dist.destroy_process_group()
