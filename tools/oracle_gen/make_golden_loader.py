"""Golden vectors for the on-disk case reader / patch sampler, produced by the REAL reference
(`nnunet.training.dataloading.dataset_loading.DataLoader3D`, `MultiTalent_trainer_ddp.get_basic_generators`) in the build
container.  Writes tests/golden/loader.npz: the synthetic cases (so the test can re-create the dataset folder), and for several
seeds / configurations the batches the reference draws, plus its sqrt sampling probabilities.
Run: python tools/oracle_gen/make_golden_loader.py"""
import os, sys, pickle, tempfile, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import
ref_import.install()

# batchgenerators.dataloading.data_loader.SlimDataLoaderBase (third party, absent): published semantics = store data and
# batch size, iterate by calling generate_train_batch()
import batchgenerators.dataloading.data_loader as bdl


class SlimDataLoaderBase:
    def __init__(self, data, batch_size, number_of_threads_in_multithreaded=None):
        self._data = data
        self.batch_size = batch_size
        self.thread_id = 0

    def __iter__(self):
        return self

    def __next__(self):
        return self.generate_train_batch()


bdl.SlimDataLoaderBase = SlimDataLoaderBase
from nnunet.training.dataloading import dataset_loading as ref_dl           # noqa: E402

CASES = [('BTCV_0001', (10, 30, 28)), ('BTCV_0002', (14, 20, 40)), ('BTCV_0003', (16, 26, 26)), ('LiTS_0001', (12, 24, 24)),
         ('KiTS_0001', (20, 33, 21)), ('KiTS_0002', (9, 18, 30))]
CONFIGS = [   # (patch_size, final_patch_size, batch, oversample, pad_mode, pad_sides, use sqrt probabilities)
    ((12, 24, 24), (12, 24, 24), 4, 0.33, 'constant', None, True),
    ((14, 28, 28), (12, 24, 24), 3, 0.5, 'edge', None, False),
    ((12, 24, 24), (12, 24, 24), 2, 0.0, 'constant', (2, 4, 4), True),
]
SEEDS = [0, 1, 2, 3]


def make_case(rs, shape, no_fg=False):
    img = rs.randint(-50, 50, size=(1,) + shape).astype(np.float32) / 8
    seg = np.zeros(shape, dtype=np.float32)
    if not no_fg:
        for lab in (1, 2, 4):
            lo = [rs.randint(0, s - 3) for s in shape]
            hi = [min(s, l + rs.randint(2, 6)) for s, l in zip(shape, lo)]
            seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = lab
    seg[:1] = -1                                  # a slab outside the nonzero mask
    props = {'class_locations': {c: np.argwhere(seg == c) for c in (1, 2, 3, 4)},    # class 3 is always empty
             'valid_regions': ('01_spleen', '03_liver'), 'valid_labels': (1, 2)}
    return np.concatenate([img, seg[None]], 0), props


def write_dataset(folder, cases, unpack):
    for name, (arr, props) in cases.items():
        np.savez_compressed(os.path.join(folder, name + '.npz'), data=arr)
        if unpack:
            np.save(os.path.join(folder, name + '.npy'), arr)
        with open(os.path.join(folder, name + '.pkl'), 'wb') as f:
            pickle.dump(props, f)


def main():
    rs = np.random.RandomState(123)
    cases = {n: make_case(rs, s, no_fg=(n == 'KiTS_0002')) for n, s in CASES}
    out = {}
    for n, (arr, props) in cases.items():
        out['case/' + n] = arr
    with tempfile.TemporaryDirectory() as d:
        write_dataset(d, cases, unpack=True)
        ds = ref_dl.load_dataset(d)
        assert list(ds.keys()) == sorted(n for n, _ in CASES)
        # sqrt probabilities through the reference trainer method (unbound, fake self)
        from nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
        import nnunet.training.network_training.custom_trainers.MultiTalent.MultiTalent.MultiTalent_Trainer_DDP as mtmod
        mtmod.DataLoader3D = ref_dl.DataLoader3D
        fake = types.SimpleNamespace(load_dataset=lambda: None, do_split=lambda: None, dataset_tr=ds, dataset_val=ds,
                                     print_to_log_file=lambda *a, **k: None, threeD=True, basic_generator_patch_size=(12, 24, 24),
                                     patch_size=(12, 24, 24), batch_size=2, oversample_foreground_percent=0.33, pad_all_sides=None)
        dl_tr, _ = MultiTalent_trainer_ddp.get_basic_generators(fake)
        out['sqrt_probabilities'] = np.asarray(dl_tr.sampling_probabilities, dtype=np.float64)
        out['sqrt_prob_keys'] = np.array(list(ds.keys()))
        out['sqrt_prob_per_dataset'] = np.array([fake.dataset_prob[k] for k in sorted(fake.dataset_prob)])
        for ci, (ps, fps, B, ov, pm, pad_sides, use_p) in enumerate(CONFIGS):
            for seed in SEEDS:
                np.random.seed(seed)
                dl = ref_dl.DataLoader3D(ds, ps, fps, B, False, oversample_foreground_percent=ov, pad_mode=pm,
                                         pad_sides=pad_sides, memmap_mode='r',
                                         sampling_probabilities=dl_tr.sampling_probabilities if use_p else None)
                for it in range(2):
                    b = next(dl)
                    k = 'cfg%d/seed%d/it%d/' % (ci, seed, it)
                    out[k + 'data'] = b['data']; out[k + 'seg'] = b['seg']; out[k + 'keys'] = np.array(b['keys'])
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'loader.npz')
    np.savez_compressed(dst, **out)
    print('wrote', os.path.normpath(dst), os.path.getsize(dst) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
