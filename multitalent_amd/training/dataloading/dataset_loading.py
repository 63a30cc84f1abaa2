"""On-disk preprocessed-case reader and 3D patch sampler.

Mirrors the reference's `nnunet/training/dataloading/dataset_loading.py`: `get_case_identifiers` (:26-28), `load_dataset`
(:91-110), `unpack_dataset` / `delete_npy` (:58-89) and `DataLoader3D` (:169-380), plus the MultiTalent sampling rule
p(case) ~ 1 / sqrt(#cases of its dataset) (`MultiTalent_Trainer_DDP.py:629-634`).

Format (written by the reference's preprocessing, read here unchanged): `<case>.npz` with key `data` = float32 `[C+1, X, Y, Z]`
(modalities, then the segmentation as LAST channel, -1 = outside the nonzero mask), optionally unpacked to `<case>.npy`
(memory-mapped); `<case>.pkl` = properties dict with `class_locations` {label: [n,3] voxel coords} and, for MultiTalent,
`valid_regions` / `valid_labels` (`Task100_MultiTalent_addregions.py:19-36`).

The sampler consumes numpy's GLOBAL random stream in exactly the reference's call order (one `choice` for the keys of a batch;
per sample either three `randint` or `choice(class)` + `choice(voxel)`), so a seeded run reproduces the reference's batches bit
for bit — that is how `tests/test_dataset_loading.py` pins it against `tests/golden/loader.npz`.
Host-side component: numpy only, no device work (the device takes over at `DeviceBatchFeeder`)."""
import os
import pickle
from collections import OrderedDict
from multiprocessing import Pool

import numpy as np


def get_case_identifiers(folder):
    return [i[:-4] for i in os.listdir(folder) if i.endswith("npz") and (i.find("segFromPrevStage") == -1)]


def _convert_to_npy(args):
    npz_file, key = args
    if not os.path.isfile(npz_file[:-3] + "npy"):
        np.save(npz_file[:-3] + "npy", np.load(npz_file)[key])


def unpack_dataset(folder, threads=8, key="data"):
    """npz -> npy next to it (memory-mappable); reference :58-73."""
    files = sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.endswith(".npz"))
    if threads <= 1 or len(files) <= 1:
        for f in files:
            _convert_to_npy((f, key))
        return
    with Pool(threads) as p:
        p.map(_convert_to_npy, zip(files, [key] * len(files)))


def delete_npy(folder):
    for c in get_case_identifiers(folder):
        f = os.path.join(folder, c + ".npy")
        if os.path.isfile(f):
            os.remove(f)


def load_pickle(file, mode='rb'):
    with open(file, mode) as f:
        return pickle.load(f)


def load_dataset(folder, num_cases_properties_loading_threshold=1000):
    """{case: {'data_file', 'properties_file'[, 'properties']}} in sorted case order; reference :91-110."""
    case_identifiers = get_case_identifiers(folder)
    case_identifiers.sort()
    dataset = OrderedDict()
    for c in case_identifiers:
        dataset[c] = OrderedDict()
        dataset[c]['data_file'] = os.path.join(folder, "%s.npz" % c)
        dataset[c]['properties_file'] = os.path.join(folder, "%s.pkl" % c)
    if len(case_identifiers) <= num_cases_properties_loading_threshold:
        for i in dataset.keys():
            dataset[i]['properties'] = load_pickle(dataset[i]['properties_file'])
    return dataset


def sqrt_sampling_probabilities(keys):
    """MultiTalent's dataset balancing (`MultiTalent_Trainer_DDP.py:629-634`): the dataset of a case is the part of its name
    before the first '_'; p(case) ~ 1/sqrt(cases in its dataset), normalised.  Returns (probabilities aligned with `keys`,
    {dataset: total probability})."""
    keys = list(keys)
    idents = list(np.unique([i.split('_')[0] for i in keys]))
    num = [len([i for i in keys if i.startswith(j + '_')]) for j in idents]
    p = np.array([1 / (num[idents.index(i.split('_')[0])] ** 0.5) for i in keys])
    p = p / sum(p)
    per_dataset = {}
    for d in idents:
        dk = [i for i in keys if i.startswith(d + '_')]
        per_dataset[d] = p[keys.index(dk[0])] * len(dk)
    return p, per_dataset


class DataLoader3D:
    """Endless iterator of {'data' [B,C,*patch] f32, 'seg' [B,1(+1),*patch] f32 (-1 padded), 'properties' [B], 'keys' [B]}.
    Same constructor and sampling as the reference's DataLoader3D (:169-380); iteration protocol of batchgenerators'
    SlimDataLoaderBase (`next(loader)` = one batch)."""

    def __init__(self, data, patch_size, final_patch_size, batch_size, has_prev_stage=False, oversample_foreground_percent=0.0,
                 memmap_mode="r", pad_mode="edge", pad_kwargs_data=None, pad_sides=None, sampling_probabilities=None):
        self._data = data
        self.batch_size = batch_size
        self.thread_id = 0
        self.pad_kwargs_data = OrderedDict() if pad_kwargs_data is None else pad_kwargs_data
        self.pad_mode = pad_mode
        self.oversample_foreground_percent = oversample_foreground_percent
        self.final_patch_size = final_patch_size
        self.has_prev_stage = has_prev_stage
        self.patch_size = patch_size
        self.list_of_keys = list(self._data.keys())
        self.need_to_pad = (np.array(patch_size) - np.array(final_patch_size)).astype(int)
        if pad_sides is not None:
            self.need_to_pad += np.asarray(pad_sides)
        self.memmap_mode = memmap_mode
        self.pad_sides = pad_sides
        self.data_shape, self.seg_shape = self.determine_shapes()
        self.sampling_probabilities = sampling_probabilities

    def __iter__(self):
        return self

    def __next__(self):
        return self.generate_train_batch()

    def set_thread_id(self, thread_id):
        self.thread_id = thread_id

    def get_do_oversample(self, batch_idx):
        """the LAST round(B * p) samples of a batch are forced to contain foreground (:204-205)."""
        return not batch_idx < round(self.batch_size * (1 - self.oversample_foreground_percent))

    def _load_case(self, entry, which='data_file'):
        f = entry[which]
        if os.path.isfile(f[:-4] + ".npy"):
            return np.load(f[:-4] + ".npy", self.memmap_mode)
        return np.load(f)['data']

    def determine_shapes(self):
        k = list(self._data.keys())[0]
        c = self._load_case(self._data[k]).shape[0] - 1
        return (self.batch_size, c, *self.patch_size), (self.batch_size, 2 if self.has_prev_stage else 1, *self.patch_size)

    def generate_train_batch(self):
        selected_keys = np.random.choice(self.list_of_keys, self.batch_size, True, self.sampling_probabilities)
        data = np.zeros(self.data_shape, dtype=np.float32)
        seg = np.zeros(self.seg_shape, dtype=np.float32)
        case_properties = []
        for j, i in enumerate(selected_keys):
            force_fg = self.get_do_oversample(j)
            entry = self._data[i]
            properties = entry['properties'] if 'properties' in entry.keys() else load_pickle(entry['properties_file'])
            case_properties.append(properties)
            case_all_data = self._load_case(entry)
            prev = None
            if self.has_prev_stage:
                segs_prev = self._load_case(entry, 'seg_from_prev_stage_file')[None]
                prev = segs_prev[np.random.choice(segs_prev.shape[0]):][:1]
            need_to_pad = self.need_to_pad.copy()
            shape = case_all_data.shape[1:]
            for d in range(3):
                if need_to_pad[d] + shape[d] < self.patch_size[d]:
                    need_to_pad[d] = self.patch_size[d] - shape[d]
            lb = [-need_to_pad[d] // 2 for d in range(3)]
            ub = [shape[d] + need_to_pad[d] // 2 + need_to_pad[d] % 2 - self.patch_size[d] for d in range(3)]
            voxels = None
            if force_fg:
                if 'class_locations' not in properties.keys():
                    raise RuntimeError("Please rerun the preprocessing with the newest version of nnU-Net!")
                fg = np.array([c for c in properties['class_locations'].keys() if len(properties['class_locations'][c]) != 0])
                fg = fg[fg > 0]
                if len(fg) == 0:
                    print('case does not contain any foreground classes', i)
                else:
                    voxels = properties['class_locations'][np.random.choice(fg)]
            if voxels is not None:
                sel = voxels[np.random.choice(len(voxels))]
                bb_lb = [max(lb[d], sel[d] - self.patch_size[d] // 2) for d in range(3)]
            else:
                bb_lb = [np.random.randint(lb[d], ub[d] + 1) for d in range(3)]
            bb_ub = [bb_lb[d] + self.patch_size[d] for d in range(3)]
            vlb = [max(0, bb_lb[d]) for d in range(3)]
            vub = [min(shape[d], bb_ub[d]) for d in range(3)]
            crop = np.copy(case_all_data[:, vlb[0]:vub[0], vlb[1]:vub[1], vlb[2]:vub[2]])
            pads = ((0, 0),) + tuple((-min(0, bb_lb[d]), max(bb_ub[d] - shape[d], 0)) for d in range(3))
            data[j] = np.pad(crop[:-1], pads, self.pad_mode, **self.pad_kwargs_data)
            seg[j, 0] = np.pad(crop[-1:], pads, 'constant', constant_values=-1)
            if prev is not None:
                seg[j, 1] = np.pad(prev[:, vlb[0]:vub[0], vlb[1]:vub[1], vlb[2]:vub[2]], pads, 'constant', constant_values=0)
        return {'data': data, 'seg': seg, 'properties': case_properties, 'keys': selected_keys}


class SegToTargetGenerator:
    """The minimum between DataLoader3D and `run_iteration` when no augmenter is attached: 'seg' -> 'target' as ONE
    full-resolution label map (the trainers build the deep-supervision pyramid and remove label -1 on the device,
    `training/data_augmentation/downsampling.py`); optional centre crop from the loader's patch to the network's patch
    (what the reference's SpatialTransform does last, `data_augmentation_moreDA.py:41-60`)."""

    def __init__(self, loader, final_patch_size=None):
        self.loader = loader
        self.final_patch_size = None if final_patch_size is None else tuple(int(i) for i in final_patch_size)

    def __iter__(self):
        return self

    def __next__(self):
        b = next(self.loader)
        data, seg = b['data'], b['seg']
        if self.final_patch_size is not None and tuple(data.shape[2:]) != self.final_patch_size:
            sl = tuple(slice((s - f) // 2, (s - f) // 2 + f) for s, f in zip(data.shape[2:], self.final_patch_size))
            data = np.ascontiguousarray(data[(slice(None), slice(None)) + sl])
            seg = np.ascontiguousarray(seg[(slice(None), slice(None)) + sl])
        return {'data': data, 'target': seg[:, :1], 'properties': b['properties'], 'keys': b['keys']}
