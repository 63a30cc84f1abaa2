"""Import shim for the reference (runs ONLY in the build container, never on the GPU box).

The reference needs third-party packages this image lacks (batchgenerators, SimpleITK, ...).
This module fabricates empty stand-in *modules* for those names so that the reference's own
network / loss / trainer code can be imported and executed on CPU to produce golden vectors
(SURVEY.md §8c, Appendix A).  Nothing here is shipped or imported by the product.
"""
import sys, types, os, pickle, importlib.abc, importlib.machinery
import numpy as np

REFERENCE_ROOT = '/root/reference'
MISSING = ('batchgenerators', 'SimpleITK', 'nibabel', 'skimage', 'medpy', 'dicom2nifti', 'tifffile',
           'hiddenlayer', 'unittest2', 'monai')


class _Dummy:
    def __init__(s, *a, **k): pass
    def __call__(s, *a, **k): return _Dummy()
    def __getattr__(s, n): return _Dummy()


class _Mod(types.ModuleType):
    __path__ = []
    def __getattr__(s, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return type(n, (_Dummy,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(s, name, path, target=None):
        if name.split('.')[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, s, is_package=True)
    def create_module(s, spec): return _Mod(spec.name)
    def exec_module(s, m): pass


def pad_nd_image(image, new_shape=None, mode="constant", kwargs=None, return_slicer=False,
                 shape_must_be_divisible_by=None):
    """Restatement of batchgenerators.augmentations.utils.pad_nd_image (batchgenerators>=0.23, third
    party, not vendored by the reference; semantics per SURVEY.md §8c)."""
    if kwargs is None:
        kwargs = {'constant_values': 0}
    if new_shape is not None:
        old_shape = np.array(image.shape[-len(new_shape):])
    else:
        assert shape_must_be_divisible_by is not None
        new_shape = image.shape[-len(shape_must_be_divisible_by):]
        old_shape = new_shape
    num_axes_nopad = len(image.shape) - len(new_shape)
    new_shape = [max(new_shape[i], old_shape[i]) for i in range(len(new_shape))]
    if not isinstance(new_shape, np.ndarray):
        new_shape = np.array(new_shape)
    if shape_must_be_divisible_by is not None:
        if not isinstance(shape_must_be_divisible_by, (list, tuple, np.ndarray)):
            shape_must_be_divisible_by = [shape_must_be_divisible_by] * len(new_shape)
        for i in range(len(new_shape)):
            if new_shape[i] % shape_must_be_divisible_by[i] == 0:
                new_shape[i] -= shape_must_be_divisible_by[i]
        new_shape = np.array([new_shape[i] + shape_must_be_divisible_by[i] - new_shape[i] %
                              shape_must_be_divisible_by[i] for i in range(len(new_shape))])
    difference = new_shape - old_shape
    pad_below = difference // 2
    pad_above = difference // 2 + difference % 2
    pad_list = [[0, 0]] * num_axes_nopad + list([list(i) for i in zip(pad_below, pad_above)])
    if not ((all([i == 0 for i in pad_below])) and (all([i == 0 for i in pad_above]))):
        res = np.pad(image, pad_list, mode, **kwargs)
    else:
        res = image
    if not return_slicer:
        return res
    pad_list = np.array(pad_list)
    pad_list[:, 1] = np.array(res.shape) - pad_list[:, 1]
    slicer = list(slice(*i) for i in pad_list)
    return res, slicer


_installed = False


def install():
    global _installed
    if _installed:
        return
    sys.meta_path.insert(0, _Finder())
    import batchgenerators.utilities.file_and_folder_operations as ffo
    ffo.join, ffo.isfile, ffo.isdir, ffo.os = os.path.join, os.path.isfile, os.path.isdir, os
    ffo.maybe_mkdir_p = lambda p: os.makedirs(p, exist_ok=True)
    ffo.load_pickle = lambda f, mode='rb': pickle.load(open(f, mode))
    ffo.__all__ = ['join', 'isfile', 'isdir', 'os', 'maybe_mkdir_p', 'load_pickle']
    import batchgenerators.augmentations.utils as bau
    bau.pad_nd_image = pad_nd_image
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
