"""Golden vectors for the export post-processing: the REAL reference's
save_segmentation_nifti_from_softmax / resample_data_or_seg run in the build container with the two absent third-party calls
substituted — skimage.transform.resize by its own delegate scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True)
(oracle.reference_ops.resize_order1_edge) and SimpleITK by a recorder that captures the array about to be written.
Writes tests/golden/export.npz.  Run: python tools/oracle_gen/make_golden_export.py"""
import os, sys, copy
from types import SimpleNamespace
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, '..', '..'))
import ref_import
ref_import.install()
import nnunet.preprocessing.preprocessing as pre
import nnunet.inference.segmentation_export as se
from oracle.reference_ops import resize_order1_edge, export_segmentation

pre.resize = lambda img, shape, order, mode='edge', anti_aliasing=False: (resize_order1_edge(img, shape) if order == 1 else 1 / 0)
captured = {}


class _Img:
    def __init__(self, a): captured['arr'] = np.array(a)
    def SetSpacing(self, s): pass
    def SetOrigin(self, s): pass
    def SetDirection(self, s): pass


se.sitk = SimpleNamespace(GetImageFromArray=lambda a: _Img(a), WriteImage=lambda img, f: None)


def smooth_probs(rs, C, shape):
    from scipy.ndimage import gaussian_filter
    p = np.stack([gaussian_filter(rs.rand(*shape), 1.5) for _ in range(C)])
    p = (p - p.min()) / (p.max() - p.min())
    return p.astype(np.float32)


CASES = [   # name, C, shape, size_after_cropping, original size, bbox lower corner, original spacing, resampled spacing, regions?, force
    ('iso_regions', 5, (10, 14, 12), (13, 20, 17), (16, 22, 20), (2, 1, 3), (1.0, 0.8, 0.8), (1.5, 1.5, 1.5), True, None),
    ('sepz_regions', 5, (6, 14, 12), (11, 20, 17), (11, 24, 17), (0, 3, 0), (5.0, 0.8, 0.8), (2.5, 1.0, 1.0), True, None),
    ('sepz_same_z', 4, (9, 12, 10), (9, 21, 16), (9, 21, 16), (0, 0, 0), (4.0, 0.7, 0.7), (4.0, 1.2, 1.2), True, None),
    ('argmax_down', 6, (12, 16, 14), (9, 11, 10), (9, 11, 10), (0, 0, 0), (1.0, 1.0, 1.0), (0.8, 0.7, 0.7), False, None),
    ('forced_noz', 3, (8, 10, 10), (12, 15, 13), (13, 17, 15), (1, 2, 1), (6.0, 1.0, 1.0), (3.0, 1.0, 1.0), True, False),
    ('sep_axis2', 4, (12, 10, 5), (18, 14, 9), (18, 14, 9), (0, 0, 0), (0.9, 0.9, 4.0), (1.3, 1.3, 2.0), True, None),
    ('identity', 3, (7, 9, 8), (7, 9, 8), (9, 9, 8), (1, 0, 0), (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), True, None),
]


def main():
    rs = np.random.RandomState(42)
    rec = {}
    for name, C, shape, after, before, lo, sp0, sp1, regions, force in CASES:
        probs = smooth_probs(rs, C, shape)
        order = [int(c) for c in rs.permutation(np.arange(1, C + 1))] if regions else None
        props = {'size_after_cropping': np.array(after), 'original_size_of_raw_data': np.array(before),
                 'crop_bbox': [[lo[i], lo[i] + after[i]] for i in range(3)], 'original_spacing': np.array(sp0),
                 'spacing_after_resampling': np.array(sp1), 'itk_spacing': (1, 1, 1), 'itk_origin': (0, 0, 0),
                 'itk_direction': tuple(np.eye(3).ravel())}
        se.save_segmentation_nifti_from_softmax(probs.copy(), 'unused.nii.gz', copy.deepcopy(props), 1, order, None, None, None, None,
                                                force, 0, verbose=False)
        got = captured['arr']
        mine = export_segmentation(probs, copy.deepcopy(props), order, force)
        assert got.dtype == np.uint8 and np.array_equal(got, mine), name       # the restatement follows the reference's control flow
        rec[name + '/probs'] = probs; rec[name + '/seg'] = got
        rec[name + '/meta'] = np.array(list(after) + list(before) + list(lo), dtype=np.int64)
        rec[name + '/spacing'] = np.array(list(sp0) + list(sp1), dtype=np.float64)
        rec[name + '/order'] = np.array(order if order else [], dtype=np.int64)
        rec[name + '/force'] = np.array([-1 if force is None else int(force)])
        print(name, got.shape, np.bincount(got.ravel())[:7])
    dst = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden', 'export.npz'))
    np.savez_compressed(dst, **rec)
    print('wrote', dst, os.path.getsize(dst) // 1024, 'KiB')


if __name__ == '__main__':
    main()
