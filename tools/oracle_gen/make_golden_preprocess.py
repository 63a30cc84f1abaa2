"""Golden vectors for the device pre-processing (first half): the REAL reference's
GenericPreprocessor.resample_and_normalize on small synthetic CT volumes, with skimage.transform.resize substituted by the
delegate it has used since skimage 0.19 (scipy.ndimage.zoom(order, mode='nearest', grid_mode=True)).  Writes
tests/golden/preprocess.npz.  Run: python tools/oracle_gen/make_golden_preprocess.py"""
import os, sys
import numpy as np
from scipy import ndimage
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import
ref_import.install()
import nnunet.preprocessing.preprocessing as pre


def resize(img, shape, order, mode='edge', anti_aliasing=False, **kw):
    assert mode == 'edge' and not anti_aliasing
    img = np.asarray(img, dtype=float)
    return ndimage.zoom(img, [n / o for n, o in zip(shape, img.shape)], order=order, mode='nearest', grid_mode=True)


pre.resize = resize
IP = {0: {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}}
CASES = [('iso_up', (18, 22, 20), (1.5, 1.2, 1.2), (1.0, 0.8, 0.9)),
         ('iso_down', (24, 26, 22), (0.8, 0.8, 0.8), (1.5, 1.4, 1.3)),
         ('sepz', (7, 24, 22), (5.0, 0.9, 0.9), (2.5, 1.2, 1.2)),
         ('sepz_same', (9, 20, 18), (4.0, 0.7, 0.7), (4.0, 1.0, 1.0)),
         ('identity', (8, 9, 10), (1.0, 1.0, 1.0), (1.0, 1.0, 1.0))]


def main():
    rs = np.random.RandomState(7)
    rec = {}
    for name, shape, sp0, sp1 in CASES:
        vol = ndimage.gaussian_filter(rs.randn(*shape), 1.0) * 600 + 50        # HU-like, some values beyond the clip bounds
        data = vol[None].astype(np.float32)
        g = pre.GenericPreprocessor({0: 'CT'}, {0: False}, [0, 1, 2], IP)
        out, _, props = g.resample_and_normalize(data.copy(), np.array(sp1), {'original_spacing': np.array(sp0)}, None, None)
        rec[name + '/data'] = data; rec[name + '/out'] = out.astype(np.float32)
        rec[name + '/spacing'] = np.array(list(sp0) + list(sp1))
        print(name, data.shape, '->', out.shape, float(out.min()), float(out.max()))
    dst = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden', 'preprocess.npz'))
    np.savez_compressed(dst, **rec)
    print('wrote', dst, os.path.getsize(dst) // 1024, 'KiB')


if __name__ == '__main__':
    main()
