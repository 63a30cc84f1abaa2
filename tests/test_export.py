"""Export post-processing.  CPU: the oracle restatement against the golden label maps captured from the REAL
reference's save_segmentation_nifti_from_softmax (tools/oracle_gen/make_golden_export.py; skimage.resize substituted by its scipy
delegate, SimpleITK by a recorder).  GPU: mt_resample_classify through the product function against the same golden maps —
labels identical except where an interpolated probability lies within 1e-5 of a decision boundary (fp32 vs fp64 interpolation)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'export.npz')
CASES = ['iso_regions', 'sepz_regions', 'sepz_same_z', 'argmax_down', 'forced_noz', 'sep_axis2', 'identity']


def _case(z, name):
    m = z[name + '/meta']
    after, before, lo = m[0:3], m[3:6], m[6:9]
    sp = z[name + '/spacing']
    props = {'size_after_cropping': np.array(after), 'original_size_of_raw_data': np.array(before),
             'crop_bbox': [[int(lo[i]), int(lo[i] + after[i])] for i in range(3)], 'original_spacing': sp[:3],
             'spacing_after_resampling': sp[3:]}
    order = [int(c) for c in z[name + '/order']] or None
    f = int(z[name + '/force'][0])
    return z[name + '/probs'], props, order, (None if f < 0 else bool(f)), z[name + '/seg']


@pytest.mark.parametrize("name", CASES)
def test_oracle_export_matches_reference(name):
    from oracle.reference_ops import export_segmentation
    probs, props, order, force, seg = _case(np.load(G), name)
    assert np.array_equal(export_segmentation(probs, props, order, force), seg)


def _near_boundary(probs, props, order, force, eps=1e-5):
    """voxels whose decision is numerically fragile: some resampled probability within eps of 0.5 (regions) / of the maximum."""
    from oracle.reference_ops import resample_probabilities, get_do_separate_z, get_lowres_axis
    after = props['size_after_cropping']
    if force is None:
        if get_do_separate_z(props['original_spacing']):
            sep, axis = True, get_lowres_axis(props['original_spacing'])
        elif get_do_separate_z(props['spacing_after_resampling']):
            sep, axis = True, get_lowres_axis(props['spacing_after_resampling'])
        else:
            sep, axis = False, None
    else:
        sep, axis = force, (get_lowres_axis(props['original_spacing']) if force else None)
    if axis is not None and len(axis) != 1:
        sep = False
    r = resample_probabilities(probs.astype(np.float64), after, axis=axis, do_separate_z=sep)
    if order is not None:
        return (np.abs(r - 0.5) < eps).any(0)
    s = np.sort(r, 0)
    return (s[-1] - s[-2]) < eps


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_export_matches_reference(dev, name):
    from multitalent_amd.inference.segmentation_export import resample_and_classify
    probs, props, order, force, seg = _case(np.load(G), name)
    got = resample_and_classify(probs, props, order, 1, force, 0).cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == seg.shape
    lo = [b[0] for b in props['crop_bbox']]
    after = props['size_after_cropping']
    fragile = np.zeros(seg.shape, bool)
    fragile[lo[0]:lo[0] + after[0], lo[1]:lo[1] + after[1], lo[2]:lo[2] + after[2]] = _near_boundary(probs, props, order, force)
    assert np.array_equal(got[~fragile], seg[~fragile])
    assert fragile.mean() < 1e-3
    outside = np.ones(seg.shape, bool)
    outside[lo[0]:lo[0] + after[0], lo[1]:lo[1] + after[1], lo[2]:lo[2] + after[2]] = False
    assert (got[outside] == 0).all()


@pytest.mark.gpu
def test_device_export_rejects_unsupported_modes(dev):
    from multitalent_amd.inference.segmentation_export import resample_and_classify, save_segmentation_nifti_from_softmax
    probs, props, order, force, _ = _case(np.load(G), 'iso_regions')
    with pytest.raises(NotImplementedError):
        resample_and_classify(probs, props, order, 3, force, 0)
    with pytest.raises(NotImplementedError):
        save_segmentation_nifti_from_softmax(probs, 'x.nii.gz', props, 1, order, resampled_npz_fname='x.npz', verbose=False)
