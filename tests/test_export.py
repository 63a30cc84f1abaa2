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
    from multitalent_amd.inference.segmentation_export import resample_and_classify
    probs, props, order, force, _ = _case(np.load(G), 'iso_regions')
    with pytest.raises(NotImplementedError):
        resample_and_classify(probs, props, order, 3, force, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ['iso_regions', 'sepz_regions', 'sep_axis2', 'identity'])
def test_device_export_writes_file_and_resampled_softmax(dev, tmp_path, name):
    """`resampled_npz_fname` (validate(save_softmax=True), segmentation_export.py:117-122): the resampled probabilities as
    float16 + the properties pickle; and the label map written as a NIfTI file carries the case's geometry."""
    import pickle
    from multitalent_amd.inference.segmentation_export import save_segmentation_nifti_from_softmax
    from multitalent_amd.utilities.nifti_io import read_image
    from oracle.reference_ops import resample_probabilities, get_do_separate_z, get_lowres_axis
    probs, props, order, force, seg = _case(np.load(G), name)
    props = dict(props, itk_spacing=(0.8, 0.9, 2.5), itk_origin=(1.0, -2.0, 3.0), itk_direction=tuple(np.eye(3).ravel()))
    out, npz = str(tmp_path / 'c.nii.gz'), str(tmp_path / 'c.npz')
    ret = save_segmentation_nifti_from_softmax(probs, out, props, 1, order, None, None, npz, None, force, 0, verbose=False)
    im = read_image(out)
    assert np.array_equal(np.asarray(im.array), ret) and np.allclose(im.spacing, (0.8, 0.9, 2.5)) and np.allclose(im.origin, (1.0, -2.0, 3.0))
    sm = np.load(npz)['softmax']
    assert sm.dtype == np.float16 and sm.shape[1:] == tuple(props['size_after_cropping'])
    if get_do_separate_z(props['original_spacing']):
        sep, axis = True, get_lowres_axis(props['original_spacing'])
    elif get_do_separate_z(props['spacing_after_resampling']):
        sep, axis = True, get_lowres_axis(props['spacing_after_resampling'])
    else:
        sep, axis = False, None
    ref = resample_probabilities(probs.astype(np.float64), props['size_after_cropping'], axis=axis, do_separate_z=sep)
    assert np.abs(sm.astype(np.float64) - ref).max() < 1e-3                      # float16 storage
    saved = pickle.load(open(npz[:-4] + '.pkl', 'rb'))
    assert ('regions_class_order' in saved) == (order is not None)


@pytest.mark.gpu
def test_case_end_to_end_on_device(dev):
    """predict_case_on_device = resample_and_normalize_ct -> predict_3D -> resample_and_classify without leaving the device; the
    device-resident input path of predict_3D equals its numpy path (same padding, same tiles)."""
    import torch
    from torch import nn
    from multitalent_amd.inference.predict import predict_case_on_device
    from multitalent_amd.inference.sliding_window import predict_3D
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.preprocessing.device_preprocessing import resample_and_normalize_ct
    torch.manual_seed(3)
    net = Generic_UNet(1, 8, 5, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, None,
                       [[2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 3, False, True, True).to(dev)
    net.eval(); net.do_ds = False
    net.inference_apply_nonlin = nn.Sigmoid()
    rs = np.random.RandomState(0)
    ct = (rs.randn(1, 14, 40, 36) * 300).astype(np.float32)
    props = {'original_spacing': np.array([3.0, 0.9, 0.9]), 'size_after_cropping': np.array([14, 40, 36]),
             'original_size_of_raw_data': np.array([16, 44, 36]), 'crop_bbox': [[1, 15], [2, 42], [0, 36]]}
    ip = {0: {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}}
    order = [1, 2, 3, 4, 5]
    seg, p2 = predict_case_on_device(net, ct, props, (2.0, 1.2, 1.2), ip, (16, 32, 32), order, do_mirroring=True)
    assert seg.dtype == torch.uint8 and tuple(seg.shape) == (16, 44, 36) and p2['size_after_resampling'] == (21, 30, 27)
    assert int(seg[0].max()) == 0 and int(seg[-1].max()) == 0 and int(seg[:, :2].max()) == 0          # outside the crop box
    x = resample_and_normalize_ct(ct, props['original_spacing'], (2.0, 1.2, 1.2), ip)
    a = predict_3D(net, x, True, (0, 1, 2), True, 0.5, (16, 32, 32), order, True, 'constant', None, True, False, True, return_device_tensors=True)
    b = predict_3D(net, x.cpu().numpy(), True, (0, 1, 2), True, 0.5, (16, 32, 32), order, True, 'constant', None, True, False, True,
                   return_device_tensors=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
