"""Device pre-processing (first half) against the REAL reference's GenericPreprocessor.resample_and_normalize
(tools/oracle_gen/make_golden_preprocess.py; skimage.resize substituted by its scipy.ndimage.zoom delegate): order-3 resampling
to the target spacing (3D, and separate z = order 3 in-plane + order 0 along the anisotropic axis) and the CT clip + z-score.
Tolerance 2e-4 absolute on z-scored intensities of range ~3.6 (float32 coefficients, mirror-initialised prefilter on the
12-voxel edge padding vs scipy's float64 'nearest' initialisation)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz')
IP = {0: {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}}


@pytest.mark.parametrize("name", ['iso_up', 'iso_down', 'sepz', 'sepz_same', 'identity'])
def test_resample_and_normalize_ct_matches_reference(dev, name):
    from multitalent_amd.preprocessing.device_preprocessing import resample_and_normalize_ct
    z = np.load(G)
    sp = z[name + '/spacing']
    out = resample_and_normalize_ct(z[name + '/data'], sp[:3], sp[3:], IP).cpu().numpy()
    ref = z[name + '/out']
    assert out.shape == ref.shape and out.dtype == np.float32
    assert np.abs(out - ref).max() < 2e-4, np.abs(out - ref).max()
