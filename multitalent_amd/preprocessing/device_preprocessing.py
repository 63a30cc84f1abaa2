"""Inference pre-processing on the device (first half): resampling a cropped CT volume to the plan's spacing and
the CT intensity normalisation — GenericPreprocessor.resample_and_normalize (preprocessing.py:226-311) with resample_patient /
resample_data_or_seg(is_seg=False, order 3, separate z with order 0) (preprocessing.py:38-197).

skimage.transform.resize(order=3, mode='edge', anti_aliasing=False) is scipy.ndimage.zoom(order=3, mode='nearest',
grid_mode=True): the volume is edge-padded by 12 voxels, spline-prefiltered and sampled at x = (o + 0.5) * in/out - 0.5.  Here:
replicate padding (torch glue), `mt_spline_prefilter3`, `mt_affine_sample` (cubic, diagonal matrix).  The prefilter initialises
with mirror boundaries where scipy uses its 'nearest' rule; twelve voxels of edge padding damp the difference to z^12 = 1.4e-7.
Not on the device: cropping to the non-zero region (scipy binary_fill_holes) and file I/O — they stay with the caller."""
import numpy as np
import torch
import torch.nn.functional as F

from ..inference.segmentation_export import get_do_separate_z, get_lowres_axis
from ..training.data_augmentation.spatial import affine_sample

NPAD = 12


def _zoom3(x, new_shape, planar):
    """x: [1, C, D, H, W] device tensor -> [1, C, *new_shape]; planar: per-slice 2D resize (new_shape[0] == D)."""
    D, H, W = (int(i) for i in x.shape[2:])
    pad = (NPAD, NPAD, NPAD, NPAD, 0, 0) if planar else (NPAD,) * 6
    xp = F.pad(x, pad, mode='replicate')
    m = np.zeros((1, 12), dtype=np.float32)
    sc = [1.0 if planar else D / new_shape[0], H / new_shape[1], W / new_shape[2]]
    m[0, 0], m[0, 4], m[0, 8] = sc
    m[0, 9:] = [0 if planar else D / 2. - 0.5 + NPAD, H / 2. - 0.5 + NPAD, W / 2. - 0.5 + NPAD]
    return affine_sample(xp, m, tuple(int(i) for i in new_shape), 3, cval=0.0, planar=planar)


def resample_data(data, new_shape, axis=None, do_separate_z=False):
    """resample_data_or_seg(data, new_shape, is_seg=False, axis, order=3, do_separate_z, order_z=0) for a [C, X, Y, Z] volume."""
    assert data.is_cuda and data.dim() == 4
    shape = tuple(int(i) for i in data.shape[1:])
    new_shape = tuple(int(i) for i in new_shape)
    if shape == new_shape:
        return data
    x = data.float()
    if not do_separate_z:
        return _zoom3(x[None].contiguous(), new_shape, planar=False)[0]
    assert len(axis) == 1, "only one anisotropic axis supported"
    ax = int(axis[0])
    perm = [0, 1 + ax] + [1 + i for i in range(3) if i != ax]                 # anisotropic axis first: slices
    inv = [perm.index(i) for i in range(4)]
    xs = x.permute(perm).contiguous()
    ns = [new_shape[ax]] + [new_shape[i] for i in range(3) if i != ax]
    out = _zoom3(xs[None], (xs.shape[1], ns[1], ns[2]), planar=True)[0]        # order 3 in-plane, slice by slice
    if xs.shape[1] != ns[0]:                                                  # order 0 along the anisotropic axis
        o = torch.arange(ns[0], device=x.device, dtype=torch.float64)
        idx = torch.floor((o + 0.5) * (xs.shape[1] / ns[0]) - 0.5 + 0.5).clamp_(0, xs.shape[1] - 1).long()
        out = out.index_select(1, idx)
    return out.permute(inv).contiguous()


def resample_and_normalize_ct(data, original_spacing, target_spacing, intensityproperties, force_separate_z=None,
                              separate_z_anisotropy_threshold=3):
    """data: [C, X, Y, Z] (already cropped and transposed) numpy or device tensor; every modality is normalised with the "CT"
    scheme (clip to the training set's 0.5 / 99.5 percentiles, subtract its mean, divide by its sd).  Returns a device tensor."""
    if not torch.is_tensor(data):
        data = torch.from_numpy(np.ascontiguousarray(data))
    if not data.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("multitalent_amd: device pre-processing runs on a HIP device only; there is no CPU fallback")
        data = data.cuda()
    data = torch.nan_to_num(data.float(), nan=0.0, posinf=None, neginf=None) if torch.isnan(data).any() else data.float()
    shape = np.array(data.shape[1:])
    new_shape = np.round(((np.array(original_spacing) / np.array(target_spacing)).astype(float) * shape)).astype(int)
    if force_separate_z is not None:
        sep, axis = force_separate_z, (get_lowres_axis(original_spacing) if force_separate_z else None)
    elif get_do_separate_z(original_spacing, separate_z_anisotropy_threshold):
        sep, axis = True, get_lowres_axis(original_spacing)
    elif get_do_separate_z(target_spacing, separate_z_anisotropy_threshold):
        sep, axis = True, get_lowres_axis(target_spacing)
    else:
        sep, axis = False, None
    if axis is not None and len(axis) != 1:
        sep = False
    out = resample_data(data, new_shape, axis, sep)
    if out is data:
        out = data.clone()
    for c in range(out.shape[0]):
        ip = intensityproperties[c]
        out[c].clamp_(float(ip['percentile_00_5']), float(ip['percentile_99_5'])).sub_(float(ip['mean'])).div_(float(ip['sd']))
    return out
