"""Export post-processing on the device (reference inference/segmentation_export.py:27-160 with
preprocessing.py:28-35,109-197): probabilities at the network's spacing -> label map on the ORIGINAL voxel grid, re-inserted into
the uncropped volume.  One kernel (`mt_resample_classify`) interpolates every channel at each output voxel and keeps only the
label, so the resampled multi-channel volume of the reference (47 x 512^3 floats = 25 GB, resized channel by channel on the CPU)
never exists.  Supported: the reference's defaults for this path — interpolation order 1, separate-z resampling with order 0
when the spacing is anisotropic (ratio > 3) — anything else raises (there is no CPU fallback)."""
import os
from copy import deepcopy

import numpy as np
import torch

from .. import _lib

RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3


def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold                 # preprocessing.py:28-30


def get_lowres_axis(new_spacing):
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]                # preprocessing.py:33-35


def _separate_z_axis(properties_dict, force_separate_z):
    """the decision tree of segmentation_export.py:80-101; returns the anisotropic axis or -1."""
    if force_separate_z is None:
        if get_do_separate_z(properties_dict.get('original_spacing')):
            sep, axis = True, get_lowres_axis(properties_dict.get('original_spacing'))
        elif get_do_separate_z(properties_dict.get('spacing_after_resampling')):
            sep, axis = True, get_lowres_axis(properties_dict.get('spacing_after_resampling'))
        else:
            sep, axis = False, None
    else:
        sep = force_separate_z
        axis = get_lowres_axis(properties_dict.get('original_spacing')) if sep else None
    if axis is not None and len(axis) != 1:
        sep = False
    return int(axis[0]) if sep else -1


def resample_and_classify(segmentation_softmax, properties_dict, region_class_order=None, order=1, force_separate_z=None,
                          interpolation_order_z=0, device=None):
    """-> uint8 device tensor of shape `original_size_of_raw_data` (or of the resampled shape when there is no crop_bbox)."""
    if order != 1 or interpolation_order_z != 0:
        raise NotImplementedError("device export supports interpolation order 1 (order_z 0), the reference's defaults")
    if torch.is_tensor(segmentation_softmax):
        p = segmentation_softmax
    else:
        p = torch.from_numpy(np.ascontiguousarray(segmentation_softmax))
    if not p.is_cuda:
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("multitalent_amd: export post-processing runs on a HIP device only; there is no CPU fallback")
            device = torch.device('cuda', torch.cuda.current_device())
        p = p.to(device)
    p = p.float().contiguous()
    C, D, H, W = (int(i) for i in p.shape)
    after = [int(i) for i in properties_dict.get('size_after_cropping')]
    sep_axis = -1
    if any(i != j for i, j in zip((D, H, W), after)):
        sep_axis = _separate_z_axis(properties_dict, force_separate_z)
    bbox = properties_dict.get('crop_bbox')
    if bbox is not None:
        full = [int(i) for i in properties_dict.get('original_size_of_raw_data')]
        lo = [int(bbox[c][0]) for c in range(3)]
        if any(lo[c] + after[c] > full[c] for c in range(3)):
            raise ValueError("crop_bbox + size_after_cropping exceeds original_size_of_raw_data")     # the reference fails here too
    else:
        full, lo = after, [0, 0, 0]
    out = torch.zeros(full, dtype=torch.uint8, device=p.device)
    order_t = None
    if region_class_order is not None:
        # an entry may be a 1-tuple: the drivers pass ((1,),) for a single-region mask (predict_MultiTalent.py:259)
        order_t = torch.tensor([int(np.asarray(c).reshape(-1)[0]) for c in region_class_order], dtype=torch.int32, device=p.device)
        assert order_t.numel() == C, "one class per channel expected in region_class_order"
    lib = _lib.load()
    _lib.check(lib.mt_resample_classify(p.data_ptr(), C, D, H, W, after[0], after[1], after[2], sep_axis,
                                        order_t.data_ptr() if order_t is not None else None, 1 if order_t is not None else 0,
                                        out.data_ptr(), full[0], full[1], full[2], lo[0], lo[1], lo[2],
                                        torch.cuda.current_stream(p.device).cuda_stream), 'resample_classify')
    return out


def resample_softmax(p, after, sep_axis):
    """The resampled multi-channel volume itself (only for `resampled_npz_fname`, i.e. ensembling exports; torch glue): order 1
    at x = (o + 0.5) in/out - 0.5 with edge clamp, = `F.interpolate(align_corners=False)`; with a separate-z axis every slice is
    resized in-plane and the slices are picked with nearest neighbour (floor(x + 0.5)), preprocessing.py:109-197."""
    import torch.nn.functional as F
    shape = tuple(int(i) for i in p.shape[1:])
    after = tuple(int(i) for i in after)
    if shape == after:
        return p
    if sep_axis < 0:
        return F.interpolate(p[None], size=after, mode='trilinear', align_corners=False)[0]
    perm = [0, 1 + sep_axis] + [1 + i for i in range(3) if i != sep_axis]
    inv = [perm.index(i) for i in range(4)]
    q = p.permute(perm)                                                       # [C, Z, A, B]
    rest = [after[i] for i in range(3) if i != sep_axis]
    C, Z = int(q.shape[0]), int(q.shape[1])
    q = F.interpolate(q.reshape(1, C * Z, q.shape[2], q.shape[3]), size=rest, mode='bilinear', align_corners=False)
    q = q.reshape(C, Z, rest[0], rest[1])
    nz = after[sep_axis]
    if nz != Z:
        o = torch.arange(nz, device=p.device, dtype=torch.float64)
        idx = torch.floor((o + 0.5) * (Z / nz) - 0.5 + 0.5).clamp_(0, Z - 1).long()
        q = q.index_select(1, idx)
    return q.permute(inv).contiguous()


def save_segmentation_nifti_from_softmax(segmentation_softmax, out_fname, properties_dict, order=1, region_class_order=None,
                                         seg_postprogess_fn=None, seg_postprocess_args=None, resampled_npz_fname=None,
                                         non_postprocessed_fname=None, force_separate_z=None, interpolation_order_z=0, verbose=True):
    """Same signature as the reference (segmentation_export.py:27-33).  Returns the uint8 array that is written.  Files are
    written through SimpleITK when it is installed, otherwise by `utilities.nifti_io`'s NIfTI-1 writer; `resample_and_classify`
    is the device-only core for callers with their own writer."""
    import pickle
    from ..utilities.nifti_io import write_image
    if verbose:
        print("force_separate_z:", force_separate_z, "interpolation order:", order)
    if isinstance(segmentation_softmax, str):
        assert os.path.isfile(segmentation_softmax), "segmentation_softmax must point to an existing npy/npz file"
        del_file = deepcopy(segmentation_softmax)
        segmentation_softmax = np.load(del_file) if del_file.endswith('.npy') else np.load(del_file)['softmax']
        os.remove(del_file)
    if resampled_npz_fname is not None:
        # the reference stores the RESAMPLED probabilities as float16 (+ the properties) for ensembling (:117-122)
        if order != 1 or interpolation_order_z != 0:
            raise NotImplementedError("device export supports interpolation order 1 (order_z 0), the reference's defaults")
        p = segmentation_softmax if torch.is_tensor(segmentation_softmax) else torch.from_numpy(np.ascontiguousarray(segmentation_softmax))
        if not p.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("multitalent_amd: export post-processing runs on a HIP device only; there is no CPU fallback")
            p = p.cuda()
        p = p.float()
        after = [int(i) for i in properties_dict.get('size_after_cropping')]
        sep = _separate_z_axis(properties_dict, force_separate_z) if any(i != j for i, j in zip(p.shape[1:], after)) else -1
        np.savez_compressed(resampled_npz_fname, softmax=resample_softmax(p, after, sep).cpu().numpy().astype(np.float16))
        if region_class_order is not None:
            properties_dict['regions_class_order'] = region_class_order
        with open(resampled_npz_fname[:-4] + ".pkl", 'wb') as f:
            pickle.dump(properties_dict, f)
        segmentation_softmax = p
    seg = resample_and_classify(segmentation_softmax, properties_dict, region_class_order, order, force_separate_z,
                                interpolation_order_z).cpu().numpy()
    post = seg_postprogess_fn(np.copy(seg), *seg_postprocess_args) if seg_postprogess_fn is not None else seg

    def write(arr, fname):
        write_image(arr.astype(np.uint8), fname, properties_dict['itk_spacing'], properties_dict['itk_origin'],
                    properties_dict['itk_direction'])

    write(post, out_fname)
    if non_postprocessed_fname is not None and seg_postprogess_fn is not None:
        write(seg, non_postprocessed_fname)
    return post
