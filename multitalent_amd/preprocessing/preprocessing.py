"""`GenericPreprocessor` as the trainers' `preprocess_patient` uses it (reference preprocessing/preprocessing.py:200-321): crop on
the host, then resample to the plan's spacing and normalise ON THE DEVICE (`device_preprocessing.resample_and_normalize_ct`:
`mt_spline_prefilter3` + `mt_affine_sample`).  Only the "CT" normalisation scheme of the MultiTalent plans is on this path; any
other scheme raises (there is no CPU fallback)."""
import numpy as np
import torch

from .cropping import ImageCropper
from .device_preprocessing import resample_and_normalize_ct

RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3


class GenericPreprocessor(object):
    def __init__(self, normalization_scheme_per_modality, use_nonzero_mask, transpose_forward, intensityproperties=None):
        self.transpose_forward = list(transpose_forward)
        self.intensityproperties = intensityproperties
        self.normalization_scheme_per_modality = normalization_scheme_per_modality
        self.use_nonzero_mask = use_nonzero_mask
        self.resample_separate_z_anisotropy_threshold = RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD
        self.resample_order_data = 3
        self.resample_order_seg = 1

    def resample_and_normalize(self, data, target_spacing, properties, seg=None, force_separate_z=None, return_device=False):
        """preprocessing.py:226-311.  `data` / `seg` are already transposed, `properties['original_spacing']` is not.  The
        returned seg (the -1 / 0 non-zero mask at test time, which no caller on this path reads) is resampled with nearest
        neighbour on the host."""
        schemes = [self.normalization_scheme_per_modality[c] for c in range(len(data))]
        if any(s != "CT" for s in schemes):
            raise NotImplementedError("device pre-processing implements the 'CT' normalisation scheme (got %s)" % schemes)
        if any(self.use_nonzero_mask[c] for c in range(len(data))):
            raise NotImplementedError("use_mask_for_norm is not on the device pre-processing path")
        assert self.intensityproperties is not None, "ERROR: if there is a CT then we need intensity properties"
        spacing = np.array(properties["original_spacing"])[self.transpose_forward]
        out = resample_and_normalize_ct(data, spacing, target_spacing, self.intensityproperties, force_separate_z,
                                        self.resample_separate_z_anisotropy_threshold)
        new_shape = tuple(int(i) for i in out.shape[1:])
        if seg is not None and tuple(seg.shape[1:]) != new_shape:
            idx = [np.clip(np.floor((np.arange(n) + 0.5) * (o / n)).astype(int), 0, o - 1) for n, o in zip(new_shape, seg.shape[1:])]
            seg = seg[:, idx[0]][:, :, idx[1]][:, :, :, idx[2]]
        if seg is not None:
            seg[seg < -1] = 0
        properties["size_after_resampling"] = new_shape
        properties["spacing_after_resampling"] = target_spacing
        return (out if return_device else out.cpu().numpy()), seg, properties

    def preprocess_test_case(self, data_files, target_spacing, seg_file=None, force_separate_z=None, return_device=False):
        """preprocessing.py:313-321."""
        data, seg, properties = ImageCropper.crop_from_list_of_files(data_files, seg_file)
        data = data.transpose((0, *[i + 1 for i in self.transpose_forward]))
        seg = seg.transpose((0, *[i + 1 for i in self.transpose_forward]))
        data, seg, properties = self.resample_and_normalize(data, target_spacing, properties, seg, force_separate_z, return_device)
        if not torch.is_tensor(data):
            data = data.astype(np.float32)
        return data, seg, properties
