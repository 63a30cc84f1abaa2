"""Device-side deep-supervision target pyramid.

Mirrors nnunet/training/data_augmentation/downsampling.py:70-104 (`DownsampleSegForDSTransform2`,
`downsample_seg_for_ds_transform2`, order 0) and `RemoveLabelTransform(-1, 0)` (data_augmentation_moreDA.py:117): the reference
runs them in CPU augmentation workers and ships five label maps per sample over PCIe; here the full-resolution label map is
uploaded once and the pyramid is built by `mt_downsample_seg_nearest` on the device."""
import numpy as np
import torch

from ... import ops


def downsample_seg_for_ds_transform2(seg, ds_scales=((1, 1, 1), (0.5, 0.5, 0.5), (0.25, 0.25, 0.25)), order=0, axes=None,
                                     remove_minus_one=False):
    """seg: [B, C, D, H, W] float32 tensor on the HIP device.  Returns the list of label maps, highest resolution first."""
    if order != 0:
        raise NotImplementedError("only nearest-neighbour (order 0) label pyramids are on the hot path (downsampling.py:86)")
    if axes is not None and list(axes) != [2, 3, 4]:
        raise NotImplementedError("axes other than the three spatial ones")
    seg = seg.contiguous().float()
    if remove_minus_one:
        seg = torch.where(seg == -1, torch.zeros((), device=seg.device), seg)
    out = []
    for s in ds_scales:
        if all(i == 1 for i in s):
            out.append(seg)                                                          # downsampling.py:92-93
        else:
            new_shape = np.round(np.array(seg.shape[2:], dtype=float) * np.array(s, dtype=float)).astype(int)   # :95-98
            out.append(ops.downsample_seg_nearest(seg, tuple(int(i) for i in new_shape)))
    return out


class DownsampleSegForDSTransform2:
    """Same call convention as the reference transform: `**data_dict` in, dict out."""

    def __init__(self, ds_scales=(1, 0.5, 0.25), order=0, input_key="seg", output_key="seg", axes=None):
        self.axes, self.output_key, self.input_key, self.order, self.ds_scales = axes, output_key, input_key, order, ds_scales

    def __call__(self, **data_dict):
        data_dict[self.output_key] = downsample_seg_for_ds_transform2(data_dict[self.input_key], self.ds_scales, self.order,
                                                                      self.axes)
        return data_dict
