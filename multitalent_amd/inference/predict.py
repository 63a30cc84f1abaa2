"""One case end to end on the device (SURVEY §3.2 call stack, predict_MultiTalent.py:222-266 + preprocessing.py:226-311 +
segmentation_export.py:27-160): cropped CT -> resample to the plan spacing + clip/z-score -> sliding window (Gaussian weighting,
optional mirroring) -> probabilities resampled to the original grid, thresholded per region and re-inserted into the uncropped
volume.  The volume never leaves HBM between the stages; reading / cropping the image and writing the NIfTI stay with the
caller (SimpleITK)."""
import numpy as np

from ..preprocessing.device_preprocessing import resample_and_normalize_ct
from .segmentation_export import resample_and_classify
from .sliding_window import predict_3D


def predict_case_on_device(network, cropped_data, properties, target_spacing, intensityproperties, patch_size,
                           regions_class_order=None, do_mirroring=True, mirror_axes=(0, 1, 2), step_size=0.5,
                           transpose_forward=(0, 1, 2), force_separate_z=None, tile_shard=None, verbose=False,
                           transpose_backward=None, mixed_precision=True):
    """cropped_data: [C, X, Y, Z] (numpy or device tensor) already transposed by `transpose_forward`; properties: the case's
    dict (`original_spacing`, `size_after_cropping`, `original_size_of_raw_data`, `crop_bbox`).  Returns the uint8 label volume
    (device tensor, shape `original_size_of_raw_data`) and the properties with the resampling entries filled in.
    mixed_precision: the reference's predict default (predict_MultiTalent.py `--disable_mixed_precision` turns it off); False = the fp32 parity path."""
    spacing = np.array(properties['original_spacing'])[list(transpose_forward)]
    x = resample_and_normalize_ct(cropped_data, spacing, target_spacing, intensityproperties, force_separate_z)
    properties = dict(properties)
    properties['size_after_resampling'] = tuple(int(i) for i in x.shape[1:])
    properties['spacing_after_resampling'] = np.array(target_spacing)
    if tile_shard is not None and tile_shard[1] > 1:
        # tiles sharded over the ranks; the export below needs whole x-columns of the probabilities, so the slabs are gathered
        # (every rank then holds the full result)
        _, probs = predict_3D(network, x, do_mirroring, mirror_axes, True, step_size, patch_size, regions_class_order, True,
                              'constant', None, True, verbose, mixed_precision, tile_shard=tile_shard, return_device_tensors='full')
    else:
        _, probs = predict_3D(network, x, do_mirroring, mirror_axes, True, step_size, patch_size, regions_class_order, True,
                              'constant', None, True, verbose, mixed_precision, return_device_tensors=True)
    # the reference transposes the probabilities back before the export matches them to size_after_cropping / crop_bbox
    # (predict_MultiTalent.py:238-240: softmax.transpose([0] + [i + 1 for i in transpose_backward]))
    if transpose_backward is None:
        transpose_backward = [int(i) for i in np.argsort(list(transpose_forward))]
    if list(transpose_backward) != [0, 1, 2]:
        probs = probs.permute(0, *[int(i) + 1 for i in transpose_backward]).contiguous()
    seg = resample_and_classify(probs, properties, regions_class_order, 1, force_separate_z, 0)
    return seg, properties
