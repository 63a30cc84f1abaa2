"""Reading a case and cropping it to its non-zero region (reference preprocessing/cropping.py:23-155) — the host-side front of
`preprocess_patient`.  binary_fill_holes (scipy) and file reading stay on the host; everything after the crop runs on the device
(preprocessing.GenericPreprocessor.resample_and_normalize)."""
from collections import OrderedDict

import numpy as np
from scipy.ndimage import binary_fill_holes

from ..utilities.nifti_io import read_image


def create_nonzero_mask(data):
    assert data.ndim == 4, "data must have shape (C, X, Y, Z)"
    nonzero = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        nonzero |= data[c] != 0
    return binary_fill_holes(nonzero)


def get_bbox_from_mask(mask, outside_value=0):
    idx = np.where(mask != outside_value)
    return [[int(np.min(i)), int(np.max(i)) + 1] for i in idx]


def crop_to_bbox(image, bbox):
    return image[tuple(slice(b[0], b[1]) for b in bbox)]


def get_case_identifier(case):
    return case[0].split("/")[-1].split(".nii.gz")[0][:-5]


def load_case_from_list_of_files(data_files, seg_file=None):
    """cropping.py:61-81: one file per modality -> ([C, Z, Y, X] float32, seg or None, properties)."""
    assert isinstance(data_files, (list, tuple)), "case must be either a list or a tuple"
    images = [read_image(f) for f in data_files]
    props = OrderedDict()
    props["original_size_of_raw_data"] = np.array(images[0].GetSize())[[2, 1, 0]]
    props["original_spacing"] = np.array(images[0].GetSpacing())[[2, 1, 0]]
    props["list_of_data_files"] = data_files
    props["seg_file"] = seg_file
    props["itk_origin"] = images[0].GetOrigin()
    props["itk_spacing"] = images[0].GetSpacing()
    props["itk_direction"] = images[0].GetDirection()
    data = np.vstack([np.asarray(i.array)[None] for i in images]).astype(np.float32)
    seg = np.asarray(read_image(seg_file).array)[None].astype(np.float32) if seg_file is not None else None
    return data, seg, props


def crop_to_nonzero(data, seg=None, nonzero_label=-1):
    """cropping.py:84-116: the crop box of the hole-filled non-zero mask; voxels outside the mask get `nonzero_label` in seg."""
    mask = create_nonzero_mask(data)
    bbox = get_bbox_from_mask(mask, 0)
    data = np.vstack([crop_to_bbox(data[c], bbox)[None] for c in range(data.shape[0])])
    if seg is not None:
        seg = np.vstack([crop_to_bbox(seg[c], bbox)[None] for c in range(seg.shape[0])])
    mask = crop_to_bbox(mask, bbox)[None]
    if seg is not None:
        seg[(seg == 0) & (mask == 0)] = nonzero_label
    else:
        m = mask.astype(int)
        m[m == 0] = nonzero_label
        m[m > 0] = 0
        seg = m
    return data, seg, bbox


class ImageCropper(object):
    @staticmethod
    def crop(data, properties, seg=None):
        data, seg, bbox = crop_to_nonzero(data, seg, nonzero_label=-1)                # cropping.py:139-150
        properties["crop_bbox"] = bbox
        properties['classes'] = np.unique(seg)
        seg[seg < -1] = 0
        properties["size_after_cropping"] = data[0].shape
        return data, seg, properties

    @staticmethod
    def crop_from_list_of_files(data_files, seg_file=None):
        data, seg, properties = load_case_from_list_of_files(data_files, seg_file)
        return ImageCropper.crop(data, properties, seg)
