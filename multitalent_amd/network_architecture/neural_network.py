"""SegmentationNetwork — base class with the sliding-window inference driver.
(filled in by multitalent_amd.inference; see predict_3D below)"""
import numpy as np
import torch
from torch import nn


class NeuralNetwork(nn.Module):
    def get_device(self):
        p = next(self.parameters())
        return "cpu" if p.device.type == "cpu" else p.device.index

    def set_device(self, device):
        if device == "cpu":
            self.cpu()
        else:
            self.cuda(device)


class SegmentationNetwork(NeuralNetwork):
    def __init__(self):
        super().__init__()
        self.input_shape_must_be_divisible_by = None
        self.conv_op = None
        self.num_classes = None
        self.inference_apply_nonlin = lambda x: x
        self._gaussian_3d = self._patch_size_for_gaussian_3d = None

    def predict_3D(self, x, do_mirroring, mirror_axes=(0, 1, 2), use_sliding_window=False, step_size=0.5,
                   patch_size=None, regions_class_order=None, use_gaussian=False, pad_border_mode="constant",
                   pad_kwargs=None, all_in_gpu=False, verbose=True, mixed_precision=True, tile_shard=None,
                   return_device_tensors=False):
        """Reference signature (neural_network.py:73-76) + the two keyword extensions of inference.sliding_window.predict_3D."""
        from ..inference.sliding_window import predict_3D
        return predict_3D(self, x, do_mirroring, mirror_axes, use_sliding_window, step_size, patch_size,
                          regions_class_order, use_gaussian, pad_border_mode, pad_kwargs, all_in_gpu, verbose,
                          mixed_precision, tile_shard=tile_shard, return_device_tensors=return_device_tensors)

    @staticmethod
    def _compute_steps_for_sliding_window(patch_size, image_size, step_size):
        from ..inference.sliding_window import compute_steps_for_sliding_window
        return compute_steps_for_sliding_window(patch_size, image_size, step_size)

    @staticmethod
    def _get_gaussian(patch_size, sigma_scale=1. / 8):
        from ..inference.sliding_window import get_gaussian
        return get_gaussian(patch_size, sigma_scale)
