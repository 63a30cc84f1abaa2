"""Intensity augmentations of nnU-Net's moreDA chain on the device (data_augmentation_moreDA.py:86-106): GaussianNoise,
BrightnessMultiplicative, ContrastAugmentation and the two GammaTransforms — elementwise passes plus per-channel min/max/mean/std,
written with torch elementwise ops (glue: < 1 % of a training step).  They restate batchgenerators' augment_* functions (third
party, absent in the build container, batchgenerators>=0.23) from the published source: parity UNPINNED.  Not on this path:
GaussianBlurTransform and SimulateLowResolutionTransform (separable filtering / resampling passes; the reference applies them to
20-25 % of the samples)."""
import numpy as np
import torch


def _range_pick(lo, hi):
    """batchgenerators' two-sided draw: below 1 with probability 1/2 when the range allows it."""
    if np.random.random() < 0.5 and lo < 1:
        return np.random.uniform(lo, 1)
    return np.random.uniform(max(lo, 1), hi)


class GaussianNoiseDevice:
    def __init__(self, noise_variance=(0, 0.1), p_per_sample=0.1):
        self.var, self.p = noise_variance, p_per_sample

    def __call__(self, data):
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                v = self.var[0] if self.var[0] == self.var[1] else np.random.uniform(self.var[0], self.var[1])
                data[b] += torch.randn_like(data[b]) * v          # augment_gaussian_noise passes the "variance" as the std
        return data


class BrightnessMultiplicativeDevice:
    def __init__(self, multiplier_range=(0.75, 1.25), per_channel=True, p_per_sample=0.15):
        self.r, self.per_channel, self.p = multiplier_range, per_channel, p_per_sample

    def __call__(self, data):
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                if self.per_channel:
                    for c in range(data.shape[1]):
                        data[b, c] *= np.random.uniform(self.r[0], self.r[1])
                else:
                    data[b] *= np.random.uniform(self.r[0], self.r[1])
        return data


class ContrastAugmentationDevice:
    def __init__(self, contrast_range=(0.75, 1.25), preserve_range=True, per_channel=True, p_per_sample=0.15):
        self.r, self.preserve, self.per_channel, self.p = contrast_range, preserve_range, per_channel, p_per_sample

    def _one(self, x):
        f = _range_pick(self.r[0], self.r[1])
        mn = x.mean()
        lo, hi = (x.min(), x.max()) if self.preserve else (None, None)
        x.sub_(mn).mul_(f).add_(mn)
        if self.preserve:
            x.clamp_(min=float(lo), max=float(hi))

    def __call__(self, data):
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                if self.per_channel:
                    for c in range(data.shape[1]):
                        self._one(data[b, c])
                else:
                    self._one(data[b])
        return data


class GammaDevice:
    def __init__(self, gamma_range=(0.7, 1.5), invert_image=False, per_channel=True, retain_stats=True, p_per_sample=0.3, epsilon=1e-7):
        self.r, self.invert, self.per_channel, self.retain, self.p, self.eps = gamma_range, invert_image, per_channel, retain_stats, p_per_sample, epsilon

    def _one(self, x):
        if self.retain:
            mn, sd = float(x.mean()), float(x.std(unbiased=False))
        g = _range_pick(self.r[0], self.r[1])
        lo = float(x.min())
        rng = float(x.max()) - lo
        x.sub_(lo).div_(rng + self.eps).pow_(g).mul_(rng).add_(lo)
        if self.retain:
            x.sub_(float(x.mean()))
            x.div_(float(x.std(unbiased=False)) + 1e-8).mul_(sd).add_(mn)

    def __call__(self, data):
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                if self.invert:
                    data[b].neg_()
                if self.per_channel:
                    for c in range(data.shape[1]):
                        self._one(data[b, c])
                else:
                    self._one(data[b])
                if self.invert:
                    data[b].neg_()
        return data


class MoreDADeviceAugmenter:
    """DataLoader3D batches (loader patch = basic_generator_patch_size) -> device -> SpatialTransform -> noise, brightness,
    contrast, gamma (inverted), gamma -> mirror -> {'data', 'target' (one label map; the trainers build the pyramid and remove
    label -1 on the device), 'properties', 'keys'}: get_moreDA_augmentation's order (data_augmentation_moreDA.py:41-153) without
    blur / simulated low resolution, the cascade transforms and the CPU worker pool."""

    def __init__(self, loader, patch_size, params, device, border_val_seg=-1, order_seg=1, order_data=3):
        from .spatial import MirrorTransformDevice, SpatialTransformDevice
        self.loader, self.device = loader, device
        self.spatial = SpatialTransformDevice(
            patch_size, patch_center_dist_from_border=None, do_elastic_deform=params.get("do_elastic"),
            do_rotation=params.get("do_rotation"), angle_x=params.get("rotation_x"), angle_y=params.get("rotation_y"),
            angle_z=params.get("rotation_z"), p_rot_per_axis=params.get("rotation_p_per_axis"), do_scale=params.get("do_scaling"),
            scale=params.get("scale_range"), border_mode_data=params.get("border_mode_data"), border_cval_data=0, order_data=order_data,
            border_mode_seg="constant", border_cval_seg=border_val_seg, order_seg=order_seg, random_crop=params.get("random_crop"),
            p_el_per_sample=params.get("p_eldef"), p_scale_per_sample=params.get("p_scale"), p_rot_per_sample=params.get("p_rot"),
            independent_scale_for_each_axis=params.get("independent_scale_factor_for_each_axis"), dummy_2d=bool(params.get("dummy_2D")))
        self.color = [GaussianNoiseDevice(p_per_sample=0.1), BrightnessMultiplicativeDevice((0.75, 1.25), p_per_sample=0.15),
                      ContrastAugmentationDevice(p_per_sample=0.15),
                      GammaDevice(params.get("gamma_range"), True, True, params.get("gamma_retain_stats"), 0.1)]
        if params.get("do_gamma"):
            self.color.append(GammaDevice(params.get("gamma_range"), False, True, params.get("gamma_retain_stats"), params["p_gamma"]))
        self.mirror = MirrorTransformDevice(params.get("mirror_axes")) if params.get("do_mirror") else None

    def __iter__(self):
        return self

    def __next__(self):
        b = next(self.loader)
        data = torch.from_numpy(b['data']).to(self.device, non_blocking=True)
        seg = torch.from_numpy(b['seg'][:, :1]).to(self.device, non_blocking=True)
        data, seg = self.spatial(data, seg)
        for t in self.color:
            data = t(data)
        if self.mirror is not None:
            data, seg = self.mirror(data, seg)
        return {'data': data, 'target': seg, 'properties': b['properties'], 'keys': b['keys']}


def default_3d_augmentation_params():
    """default_3D_augmentation_params (default_data_augmentation.py:39-92) with nnUNetTrainerV2.setup_DA_params's overrides
    (nnUNetTrainerV2.py:352-389): rotations +-30 degrees, scale (0.7, 1.4), no elastic deformation."""
    r = 30. / 360 * 2. * np.pi
    return {"do_elastic": False, "p_eldef": 0.2, "do_scaling": True, "scale_range": (0.7, 1.4),
            "independent_scale_factor_for_each_axis": False, "p_scale": 0.2, "do_rotation": True, "rotation_x": (-r, r),
            "rotation_y": (-r, r), "rotation_z": (-r, r), "rotation_p_per_axis": 1, "p_rot": 0.2, "random_crop": False,
            "do_gamma": True, "gamma_retain_stats": True, "gamma_range": (0.7, 1.5), "p_gamma": 0.3, "do_mirror": True,
            "mirror_axes": (0, 1, 2), "dummy_2D": False, "border_mode_data": "constant"}
