"""Intensity augmentations of nnU-Net's moreDA chain on the device (data_augmentation_moreDA.py:86-106): GaussianNoise,
GaussianBlur, BrightnessMultiplicative, ContrastAugmentation, SimulateLowResolution and the two GammaTransforms.  The elementwise
ones (+ per-channel min/max/mean/std) are torch glue (< 1 % of a training step); the two filtering / resampling transforms run on
HIP kernels: GaussianBlur on `mt_gaussian_blur_axis` (= scipy.ndimage.gaussian_filter), SimulateLowResolution on
`mt_downsample_seg_nearest` (order-0 resize) followed by `mt_spline_prefilter3` + `mt_affine_sample` (order-3 resize), each pinned
against its scipy formulation in tests/test_spatial_gpu.py.  The RANDOM-PARAMETER logic (which samples / channels, sigma and zoom
draws) restates batchgenerators' augment_* functions (third party, absent in the build container, batchgenerators>=0.23) from
the published source: parity UNPINNED."""
import numpy as np
import torch

from ... import _lib


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


def gaussian_filter_device(x, sigma):
    """scipy.ndimage.gaussian_filter(x[n, c], sigma[n, c], order=0) (mode 'reflect', truncate 4) for every (sample, channel) of a
    [N, C, D, H, W] device tensor; sigma <= 0 leaves that channel untouched.  Axis order D, H, W like scipy; every pass rounds to
    float32 like scipy does for float32 input."""
    assert x.is_cuda and x.dim() == 5 and x.dtype == torch.float32
    lib = _lib.load()
    st = torch.cuda.current_stream(x.device).cuda_stream
    N, C, D, H, W = (int(i) for i in x.shape)
    sg = torch.as_tensor(np.asarray(sigma, dtype=np.float32).reshape(N * C)).to(x.device)
    src = x.contiguous()
    t0, t1 = torch.empty_like(src), torch.empty_like(src)          # D: src -> t0, H: t0 -> t1, W: t1 -> t0 (the input is never written)
    for axis, (a, b) in enumerate(((src, t0), (t0, t1), (t1, t0))):
        _lib.check(lib.mt_gaussian_blur_axis(a.data_ptr(), b.data_ptr(), N * C, D, H, W, axis, sg.data_ptr(), st), 'gaussian_blur_axis')
    a = t0
    return a


class GaussianBlurDevice:
    """GaussianBlurTransform((0.5, 1.), different_sigma_per_channel=True, p_per_sample=0.2, p_per_channel=0.5)
    (data_augmentation_moreDA.py:87-88; batchgenerators augment_gaussian_blur): per chosen sample, every channel is blurred with
    probability p_per_channel with its own sigma ~ U(blur_sigma)."""

    def __init__(self, blur_sigma=(0.5, 1.), different_sigma_per_channel=True, p_per_channel=0.5, p_per_sample=0.2):
        self.sigma, self.per_channel, self.ppc, self.p = blur_sigma, different_sigma_per_channel, p_per_channel, p_per_sample

    def _draw(self):
        return self.sigma[0] if self.sigma[0] == self.sigma[1] else np.random.uniform(self.sigma[0], self.sigma[1])

    def __call__(self, data):
        N, C = int(data.shape[0]), int(data.shape[1])
        sg = np.zeros((N, C), dtype=np.float32)
        for b in range(N):
            if np.random.uniform() < self.p:
                shared = None if self.per_channel else self._draw()
                for c in range(C):
                    if np.random.uniform() <= self.ppc:
                        sg[b, c] = self._draw() if self.per_channel else shared
        if not (sg > 0).any():
            return data
        return gaussian_filter_device(data, sg)


def simulate_low_resolution_device(x, target_shape, planar=False):
    """resize(resize(x, target_shape, order=0), x.shape, order=3) with skimage's mode='edge', anti_aliasing=False for ONE channel
    volume x [D, H, W] (augment_linear_downsampling_scipy's body): nearest sampling at floor((o + 0.5) in/out), then the cubic
    B-spline resize at (o + 0.5) in/out - 0.5 behind 12 voxels of edge padding.  planar: axis 0 keeps its size (ignore_axes=(0,)),
    every slice is resized on its own."""
    from ... import ops
    from ...preprocessing.device_preprocessing import _zoom3
    shp = tuple(int(i) for i in x.shape)
    tgt = tuple(int(i) for i in target_shape)
    if tgt == shp:
        return x
    low = ops.downsample_seg_nearest(x[None, None].contiguous(), tgt)
    return _zoom3(low, shp, planar=planar and tgt[0] == shp[0])[0, 0]


class SimulateLowResolutionDevice:
    """SimulateLowResolutionTransform(zoom_range=(0.5, 1), per_channel=True, p_per_channel=0.5, order_downsample=0,
    order_upsample=3, p_per_sample=0.25, ignore_axes) (data_augmentation_moreDA.py:100-103)."""

    def __init__(self, zoom_range=(0.5, 1), per_channel=True, p_per_channel=0.5, p_per_sample=0.25, ignore_axes=None):
        self.zoom, self.per_channel, self.ppc, self.p, self.ignore = zoom_range, per_channel, p_per_channel, p_per_sample, ignore_axes

    def __call__(self, data):
        shp = np.array([int(i) for i in data.shape[2:]])
        planar = self.ignore is not None and tuple(self.ignore) == (0,)
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                target = None
                if not self.per_channel:
                    target = np.round(shp * np.random.uniform(self.zoom[0], self.zoom[1])).astype(int)
                for c in range(data.shape[1]):
                    if np.random.uniform() < self.ppc:
                        if self.per_channel:
                            target = np.round(shp * np.random.uniform(self.zoom[0], self.zoom[1])).astype(int)
                        t = target.copy()
                        if self.ignore is not None:
                            for i in self.ignore:
                                t[i] = shp[i]
                        data[b, c] = simulate_low_resolution_device(data[b, c], t, planar)
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
    """DataLoader3D batches (loader patch = basic_generator_patch_size) -> device -> SpatialTransform -> noise, blur, brightness,
    contrast, simulated low resolution, gamma (inverted), gamma -> mirror -> {'data', 'target' (one label map; the trainers build
    the pyramid and remove label -1 on the device), 'properties', 'keys'}: get_moreDA_augmentation's order
    (data_augmentation_moreDA.py:41-153) without the cascade transforms and the CPU worker pool."""

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
        ignore_axes = (0,) if params.get("dummy_2D") else None               # data_augmentation_moreDA.py:55-64
        self.color = [GaussianNoiseDevice(p_per_sample=0.1),
                      GaussianBlurDevice((0.5, 1.), different_sigma_per_channel=True, p_per_sample=0.2, p_per_channel=0.5),
                      BrightnessMultiplicativeDevice((0.75, 1.25), p_per_sample=0.15),
                      ContrastAugmentationDevice(p_per_sample=0.15),
                      SimulateLowResolutionDevice((0.5, 1), per_channel=True, p_per_channel=0.5, p_per_sample=0.25, ignore_axes=ignore_axes),
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
