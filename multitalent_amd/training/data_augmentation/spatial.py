"""Spatial augmentation on the device (second half).

`affine_sample` is scipy.ndimage.map_coordinates over an affine coordinate field (the compute core of batchgenerators'
SpatialTransform as nnU-Net configures it, data_augmentation_moreDA.py:66-80), on `mt_spline_prefilter3` / `mt_affine_sample`.
`SpatialTransformDevice` draws the per-sample rotation / scaling like batchgenerators' `augment_spatial` (third party, absent in
the build container: restated from its published source, batchgenerators>=0.23 — parity UNPINNED; the kernels themselves are
pinned against scipy in tests/test_spatial_gpu.py) and hands the matrices to the kernels; `MirrorTransformDevice` is the last
spatial step of the chain (data_augmentation_moreDA.py:108-109)."""
import numpy as np
import torch

from ... import _lib


def affine_sample(x, mats, out_shape, order, cval=0.0, is_seg=False, planar=False):
    """x: [N, C, D, H, W] float32 device tensor; mats: [N, 12] (row-major 3x3 M then the centre) so that output voxel o reads
    input coordinate M (o - (O-1)/2) + centre; order 0 / 1 / 3; is_seg with order 1 = batchgenerators' per-label rule."""
    assert x.is_cuda and x.dim() == 5 and x.dtype == torch.float32
    lib = _lib.load()
    st = torch.cuda.current_stream(x.device).cuda_stream
    N, C, D, H, W = (int(i) for i in x.shape)
    mats = torch.as_tensor(np.asarray(mats, dtype=np.float32)).to(x.device).contiguous() if not torch.is_tensor(mats) else mats.float().contiguous()
    assert tuple(mats.shape) == (N, 12)
    if order == 3 and not is_seg:
        src = x.clone()
        _lib.check(lib.mt_spline_prefilter3(src.data_ptr(), N * C, D, H, W, 3 if planar else 7, st), 'spline_prefilter3')
        mode = 3
    elif order == 1:
        src, mode = x.contiguous(), (11 if is_seg else 1)
    elif order == 0:
        src, mode = x.contiguous(), 0
    else:
        raise NotImplementedError("interpolation order %d (is_seg=%s) is not on the device path" % (order, is_seg))
    out = torch.empty((N, C) + tuple(int(i) for i in out_shape), dtype=torch.float32, device=x.device)
    _lib.check(lib.mt_affine_sample(src.data_ptr(), N, C, D, H, W, out.data_ptr(), out.shape[2], out.shape[3], out.shape[4],
                                    mats.data_ptr(), mode, float(cval), 1 if planar else 0, st), 'affine_sample')
    return out


def rotation_matrix_3d(angle_x, angle_y, angle_z):
    """batchgenerators rotate_coords_3d: coords(row vectors) @ (Rx @ Ry @ Rz), i.e. x' = (Rx Ry Rz)^T x."""
    cx, sx, cy, sy, cz, sz = np.cos(angle_x), np.sin(angle_x), np.cos(angle_y), np.sin(angle_y), np.cos(angle_z), np.sin(angle_z)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (rx @ ry @ rz).T


class SpatialTransformDevice:
    """Rotation + scaling + (centre | random) crop to `patch_size` in one resampling pass per sample.  Parameter names and
    defaults follow batchgenerators.transforms.spatial_transforms.SpatialTransform as called by get_moreDA_augmentation;
    elastic deformation (off in nnU-Net's 3D defaults) is not on this path."""

    def __init__(self, patch_size, patch_center_dist_from_border=30, do_elastic_deform=False, do_rotation=True,
                 angle_x=(0, 2 * np.pi), angle_y=(0, 2 * np.pi), angle_z=(0, 2 * np.pi), do_scale=True, scale=(0.75, 1.25),
                 border_mode_data='constant', border_cval_data=0, order_data=3, border_mode_seg='constant', border_cval_seg=0,
                 order_seg=1, random_crop=False, p_el_per_sample=1, p_scale_per_sample=1, p_rot_per_sample=1,
                 independent_scale_for_each_axis=False, p_rot_per_axis=1, p_independent_scale_per_axis=1, dummy_2d=False):
        self.dummy_2d = dummy_2d              # Convert3DTo2DTransform around the transform (data_augmentation_moreDA.py:57-82)
        if do_elastic_deform:
            raise NotImplementedError("elastic deformation is not on the device path (off in nnU-Net's 3D defaults)")
        if border_mode_data != 'constant' or border_mode_seg != 'constant':
            raise NotImplementedError("only constant borders (nnU-Net's setting) are on the device path")
        self.patch_size = tuple(int(i) for i in patch_size)
        self.dist = patch_center_dist_from_border
        self.do_rotation, self.angles = do_rotation, (angle_x, angle_y, angle_z)
        self.do_scale, self.scale = do_scale, scale
        self.cval_data, self.cval_seg, self.order_data, self.order_seg = border_cval_data, border_cval_seg, order_data, order_seg
        self.random_crop = random_crop
        self.p_scale, self.p_rot, self.p_rot_axis = p_scale_per_sample, p_rot_per_sample, p_rot_per_axis
        self.indep, self.p_indep = independent_scale_for_each_axis, p_independent_scale_per_axis

    def draw(self, in_shape):
        """one sample's (M, centre, modified) with numpy's global random stream, in augment_spatial's call order."""
        m, modified = np.eye(3), False
        nd = 2 if self.dummy_2d else 3
        if self.do_rotation and np.random.uniform() < self.p_rot:
            if self.dummy_2d:                 # 2D: one angle (angle_x), coords @ [[c, -s], [s, c]]
                a = np.random.uniform(*self.angles[0]) if np.random.uniform() <= self.p_rot_axis else 0
                m[1:, 1:] = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T
                modified = True
            else:
                a = [np.random.uniform(lo, hi) if np.random.uniform() <= self.p_rot_axis else 0 for lo, hi in self.angles]
                m, modified = rotation_matrix_3d(*a), True
        if self.do_scale and np.random.uniform() < self.p_scale:
            def one():
                if np.random.random() < 0.5 and self.scale[0] < 1:
                    return np.random.uniform(self.scale[0], 1)
                return np.random.uniform(max(self.scale[0], 1), self.scale[1])
            if self.indep and np.random.uniform() < self.p_indep:
                sc = np.array([one() for _ in range(nd)])
            else:
                sc = np.full(nd, one())
            if self.dummy_2d:
                sc = np.concatenate([[1.0], sc])
            m, modified = sc[:, None] * m, True            # scale_coords: coords[d] *= scale[d] after the rotation
        if self.random_crop:
            dist = self.dist if isinstance(self.dist, (list, tuple, np.ndarray)) else [self.dist] * 3
            ctr = np.array([np.random.uniform(dist[d], in_shape[d] - dist[d]) for d in range(3)])
        else:
            ctr = np.array([in_shape[d] / 2. - 0.5 for d in range(3)])
        return m, ctr, modified

    def __call__(self, data, seg=None):
        """data [B, C, D, H, W], seg [B, Cs, D, H, W] device float tensors -> patch-sized tensors."""
        B = data.shape[0]
        in_shape = tuple(int(i) for i in data.shape[2:])
        mats = np.zeros((B, 12), dtype=np.float32)
        for b in range(B):
            m, ctr, modified = self.draw(in_shape)
            if not modified and not self.random_crop:
                # batchgenerators centre-crops without interpolation: integer lower corner (s - p) // 2
                ctr = np.array([(in_shape[d] - self.patch_size[d]) // 2 + (self.patch_size[d] - 1) / 2. for d in range(3)])
            elif not modified:
                ctr = np.round(ctr - (np.array(self.patch_size) - 1) / 2.) + (np.array(self.patch_size) - 1) / 2.
            mats[b, :9], mats[b, 9:] = m.reshape(-1), ctr
        if self.dummy_2d:
            assert in_shape[0] == self.patch_size[0], "dummy 2D: the loader patch keeps the slice axis (nnUNetTrainerV2.py:374-379)"
            mats[:, 9] = 0
        out = affine_sample(data, mats, self.patch_size, self.order_data, self.cval_data, planar=self.dummy_2d)
        out_seg = None
        if seg is not None:
            out_seg = affine_sample(seg, mats, self.patch_size, self.order_seg, self.cval_seg, is_seg=True, planar=self.dummy_2d)
        return out, out_seg


class MirrorTransformDevice:
    """batchgenerators MirrorTransform(axes): per sample and axis, flip data and seg with probability 0.5."""

    def __init__(self, axes=(0, 1, 2), p_per_sample=1.0):
        self.axes, self.p = tuple(axes), p_per_sample

    def __call__(self, data, seg=None):
        for b in range(data.shape[0]):
            if np.random.uniform() < self.p:
                dims = [a + 1 for a in self.axes if np.random.uniform() < 0.5]       # dims of the [C, D, H, W] sample
                if dims:
                    data[b] = torch.flip(data[b], dims)
                    if seg is not None:
                        seg[b] = torch.flip(seg[b], dims)
        return data, seg


def get_patch_size(final_patch_size, rot_x, rot_y, rot_z, scale_range):
    """default_data_augmentation.py:111-131: the loader's patch must contain the final patch under the largest rotation about each
    axis (capped at 90 degrees) and the smallest scale factor."""
    rot = [min(90 / 360 * 2. * np.pi, max(np.abs(r)) if isinstance(r, (tuple, list)) else r) for r in (rot_x, rot_y, rot_z)]
    coords = np.array(final_patch_size, dtype=float)
    final_shape = np.copy(coords)
    if len(coords) == 3:
        for k in range(3):
            a = [0, 0, 0]; a[k] = rot[k]
            final_shape = np.max(np.vstack((np.abs(rotation_matrix_3d(*a) @ coords), final_shape)), 0)
    else:
        c, s_ = np.cos(rot[0]), np.sin(rot[0])
        final_shape = np.max(np.vstack((np.abs(np.array([[c, -s_], [s_, c]]).T @ coords), final_shape)), 0)
    final_shape /= min(scale_range)
    return final_shape.astype(int)
