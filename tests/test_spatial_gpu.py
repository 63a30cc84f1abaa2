"""Device spatial augmentation kernels (second half) against scipy.ndimage — the library batchgenerators'
SpatialTransform computes with: spline prefilter vs spline_filter(order 3, mode 'constant'), affine sampling vs
map_coordinates(order 0/1/3, mode 'constant', cval) on random rotations + anisotropic scalings + off-centre crops that run off
the volume, and batchgenerators' per-label order-1 rule for segmentations restated with map_coordinates.  Tolerance 2e-5 of the
data range (float32 coefficients and weights vs scipy's float64)."""
import numpy as np
import pytest
import torch
from scipy import ndimage

pytestmark = pytest.mark.gpu


def _coords(m, ctr, out_shape):
    g = np.meshgrid(*[np.arange(s) - (s - 1) / 2. for s in out_shape], indexing='ij')
    g = np.stack([x.ravel() for x in g])
    return (m @ g + ctr[:, None]).reshape((3,) + tuple(out_shape))


def _cases(rs, n, in_shape):
    from multitalent_amd.training.data_augmentation.spatial import rotation_matrix_3d
    mats = np.zeros((n, 12), np.float32)
    for i in range(n):
        m = rotation_matrix_3d(*rs.uniform(-0.6, 0.6, 3)) if i else np.eye(3)
        m = rs.uniform(0.7, 1.4, 3)[:, None] * m if i > 1 else m
        ctr = np.array(in_shape) / 2. - 0.5 + (rs.uniform(-4, 4, 3) if i > 2 else 0)
        mats[i, :9], mats[i, 9:] = m.reshape(-1), ctr
    return mats


def test_spline_prefilter_matches_scipy(dev):
    from multitalent_amd import _lib
    rs = np.random.RandomState(0)
    x = rs.randn(3, 9, 14, 21).astype(np.float32)
    x[1, :, :, 0] += 5                       # something at the border
    t = torch.from_numpy(x).to(dev)
    _lib.check(_lib.load().mt_spline_prefilter3(t.data_ptr(), 3, 9, 14, 21, 7, torch.cuda.current_stream().cuda_stream), 'prefilter')
    ref = np.stack([ndimage.spline_filter(c.astype(np.float64), 3, mode='constant') for c in x])
    assert np.abs(t.cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("order", [3, 1, 0])
def test_affine_sample_matches_map_coordinates(dev, order):
    from multitalent_amd.training.data_augmentation.spatial import affine_sample
    rs = np.random.RandomState(1)
    in_shape, out_shape, N, C = (20, 26, 30), (12, 16, 18), 5, 2
    x = rs.randn(N, C, *in_shape).astype(np.float32)
    mats = _cases(rs, N, in_shape)
    if order == 0:                           # keep the coordinates away from the .5 rounding ties of nearest sampling
        mats[:, 9:] += 0.013
    got = affine_sample(torch.from_numpy(x).to(dev), mats, out_shape, order, cval=-3.0).cpu().numpy()
    for n in range(N):
        co = _coords(mats[n, :9].reshape(3, 3).astype(np.float64), mats[n, 9:].astype(np.float64), out_shape)
        for c in range(C):
            ref = ndimage.map_coordinates(x[n, c].astype(np.float64), co, order=order, mode='constant', cval=-3.0)
            err = np.abs(got[n, c] - ref)
            if order == 0:
                assert (err > 0).mean() < 1e-3
            else:
                # an output whose coordinate lies within 1e-4 of the volume boundary may fall on the other side in float32
                edge = np.zeros(out_shape, bool)
                for d in range(3):
                    edge |= (np.abs(co[d]) < 1e-4) | (np.abs(co[d] - (in_shape[d] - 1)) < 1e-4)
                assert err[~edge].max() < 2e-5 * 8, (order, n, c, err[~edge].max())
    assert (got == -3.0).any() and (got != -3.0).any()


@pytest.mark.parametrize("order", [3, 1])
def test_planar_sampling_matches_2d_map_coordinates(dev, order):
    """nnU-Net's "dummy 2D" augmentation (Convert3DTo2DTransform around the SpatialTransform): every D slice is an independent 2D
    image — per-slice 2D prefilter and 2D map_coordinates."""
    from multitalent_amd.training.data_augmentation.spatial import affine_sample
    rs = np.random.RandomState(5)
    D, in_hw, out_hw, N = 6, (26, 30), (16, 18), 3
    x = rs.randn(N, 1, D, *in_hw).astype(np.float32)
    mats = np.zeros((N, 12), np.float32)
    for n in range(N):
        a = rs.uniform(-0.5, 0.5) if n else 0.0
        r2 = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T * (rs.uniform(0.7, 1.4) if n > 1 else 1.0)
        m = np.eye(3); m[1:, 1:] = r2
        mats[n, :9] = m.reshape(-1); mats[n, 9:] = [0, in_hw[0] / 2. - 0.5, in_hw[1] / 2. - 0.5]
    got = affine_sample(torch.from_numpy(x).to(dev), mats, (D,) + out_hw, order, cval=0.0, planar=True).cpu().numpy()
    for n in range(N):
        g = np.meshgrid(*[np.arange(s) - (s - 1) / 2. for s in out_hw], indexing='ij')
        co = (mats[n, :9].reshape(3, 3)[1:, 1:].astype(np.float64) @ np.stack([v.ravel() for v in g]) + mats[n, 10:, None].astype(np.float64)).reshape((2,) + out_hw)
        for d in range(D):
            ref = ndimage.map_coordinates(x[n, 0, d].astype(np.float64), co, order=order, mode='constant', cval=0.0)
            edge = np.zeros(out_hw, bool)
            for k in range(2):
                edge |= (np.abs(co[k]) < 1e-4) | (np.abs(co[k] - (in_hw[k] - 1)) < 1e-4)
            assert np.abs(got[n, 0, d] - ref)[~edge].max() < 1e-4


def test_segmentation_rule_matches_batchgenerators_restatement(dev):
    """interpolate_img(seg, coords, order=1, 'constant', cval, is_seg=True): result = zeros; for c in unique(seg) ascending:
    result[map_coordinates((seg == c).astype(float), coords, order=1, mode='constant', cval=cval) >= 0.5] = c."""
    from multitalent_amd.training.data_augmentation.spatial import affine_sample
    rs = np.random.RandomState(2)
    in_shape, out_shape, N = (18, 22, 24), (12, 14, 16), 4
    coarse = rs.randint(-1, 4, size=(N, 1, 5, 6, 6)).astype(np.float32)
    seg = np.kron(coarse, np.ones((1, 1, 4, 4, 4), np.float32))[:, :, :18, :22, :24]
    mats = _cases(rs, N, in_shape)
    got = affine_sample(torch.from_numpy(np.ascontiguousarray(seg)).to(dev), mats, out_shape, 1, cval=-1.0, is_seg=True).cpu().numpy()
    bad = total = 0
    for n in range(N):
        co = _coords(mats[n, :9].reshape(3, 3).astype(np.float64), mats[n, 9:].astype(np.float64), out_shape)
        ref = np.zeros(out_shape, np.float32)
        fragile = np.zeros(out_shape, bool)
        for c in np.unique(seg[n, 0]):
            r = ndimage.map_coordinates((seg[n, 0] == c).astype(float), co, order=1, mode='constant', cval=-1.0)
            ref[r >= 0.5] = c
            fragile |= np.abs(r - 0.5) < 1e-4
        bad += (got[n, 0][~fragile] != ref[~fragile]).sum(); total += (~fragile).sum()
    assert bad == 0 and total > 0.9 * N * np.prod(out_shape)


def test_spatial_transform_shapes_and_identity(dev):
    from multitalent_amd.training.data_augmentation.spatial import SpatialTransformDevice, MirrorTransformDevice
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randn(2, 1, 20, 30, 30).astype(np.float32)).to(dev)
    s = torch.from_numpy(rs.randint(0, 3, size=(2, 1, 20, 30, 30)).astype(np.float32)).to(dev)
    ident = SpatialTransformDevice((12, 20, 20), do_rotation=False, do_scale=False, border_cval_seg=-1)
    d, g = ident(x, s)
    assert torch.allclose(d, x[:, :, 4:16, 5:25, 5:25], atol=1e-5) and torch.equal(g, s[:, :, 4:16, 5:25, 5:25])      # pure centre crop
    np.random.seed(0)
    aug = SpatialTransformDevice((12, 20, 20), angle_x=(-0.5, 0.5), angle_y=(-0.5, 0.5), angle_z=(-0.5, 0.5), scale=(0.7, 1.4),
                                 p_rot_per_sample=1.0, p_scale_per_sample=1.0, border_cval_seg=-1)
    d, g = aug(x, s)
    assert d.shape == (2, 1, 12, 20, 20) and g.shape == (2, 1, 12, 20, 20) and torch.isfinite(d).all()
    assert set(np.unique(g.cpu().numpy())) <= {-1.0, 0.0, 1.0, 2.0}
    d2, g2 = MirrorTransformDevice((0, 1, 2))(d.clone(), g.clone())
    assert d2.shape == d.shape and float(d2.abs().sum()) == pytest.approx(float(d.abs().sum()), rel=1e-5)


def test_gaussian_blur_matches_scipy(dev):
    """GaussianBlurTransform's compute core = scipy.ndimage.gaussian_filter(channel, sigma, order=0) (mode 'reflect', truncate 4):
    sigma over the transform's range (0.5, 1) and beyond (radius up to 12, wider than a short axis: multiple reflections), untouched
    channels (sigma 0) bit-identical."""
    from multitalent_amd.training.data_augmentation.color import gaussian_filter_device
    rs = np.random.RandomState(3)
    x = (rs.randn(2, 3, 7, 19, 26) * 3 + 1).astype(np.float32)
    sg = np.array([[0.5, 0.0, 1.0], [0.73, 3.0, 0.0]], dtype=np.float32)
    t = torch.from_numpy(x).to(dev)
    got = gaussian_filter_device(t, sg).cpu().numpy()
    assert np.array_equal(t.cpu().numpy(), x)                         # the input is not modified
    for n in range(2):
        for c in range(3):
            if sg[n, c] <= 0:
                assert np.array_equal(got[n, c], x[n, c])
            else:
                ref = ndimage.gaussian_filter(x[n, c], float(sg[n, c]), order=0)
                assert np.abs(got[n, c] - ref).max() < 2e-6 * np.abs(x).max(), (n, c)


@pytest.mark.parametrize("shape,target,planar", [((12, 30, 26), (7, 17, 15), False), ((9, 33, 20), (9, 21, 11), True),
                                                 ((16, 16, 16), (8, 8, 8), False), ((10, 21, 18), (10, 21, 18), False),
                                                 ((11, 20, 23), (10, 19, 22), False)])
def test_simulate_low_resolution_matches_scipy(dev, shape, target, planar):
    """SimulateLowResolutionTransform's body (augment_linear_downsampling_scipy): resize(order 0) then resize(order 3) with
    mode='edge', anti_aliasing=False = scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) (skimage's delegate).  Tolerance
    2e-5 of the range (fp32 spline coefficients; the prefilter's boundary rule behind 12 voxels of edge padding)."""
    from multitalent_amd.training.data_augmentation.color import simulate_low_resolution_device
    rs = np.random.RandomState(4)
    x = ndimage.gaussian_filter(rs.randn(*shape), 0.7).astype(np.float32) * 4
    zoom = lambda a, new, order: ndimage.zoom(a.astype(float), [n / o for n, o in zip(new, a.shape)], order=order, mode='nearest', grid_mode=True)
    ref = zoom(zoom(x, target, 0), shape, 3) if tuple(target) != tuple(shape) else x
    got = simulate_low_resolution_device(torch.from_numpy(x).to(dev), target, planar).cpu().numpy()
    assert got.shape == tuple(shape)
    assert np.abs(got - ref).max() < 2e-5 * (x.max() - x.min()), np.abs(got - ref).max()


def test_blur_and_lowres_transforms_draw_like_the_reference_chain(dev):
    """The two transform objects at their positions in MoreDADeviceAugmenter: probabilities 1 -> every channel changes; probabilities
    0 -> the batch passes through untouched (same storage)."""
    from multitalent_amd.training.data_augmentation.color import GaussianBlurDevice, SimulateLowResolutionDevice
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(2, 2, 8, 20, 24).astype(np.float32)).to(dev)
    np.random.seed(0)
    assert GaussianBlurDevice(p_per_sample=0.0)(x) is x and SimulateLowResolutionDevice(p_per_sample=0.0)(x) is x
    b = GaussianBlurDevice(p_per_sample=1.0, p_per_channel=1.0)(x.clone())
    assert float((b - x).abs().amax(dim=(2, 3, 4)).min()) > 1e-2 and float(b.std()) < float(x.std())
    l = SimulateLowResolutionDevice(p_per_sample=1.0, p_per_channel=1.0, ignore_axes=(0,))(x.clone())
    assert l.shape == x.shape and float((l - x).abs().amax(dim=(2, 3, 4)).min()) > 1e-2
