"""Sliding-window inference on the HIP engine — SegmentationNetwork.predict_3D and its tiled 3D path
(reference neural_network.py:73-163, 246-285, 287-428, 502-591).

What changes relative to the reference (results do not): the mirror flips of the PREDICTION are index arithmetic inside
mt_flip_accumulate (fused with sigmoid/softmax and the 1/8 mean), the Gaussian-weighted overlap-add runs on the device
into an aggregate that lives in HBM for the whole volume (the reference copies every 333 MB tile to the host and adds it
in numpy), and the final divide + threshold/argmax is one kernel.  Accumulation order per voxel is the reference's
(tiles in x -> y -> z loop order, fp32), so probabilities agree to rounding and masks bit-exactly away from ties.
With tile sharding (rank, world) each process handles a contiguous run of tiles and the aggregates are summed with one
RCCL all_reduce — functionality the reference does not have (it only strides CASES across processes,
predict_MultiTalent.py:362)."""
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..ops import Act


def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """neural_network.py:262-285."""
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be as large or larger than patch_size"
    assert 0 < step_size <= 1, 'step_size must be larger than 0 and smaller or equal to 1'
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step = image_size[dim] - patch_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def get_gaussian(patch_size, sigma_scale=1. / 8):
    """neural_network.py:246-259 (scipy gaussian_filter of a centred impulse, normalised, zeros -> min)."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
    g = (g / np.max(g) * 1).astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_nd_image(image, new_shape, mode="constant", kwargs=None):
    """Published semantics of batchgenerators.augmentations.utils.pad_nd_image (third party, batchgenerators>=0.23):
    pad the trailing axes to max(new_shape, shape), below = diff // 2, above = diff // 2 + diff % 2."""
    kwargs = {'constant_values': 0} if kwargs is None else kwargs
    old = np.array(image.shape[-len(new_shape):])
    new = np.array([max(n, o) for n, o in zip(new_shape, old)])
    diff = new - old
    below, above = diff // 2, diff // 2 + diff % 2
    pads = [[0, 0]] * (image.ndim - len(new_shape)) + [list(i) for i in zip(below, above)]
    res = np.pad(image, pads, mode, **kwargs) if diff.any() else image
    pads = np.array(pads)
    pads[:, 1] = np.array(res.shape) - pads[:, 1]
    return res, tuple(slice(*i) for i in pads)


def _nonlin_code(net):
    f = getattr(net, 'inference_apply_nonlin', None)
    if isinstance(f, nn.Sigmoid):
        return 1
    name = getattr(f, '__name__', '')
    if name == 'softmax_helper' or isinstance(f, nn.Softmax):
        return 2
    if f is None or name == '<lambda>':
        # the base class default is the identity lambda
        try:
            t = torch.tensor([[1.0, 2.0]])
            if torch.equal(f(t), t):
                return 0
        except Exception:
            pass
    raise NotImplementedError("inference_apply_nonlin must be nn.Sigmoid(), softmax_helper or identity for the fused path")


def predict_3D(net, x, do_mirroring, mirror_axes=(0, 1, 2), use_sliding_window=False, step_size=0.5, patch_size=None,
               regions_class_order=None, use_gaussian=False, pad_border_mode="constant", pad_kwargs=None, all_in_gpu=False,
               verbose=True, mixed_precision=True, tile_shard=None, return_device_tensors=False):
    """Signature of SegmentationNetwork.predict_3D (neural_network.py:73-76) + `tile_shard=(rank, world)`.
    x: np.ndarray [C, X, Y, Z].  Returns (seg [X,Y,Z], probabilities [num_classes, X, Y, Z]) as numpy."""
    assert step_size <= 1, 'step_size must be smaller than 1. Otherwise there will be a gap between consecutive predictions'
    pad_kwargs = {'constant_values': 0} if pad_kwargs is None else pad_kwargs
    if len(mirror_axes):
        if max(mirror_axes) > 2:
            raise ValueError("mirror axes. duh")
    if net.training:
        print('WARNING! Network is in train mode during inference. This may be intended, or not...')
    assert len(x.shape) == 4, "data must have shape (c,x,y,z)"
    if not use_sliding_window:
        raise NotImplementedError("fully convolutional (non-tiled) 3D prediction is not on the north-star path; "
                                  "predict_MultiTalent always uses the sliding window (predict_MultiTalent.py:222-233)")
    assert patch_size is not None, "patch_size cannot be None for tiled prediction"
    dev = next(net.parameters()).device
    if dev.type != 'cuda':
        if not torch.cuda.is_available():
            raise RuntimeError("multitalent_amd inference needs a HIP device (no CPU fallback)")
        dev = torch.device('cuda', torch.cuda.current_device())
    patch_size = tuple(int(i) for i in patch_size)
    num_classes = net.num_classes
    nonlin = _nonlin_code(net)

    if torch.is_tensor(x) and x.is_cuda:
        # device-resident input (e.g. from preprocessing.device_preprocessing): pad_nd_image's constant padding with F.pad
        if pad_border_mode != 'constant':
            raise NotImplementedError("device-resident volumes are padded with pad_border_mode='constant' only")
        old = [int(i) for i in x.shape[1:]]
        diff = [max(p, o) - o for p, o in zip(patch_size, old)]
        below = [d // 2 for d in diff]
        above = [d // 2 + d % 2 for d in diff]
        data = x.float()
        if any(diff):
            data = torch.nn.functional.pad(data, (below[2], above[2], below[1], above[1], below[0], above[0]), mode='constant',
                                           value=float(pad_kwargs.get('constant_values', 0)))
        slicer = [slice(None)] + [slice(b, b + o) for b, o in zip(below, old)]
        dev = x.device
    else:
        data, slicer = pad_nd_image(np.asarray(x, dtype=np.float32), patch_size, pad_border_mode, pad_kwargs)
    shp = tuple(data.shape)
    steps = compute_steps_for_sliding_window(patch_size, shp[1:], step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    if verbose:
        print("data shape:", shp, "patch size:", patch_size, "steps (x, y, and z):", steps, "number of tiles:", num_tiles)
    if use_gaussian and num_tiles > 1:
        if getattr(net, '_gaussian_3d', None) is None or net._patch_size_for_gaussian_3d != patch_size:
            net._gaussian_3d_host = get_gaussian(patch_size, 1. / 8)
            net._gaussian_3d = torch.from_numpy(net._gaussian_3d_host).to(dev)
            net._patch_size_for_gaussian_3d = patch_size
        gaussian = net._gaussian_3d.to(dev)
    else:
        gaussian = torch.ones(patch_size, dtype=torch.float32, device=dev)
    mult = gaussian if (use_gaussian and num_tiles > 1) else None          # neural_network.py:384-386

    vol = data.contiguous() if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data)).to(dev)   # [C, X, Y, Z]
    V = int(np.prod(shp[1:]))
    # The aggregates (25 GB for 47 classes at 512^3) are kept on the network between calls: allocating and freeing them per volume
    # costs a device malloc of that size each time (0.3-1 s, varying from box to box — it dominated the un-mirrored timings).
    # With return_device_tensors=True the returned probabilities alias this cache until the next call.
    key = (num_classes, tuple(shp[1:]), patch_size, str(dev))
    cache = getattr(net, '_sliding_window_cache', None)
    if cache is None or cache[0] != key:
        net._sliding_window_cache = None
        cache = (key, torch.empty((num_classes,) + tuple(shp[1:]), dtype=torch.float32, device=dev),
                 torch.empty(tuple(shp[1:]), dtype=torch.float32, device=dev),
                 torch.empty((num_classes,) + patch_size, dtype=torch.float32, device=dev))
        net._sliding_window_cache = cache
    _, agg, nb, acc = cache
    agg.zero_(); nb.zero_()
    if do_mirroring:
        combos = [(), (2,), (1,), (2, 1), (0,), (2, 0), (1, 0), (2, 1, 0)]          # neural_network.py:531-586 order
        combos = [c for c in combos if all(a in mirror_axes for a in c)]
        num_results = 2 ** len(mirror_axes)
    else:
        combos, num_results = [()], 1
    eng = net.engine()
    tiles = [(xs, ys, zs) for xs in steps[0] for ys in steps[1] for zs in steps[2]]
    if tile_shard is not None:
        rank, world = tile_shard
        per = (len(tiles) + world - 1) // world
        tiles = tiles[rank * per:(rank + 1) * per]           # contiguous run keeps the per-voxel order of the reference
    was_training = net.training
    with torch.no_grad():
        # all mirrored versions of a tile — and several consecutive tiles — go through the network as ONE batch (per-sample
        # results do not depend on the batch, and the aggregate is still updated tile by tile in the reference's x -> y -> z
        # order with the reference's mirror order inside a tile): up to 8x larger grids on the low-resolution stages
        group = max(1, 8 // len(combos))
        fuse_head = num_classes <= 64 and os.environ.get('MT_INFER_FUSED_HEAD', '1') != '0'
        for g0 in range(0, len(tiles), group):
            chunk = tiles[g0:g0 + group]
            inp = []
            for (xs, ys, zs) in chunk:
                tile = vol[None, :, xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]]
                inp += [torch.flip(tile, tuple(a + 2 for a in c)) if len(c) else tile for c in combos]
            batch = torch.cat(inp, 0).contiguous()
            if fuse_head:
                # head + nonlinearity + un-flip + accumulation in one kernel per sample: the logits never reach HBM
                hp = eng.forward_to_final_head(batch)
            else:
                logits = eng.forward(batch, need_grad=False, all_heads=False)[0]                      # [B, D, H, W, C]
            for t, (xs, ys, zs) in enumerate(chunk):
                if fuse_head and len(combos) > 1:
                    # every mirror combination of the tile, the Gaussian and the overlap-add in one launch (the sum over the
                    # combinations stays in registers).  Without mirroring the two-kernel form below is faster (47.4 vs 43.0
                    # volumes/min in bf16): its streaming overlap-add beats 128-byte read-modify-writes from the MFMA epilogue.
                    ops.head_mirror_accumulate(hp, t * len(combos), [(0 in c, 1 in c, 2 in c) for c in combos], nonlin,
                                               1.0 / num_results, mult, agg, nb, shp[1:], (xs, ys, zs))
                    continue
                for i, c in enumerate(combos):
                    k = t * len(combos) + i
                    if fuse_head:
                        ops.head_flip_accumulate(hp, k, (0 in c, 1 in c, 2 in c), nonlin, 1.0 / num_results, acc, i == 0)
                    else:
                        ops.flip_accumulate(Act(logits[k:k + 1]), (0 in c, 1 in c, 2 in c), nonlin, 1.0 / num_results, acc, i == 0)
                ops.tile_accumulate(acc, mult, num_classes, patch_size, agg, nb, shp[1:], (xs, ys, zs))
    if tile_shard is not None and tile_shard[1] > 1:
        import torch.distributed as dist
        dist.all_reduce(agg)
        dist.all_reduce(nb)
    seg = torch.empty(tuple(shp[1:]), dtype=torch.int32, device=dev)
    if regions_class_order is not None:
        order = torch.tensor([int(c) for c in regions_class_order], dtype=torch.int32, device=dev)
        ops.normalize_threshold(agg, nb, num_classes, V, order, True, seg)
    else:
        ops.normalize_threshold(agg, nb, num_classes, V, None, False, seg)
    sl = (slice(None),) + tuple(slicer[1:])
    probs = agg[sl]
    seg = seg[tuple(slicer[1:])]
    if was_training:
        net.train()
    if return_device_tensors:
        return seg, probs
    seg_np = seg.cpu().numpy()
    seg_np = seg_np.astype(np.float32) if regions_class_order is not None else seg_np.astype(np.int64)
    return seg_np, probs.cpu().numpy()
