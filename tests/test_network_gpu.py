"""Whole-network parity on the GPU: the HIP engine (through the C ABI) against the oracle's CPU restatement of
Generic_UNet / FabiansUNet forward, the MultiTalent and softmax Dice+CE losses, and autograd gradients.
Tolerance: logits / loss 1e-3 (north_star: 1e-3 fp32), gradients 2e-3 relative to the largest gradient entry."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

from oracle import reference_ops as R


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def build_plain(nc, base=10, pools=((2, 2, 2), (2, 2, 2), (1, 2, 2)), kernels=None, in_ch=1):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.network_architecture.initialization import InitWeights_He
    pools = [list(p) for p in pools]
    kernels = [[3, 3, 3]] * (len(pools) + 1) if kernels is None else kernels
    torch.manual_seed(0)
    net = Generic_UNet(in_ch, base, nc, len(pools), 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                       nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                       True, False, lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)
    # make norm affine params and biases non-trivial
    g = torch.Generator().manual_seed(1)
    for n, p in net.named_parameters():
        if 'instnorm.weight' in n:
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        elif n.endswith('bias'):
            p.data = 0.2 * torch.randn(p.shape, generator=g)
    return net, pools, kernels


def make_targets(shape, scales, nlabels, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    full = torch.randint(0, nlabels, (B, 1) + tuple(shape), generator=g).float()
    # blocky labels: nearest-neighbour upsample of a coarse grid
    coarse = torch.randint(0, nlabels, (B, 1) + tuple(max(s // 4, 1) for s in shape), generator=g).float()
    full = torch.nn.functional.interpolate(coarse, size=tuple(shape), mode='nearest')
    out = []
    for sc in scales:
        size = tuple(int(round(s * f)) for s, f in zip(shape, sc))
        out.append(torch.nn.functional.interpolate(full, size=size, mode='nearest'))
    return out


def test_plain_unet_softmax_fwd_bwd(dev):
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    nc, B, shape = 3, 2, (8, 32, 32)
    net, pools, kernels = build_plain(nc)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, 1) + shape, generator=g)
    ref_out = R.generic_unet_forward(sd, x, pools, kernels)
    scales = [[1, 1, 1], [.5, .5, .5], [.25, .25, .25]]
    targets = make_targets(shape, scales, nc, B)
    w = R.ds_loss_weights(len(pools))
    ref_loss = R.multiple_output_loss(ref_out, targets, w, batch_dice=False)
    ref_loss.backward()

    net.train()
    out = net(x.to(dev))
    assert isinstance(out, tuple) and len(out) == len(ref_out)
    for o, r in zip(out, ref_out):
        assert tuple(o.shape) == tuple(r.shape)
        assert float((o.detach().cpu() - r.detach()).abs().max()) < 1e-3
    loss_fn = DC_and_CE_DS_loss(w, batch_dice=False)
    loss = loss_fn(out, [t.to(dev) for t in targets])
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-3
    loss.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        if sd[n].grad is None:      # head of the zero-weight deep-supervision level (deep_supervision.py:41)
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        e = relerr(p.grad.cpu(), sd[n].grad)
        scale = float(sd[n].grad.abs().max())
        if scale > 1e-6:
            worst = max(worst, e)
            assert e < 2e-3, (n, e, scale)
        else:   # bias before InstanceNorm: analytically zero gradient
            assert float(p.grad.abs().max()) < 1e-4, n


def test_plain_unet_multitalent_loss(dev):
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_regions, MultiTalent_region_output_idx_mapping,
                                                                        MultiTalent_valid_regions)
    nc, B, shape = 47, 2, (8, 32, 32)
    net, pools, kernels = build_plain(nc, base=8)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    x = torch.randn((B, 1) + shape, generator=g)
    valid = [MultiTalent_valid_regions['Task003_Liver'], MultiTalent_valid_regions['Task017_AbdominalOrganSegmentation']]
    scales = [[1, 1, 1], [.5, .5, .5], [.25, .25, .25]]
    targets = make_targets(shape, scales, 23, B, seed=3)
    w = R.ds_loss_weights(len(pools))
    ref_out = R.generic_unet_forward(sd, x, pools, kernels)
    rl, rce, rdc = R.multitalent_loss(list(ref_out), targets, valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
    rl.backward()
    net.train()
    out = net(x.to(dev))
    for o, r in zip(out, ref_out):
        assert float((o.detach().cpu() - r.detach()).abs().max()) < 1e-3
    l, ce, dc = MultiTalentLoss(w, batch_dice=True)(out, [t.to(dev) for t in targets], valid)
    assert abs(float(l) - float(rl)) < 1e-3 * max(1.0, abs(float(rl)))
    assert abs(float(ce) - float(rce)) < 1e-3 * max(1.0, abs(float(rce)))
    assert abs(float(dc) - float(rdc)) < 1e-3 * max(1.0, abs(float(rdc)))
    l.backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        scale = float(sd[n].grad.abs().max())
        if scale > 1e-6:
            assert relerr(p.grad.cpu(), sd[n].grad) < 2e-3, n


def test_resenc_unet_fwd_bwd(dev):
    from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
    from multitalent_amd.network_architecture.initialization import InitWeights_He
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_regions, MultiTalent_region_output_idx_mapping,
                                                                        MultiTalent_valid_regions)
    pools = [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
    kernels = [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]
    blocks = [1, 2, 2, 2]
    torch.manual_seed(0)
    net = FabiansUNet(1, 8, blocks, 2, pools, kernels, get_default_network_config(3, None, norm_type="in"), 47, [1, 1, 1],
                      True, False, 24, InitWeights_He(1e-2))
    g = torch.Generator().manual_seed(1)
    for n, p in net.named_parameters():
        if 'norm' in n and n.endswith('weight') and p.dim() == 1:
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        elif n.endswith('bias'):
            p.data = 0.2 * torch.randn(p.shape, generator=g)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items() if '.all.' not in k}
    B, shape = 2, (8, 32, 32)
    x = torch.randn((B, 1) + shape, generator=g)
    ref_out = R.fabians_unet_forward(sd, x, pools, kernels, blocks)
    valid = [MultiTalent_valid_regions['Task046_AbdOrgSegm2'], MultiTalent_valid_regions['Task064_KiTS_labelsFixed']]
    scales = [[1, 1, 1], [1, .5, .5], [.5, .25, .25]]
    targets = make_targets(shape, scales, 44, B, seed=4)
    w = [0.5, 0.3, 0.2]
    rl, rce, rdc = R.multitalent_loss(list(ref_out), targets, valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
    rl.backward()
    net.train()
    out = net(x.to(dev))
    assert len(out) == len(ref_out)
    for o, r in zip(out, ref_out):
        assert tuple(o.shape) == tuple(r.shape)
        assert float((o.detach().cpu() - r.detach()).abs().max()) < 1e-3
    l, ce, dc = MultiTalentLoss(w, batch_dice=True)(out, [t.to(dev) for t in targets], valid)
    assert abs(float(l) - float(rl)) < 1e-3 * max(1.0, abs(float(rl)))
    l.backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        scale = float(sd[n].grad.abs().max())
        if scale > 1e-6:
            assert relerr(p.grad.cpu(), sd[n].grad) < 2e-3, n


def test_inference_mode_single_output(dev):
    nc, B, shape = 3, 1, (8, 32, 32)
    net, pools, kernels = build_plain(nc)
    sd = net.state_dict()
    x = torch.randn((B, 1) + shape)
    ref = R.generic_unet_forward(sd, x, pools, kernels, deep_supervision=False)
    net.eval()
    net.do_ds = False
    with torch.no_grad():
        out = net(x.to(dev))
    assert tuple(out.shape) == tuple(ref.shape)
    assert float((out.cpu() - ref).abs().max()) < 1e-3


@pytest.mark.parametrize("arch", ["plain", "resenc"])
def test_weight_gradient_stream_changes_nothing(dev, arch):
    """Engine.weight_stream: the backward-weight launches (and the backward-only weight packings) run on a side HIP stream beside the
    backward-data chain.  Same kernels, same ordered reductions: logits and EVERY gradient must be bit-identical with the stream on
    (default) and off, three iterations in a row (the second and third reuse the buffers while the side stream may still be busy)."""
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    B, shape = 2, (8, 32, 32)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn((B, 1) + shape, generator=g).to(dev) for _ in range(3)]

    def run(streams):
        if arch == "plain":
            nc = 3
            net, pools, _ = build_plain(nc)
            scales = [[1, 1, 1], [.5, .5, .5], [.25, .25, .25]]
        else:
            from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
            from multitalent_amd.network_architecture.initialization import InitWeights_He
            nc = 3
            torch.manual_seed(0)
            net = FabiansUNet(1, 8, [1, 2, 2, 2], 2, [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2]], [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
                              get_default_network_config(3, None, norm_type="in"), nc, [1, 1, 1], True, False, 24, InitWeights_He(1e-2))
            scales = [[1, 1, 1], [1, .5, .5], [.5, .25, .25]]
        net.train()
        net.engine().bwdw_streams = streams
        targets = [t.to(dev) for t in make_targets(shape, scales, nc, B, seed=4)]
        loss = DC_and_CE_DS_loss([0.5, 0.3, 0.2])
        outs = []
        for x in xs:
            for p in net.parameters():
                p.grad = None
            out = net(x)
            l = loss(out, targets)
            l.backward()
            torch.cuda.synchronize()
            outs.append([out[0].detach().clone()] + [p.grad.detach().clone() for p in net.parameters()])
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(p.grad, alpha=-0.05)            # the next iteration packs new weights (late packing on the side stream)
            net.engine().mark_params_dirty()
        return outs

    on, off = run(1), run(0)
    for it, (a, b) in enumerate(zip(on, off)):
        for i, (u, v) in enumerate(zip(a, b)):
            assert torch.equal(u, v), (it, i)


@pytest.mark.parametrize("arch", ["plain", "resenc"])
def test_fused_norm_backward_statistics_through_the_engine(dev, arch):
    """Engine.fuse_norm_bwd (default on only without the weight-gradient stream): the convolution / residual add that produces a norm
    layer's output gradient also emits the first pass of that norm's backward (mt_bwd_stats_t, mt_lrelu_bwd_stats).  Same sums in a
    different order: every gradient within 2e-5 of the unfused engine's, relative to the tensor's largest entry."""
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    from multitalent_amd import ops
    B, shape = 2, (8, 32, 64)
    g = torch.Generator().manual_seed(19)
    x = torch.randn((B, 1) + shape, generator=g).to(dev)
    ops.set_option('conv_wino', 2)                   # the Winograd kernels (the ones that fuse) also at this small size
    try:
        def run(fuse):
            nc = 3
            if arch == "plain":
                net, pools, _ = build_plain(nc, base=16)
                scales = [[1, 1, 1], [.5, .5, .5], [.25, .25, .25]]
            else:
                from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
                from multitalent_amd.network_architecture.initialization import InitWeights_He
                torch.manual_seed(0)
                net = FabiansUNet(1, 16, [1, 2, 2, 2], 2, [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2]], [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
                                  get_default_network_config(3, None, norm_type="in"), nc, [1, 1, 1], True, False, 48, InitWeights_He(1e-2))
                scales = [[1, 1, 1], [1, .5, .5], [.5, .25, .25]]
            net.train()
            eng = net.engine()
            eng.fuse_norm_bwd = fuse
            targets = [t.to(dev) for t in make_targets(shape, scales, nc, B, seed=4)]
            out = net(x)
            DC_and_CE_DS_loss([0.5, 0.3, 0.2])(out, targets).backward()
            torch.cuda.synchronize()
            fused = sum(1 for k in eng._buffers if k.endswith('.bwdpart'))
            return [p.grad.detach().clone() for p in net.parameters()], fused
        on, n_on = run(1)
        off, n_off = run(0)
        assert n_on > 0 and n_off == 0, (n_on, n_off)          # the fused path really ran (its partial buffers exist) / did not
        for i, (a, b) in enumerate(zip(on, off)):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-3), i
    finally:
        ops.set_option('conv_wino', 1)
