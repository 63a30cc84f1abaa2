"""Whole-network parity AT THE BENCHMARK'S OWN SIZE (one 48x192x192 patch per sample) against the CPU oracle
(oracle/reference_ops.py — pinned to the real reference by tests/test_oracle_golden*.py): logits of every deep-supervision level,
the loss, and the gradient of every parameter, for

  (a) BASELINE configs[1]: Task009 Generic_UNet nc = 2, softmax Dice + CE                    (fp32: logits/loss 1e-3, grads 2e-3 of max)
  (b) BASELINE configs[2]: Task100 Generic_UNet nc = 47, MultiTalent BCE + Dice, batch Dice   (same tolerances)
  (c) BASELINE configs[3]: Task100 residual-encoder FabiansUNet, fp32 AND bf16 mixed precision (bf16: the bounds of
      tests/test_mixed_precision_gpu.py — 8 mantissa bits — against the exact fp32 oracle)

and asserts, through mt_conv3d_*_kernel_name, that the kernels the benchmark spends its time in (Winograd forward / backward-data /
backward-weight, strided stage kernels, stem kernels, tap-split low-resolution kernel) are the ones that produced these numbers.
The torch-CPU oracle needs a few seconds per network on the GPU box's host cores."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PATCH = (48, 192, 192)


@contextlib.contextmanager
def recorded_kernels():
    """names of every convolution kernel family launched inside the block (forward-type, backward-weight, strided backward-data)."""
    from multitalent_amd import ops
    names = {'fwd': [], 'bwdw': [], 'bwdd': []}
    orig = (ops.conv3d_fwd, ops.conv3d_bwd_weight, ops.conv3d_bwd_data_strided)

    def fwd(p):
        names['fwd'].append(ops.conv_kernel_name(p))
        orig[0](p)

    def bwdw(p, y, *a):
        names['bwdw'].append(ops.conv_bwd_weight_kernel_name(p, y))
        orig[1](p, y, *a)

    def bwdd(p):
        names['bwdd'].append(ops.conv_bwd_data_strided_kernel_name(p))
        orig[2](p)

    ops.conv3d_fwd, ops.conv3d_bwd_weight, ops.conv3d_bwd_data_strided = fwd, bwdw, bwdd
    try:
        yield names
    finally:
        ops.conv3d_fwd, ops.conv3d_bwd_weight, ops.conv3d_bwd_data_strided = orig


def hip_forward_backward(net, loss_fn, x, largs):
    """forward + loss + backward on the engine WITHOUT the optimizer step; returns (logits [NCDHW, cpu], loss tuple, {name: grad})."""
    from multitalent_amd.training.hot_loop import FusedTrainStep
    step = FusedTrainStep(net, loss_fn, lr=0.0)
    eng = net.engine()
    leaves, res = step.forward_loss(x, largs)
    loss = res[0] if isinstance(res, (tuple, list)) else res
    loss.backward()
    dl = [None if l.grad is None else l.grad.permute(0, 2, 3, 4, 1).contiguous() for l in leaves]
    eng.backward(dl)
    torch.cuda.synchronize()
    logits = [l.detach().float().cpu() for l in leaves]
    grads = {n: eng.grad_of(p).detach().cpu().clone() for n, p in net.named_parameters()}
    vals = [float(r.detach()) for r in res] if isinstance(res, (tuple, list)) else [float(res.detach())]
    return logits, vals, grads


def compare(tag, logits, ref_logits, loss, ref_loss, grads, ref_sd, logit_tol, loss_tol, grad_tol):
    for i, (a, b) in enumerate(zip(logits, ref_logits)):
        d = float((a - b.detach()).abs().max())
        assert d < logit_tol, "%s: logits of level %d differ by %.3e" % (tag, i, d)
    for a, b in zip(loss, ref_loss):
        assert abs(a - float(b)) < loss_tol * max(1.0, abs(float(b))), (tag, loss, [float(r) for r in ref_loss])
    worst = ('', 0.0)
    for n, g in grads.items():
        ref = ref_sd[n].grad
        if ref is None:
            assert float(g.abs().max()) == 0.0, n           # e.g. the head of a zero-weight deep-supervision level
            continue
        rel = float((g - ref).abs().max()) / max(float(ref.abs().max()), 1e-3)
        if rel > worst[1]:
            worst = (n, rel)
    assert worst[1] < grad_tol, "%s: gradient of %s off by %.3e of its largest entry" % (tag, worst[0], worst[1])
    return worst


def test_task009_fullsize_forward_loss_backward_vs_oracle(dev):
    import bench
    from oracle import reference_ops as R
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    torch.manual_seed(1234)
    net = bench.build_network('task009')
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net.train()
    x = synthetic_ct(1, PATCH, 77, dev)
    tg = synthetic_targets(1, PATCH, ds_scales(bench.POOLS), [[1]], 77, dev)
    w = R.ds_loss_weights(len(bench.POOLS))
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, DC_and_CE_DS_loss(w, batch_dice=False), x, (tg,))
    out = R.generic_unet_forward(sd, x.cpu(), bench.POOLS, bench.KERNELS)
    ref_loss = R.multiple_output_loss(out, [t.cpu() for t in tg], w)
    ref_loss.backward()
    compare('task009', logits, out, loss, [ref_loss], grads, sd, 1e-3, 1e-3, 2e-3)
    # the kernels bench.py times are the ones checked here
    assert any(n.startswith('conv_wino') for n in names['fwd']), names['fwd']
    assert sum(n.startswith('conv_wino') for n in names['fwd']) >= 10          # forward + backward-data of the three top stages
    assert any(n.startswith('conv_fast_strided_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_stem_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_tapsplit_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_gather_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_bwdw_wino_kernel') for n in names['bwdw']), names['bwdw']
    assert any(n.startswith('conv_bwdw_stem_kernel') for n in names['bwdw'])
    assert any(n.startswith('conv_bwdw_fast_kernel<3, 3, 3, 2, 2, 2>') for n in names['bwdw'])
    assert any(n.startswith('conv_bwdd_strided_kernel<2') for n in names['bwdd']) and any(n.startswith('conv_bwdd_strided_kernel<1') for n in names['bwdd'])


def test_task100_fullsize_multitalent_loss_vs_oracle(dev):
    import bench
    from oracle import reference_ops as R
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                                        MultiTalent_valid_regions)
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    torch.manual_seed(4321)
    net = bench.build_network('task100')
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net.train()
    B = 2
    valid = [MultiTalent_valid_regions['Task046_AbdOrgSegm2'], MultiTalent_valid_regions['Task003_Liver']]
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    x = synthetic_ct(B, PATCH, 78, dev)
    tg = synthetic_targets(B, PATCH, ds_scales(bench.POOLS), label_sets, 78, dev)
    w = R.ds_loss_weights(len(bench.POOLS))
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, MultiTalentLoss(w, batch_dice=True), x, (tg, valid))
    out = R.generic_unet_forward(sd, x.cpu(), bench.POOLS, bench.KERNELS)
    rl = R.multitalent_loss(list(out), [t.cpu() for t in tg], valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
    rl[0].backward()
    compare('task100', logits, out, loss, rl, grads, sd, 1e-3, 1e-3, 2e-3)
    assert any(n.startswith('conv_wino') for n in names['fwd']) and any(n.startswith('conv_bwdw_wino_kernel') for n in names['bwdw'])


def _resenc(dev, precision, B=1):
    import bench
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    from oracle import reference_ops as R
    torch.manual_seed(99)
    net = bench.build_network('resenc')
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.train()
    net.engine().set_precision(precision)
    valid = [MultiTalent_valid_regions['Task064_KiTS_labelsFixed']] * B
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    x = synthetic_ct(B, PATCH, 79, dev)
    tg = synthetic_targets(B, PATCH, ds_scales(bench.RESENC_POOLS, skip_first=True), label_sets, 79, dev)
    w = R.ds_loss_weights(len(bench.RESENC_POOLS) - 1)
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, MultiTalentLoss(w, batch_dice=True), x, (tg, valid))
    return sd0, x, tg, valid, w, logits, loss, grads, names


def _resenc_oracle(sd0, x, tg, valid, w):
    import bench
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_region_output_idx_mapping, MultiTalent_regions
    from oracle import reference_ops as R
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    out = R.fabians_unet_forward(sd, x.cpu(), bench.RESENC_POOLS, bench.RESENC_KERNELS, bench.RESENC_BLOCKS)
    rl = R.multitalent_loss(list(out), [t.cpu() for t in tg], valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
    rl[0].backward()
    return sd, out, rl


def test_resenc_fullsize_fp32_and_bf16_vs_oracle(dev):
    """configs[3]: the residual-encoder network at full size.  fp32 within the fp32 tolerances; bf16 mixed precision — the mode
    configs[3] names — against the SAME exact oracle within what 8 mantissa bits allow (tests/test_mixed_precision_gpu.py):
    logits within 3e-2 of the largest logit, loss within 1e-2 (relative), gradient direction cos > 0.995 overall."""
    sd0, x, tg, valid, w, logits, loss, grads, names = _resenc(dev, 'fp32')
    sd, out, rl = _resenc_oracle(sd0, x, tg, valid, w)
    compare('resenc fp32', logits, out, loss, rl, grads, sd, 1e-3, 1e-3, 2e-3)
    assert any('1' == n.split(',')[-1].strip(' >') for n in names['fwd'] if n.startswith('conv_fast_kernel')) or \
        any(n.startswith('conv_rt_kernel') or n.startswith('conv_fast_kernel') for n in names['fwd'])          # the 1x3x3 first stage
    del grads, logits
    torch.cuda.empty_cache()
    _, _, _, _, _, lb, lossb, gb, nb = _resenc(dev, 'bf16')
    assert sum(n.startswith('conv_bf16_kernel') for n in nb['fwd']) >= 20, nb['fwd']
    assert any(n.startswith('conv_bwdw_wino_bf16_kernel<3>') for n in nb['bwdw']) and any(n.startswith('conv_bwdw_wino_bf16_kernel<1>') for n in nb['bwdw'])
    for i, (a, b) in enumerate(zip(lb, out)):
        b = b.detach()
        assert float((a - b).abs().max()) < 3e-2 * float(b.abs().max()), "bf16 logits level %d" % i
    for a, b in zip(lossb, rl):
        assert abs(a - float(b)) < 1e-2 * max(1.0, abs(float(b))), (lossb, [float(r) for r in rl])
    ga = torch.cat([gb[n].reshape(-1) for n in gb]).double()
    gr = torch.cat([(sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])).reshape(-1) for n in gb]).double()
    cos = float((ga * gr).sum() / (ga.norm() * gr.norm()))
    assert cos > 0.995, cos
    # per-tensor direction for the big convolution weights (every one of them went through a bf16 kernel somewhere)
    for n in gb:
        if n.endswith('.weight') and gb[n].dim() == 5 and gb[n].numel() > 50000:
            a, r = gb[n].double().reshape(-1), sd[n].grad.double().reshape(-1)
            c = float((a * r).sum() / (a.norm() * r.norm() + 1e-30))
            assert c > 0.98, (n, c)
