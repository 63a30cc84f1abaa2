"""Whole-network parity AT THE BENCHMARK'S OWN SIZE (one 48x192x192 patch per sample) against the CPU oracle
(oracle/reference_ops.py — pinned to the real reference by tests/test_oracle_golden*.py): logits of every deep-supervision level,
the loss, and the gradient of every parameter, for

  (a) BASELINE configs[1]: Task009 Generic_UNet nc = 2, softmax Dice + CE                    (fp32: logits/loss 1e-3; gradients: see compare())
  (b) BASELINE configs[2]: Task100 Generic_UNet nc = 47, MultiTalent BCE + Dice, batch Dice   (same tolerances)
  (c) BASELINE configs[3]: Task100 residual-encoder FabiansUNet, fp32 AND bf16 mixed precision (bf16: the bounds of
      tests/test_mixed_precision_gpu.py — 8 mantissa bits — against the exact fp32 oracle)

and asserts, through mt_conv3d_*_kernel_name, that the kernels the benchmark spends its time in (Winograd forward / backward-data /
backward-weight, strided stage kernels, stem kernels, tap-split low-resolution kernel) are the ones that produced these numbers.
The torch-CPU oracle runs in fp32 (a few seconds per network on the GPU box's host) and in fp64 (the exact gradient, ~40 s)."""
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


def oracle_two_precisions(run):
    """run(dtype) -> (sd with .grad, outputs, loss tuple): the oracle in fp32 (the reference's arithmetic) and in fp64 (the exact
    answer both fp32 implementations approximate)."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))       # oneDNN conv3d is fastest at 32 threads on the GPU box's host
    return run(torch.float32), run(torch.float64)


ENTRY_BOUND = 0.03       # largest entry error of a gradient tensor, as a fraction of the tensor's largest exact entry


def compare(tag, logits, loss, grads, o32, o64, logit_tol=1e-3, loss_tol=1e-3):
    """Logits and loss: within 1e-3 of the fp32 oracle (north_star's tolerance; observed ~3e-5).
    Gradients: at this size the fp32 oracle ITSELF is 1e-3 .. 8e-3 (of a tensor's largest entry) away from the exact gradient:
    a LeakyReLU decision of a voxel within rounding of zero flips and changes that voxel's gradient hundredfold — at the 3x12x12
    stages one flipped voxel moves a weight-gradient entry by ~5 % of its typical size — and InstanceNorm backward subtracts means
    of 1.8 M-voxel sums.  So the truth is the fp64 oracle, and the HIP path is held to: per parameter tensor relative L2 error
    < 1e-2 (tensors whose exact gradient is not numerically zero), largest entry error < 0.03 max|g_64| + 1e-4 G (G = largest
    gradient entry of the network; the second term is the noise floor of gradients that are mathematically ZERO — the bias of every
    conv that feeds an InstanceNorm: 1e-19 in fp64, 1e-10 noise in both fp32 paths), globally relative L2 < max(2e-3, 1.6x the fp32
    torch oracle's own error) and 1 - cos < max(1e-5, 4x the oracle's).  (Round 5: tightened from 0.1 max|g| and max(5e-3, 2x); measured
    1.51e-3 against the oracle's 0.96e-3 on Task009, 0.87e-3 / 0.81e-3 on Task100, 5.0e-3 / 3.3e-3 on the residual encoder.)
    The HIP path normalises with ONE fma per element, t = y * (gamma rstd) + (beta - mu gamma rstd) (DESIGN.md §2 "lazy activations");
    rounding that per-channel offset to fp32 is a coherent 1-ulp perturbation: the same formula emulated inside the fp32 torch
    oracle raises ITS relative L2 gradient error from 0.9e-3 to 2.4e-3 (measured, Task009 network) — the HIP path measures 1.5e-3 on this test's batch (2.1e-3 on the batch of tools/diag_fullsize_grads.py)."""
    sd32, out32, l32 = o32
    sd64, out64, l64 = o64
    for i, (a, b) in enumerate(zip(logits, out32)):
        d = float((a - b.detach()).abs().max())
        assert d < logit_tol, "%s: logits of level %d differ by %.3e" % (tag, i, d)
    for a, b in zip(loss, l32):
        assert abs(a - float(b)) < loss_tol * max(1.0, abs(float(b))), (tag, loss, [float(r) for r in l32])
    G = max(float(v.grad.abs().max()) for v in sd64.values() if v.grad is not None)
    rows, ga, gt, gc = [], [], [], []
    for n, g in grads.items():
        t = sd64[n].grad
        if t is None:
            assert float(g.abs().max()) == 0.0, n           # e.g. the head of a zero-weight deep-supervision level
            continue
        c = sd32[n].grad.double()
        err, cerr, mx = float((g.double() - t).abs().max()), float((c - t).abs().max()), float(t.abs().max())
        # relative L2 only for tensors with a gradient worth the name (rms >= 1e-3 G): the norm biases of the 3x6x6 stage have
        # max|g| = 3e-4 G, and there 2e-5 G of backpropagated rounding noise is 6 % "relative" (r3: all five resenc outputs weighted)
        l2t = float((g.double() - t).norm() / t.norm()) if float(t.norm()) > 1e-3 * G * t.numel() ** 0.5 else 0.0
        rows.append((max(err / (ENTRY_BOUND * mx + 1e-4 * G), l2t / 1e-2), err, cerr, mx, n))
        ga.append(g.double().reshape(-1)); gt.append(t.reshape(-1)); gc.append(c.reshape(-1))
    rows.sort(reverse=True)
    ga, gt, gc = torch.cat(ga), torch.cat(gt), torch.cat(gc)
    l2, l2c = float((ga - gt).norm() / gt.norm()), float((gc - gt).norm() / gt.norm())
    cos = float((ga * gt).sum() / (ga.norm() * gt.norm()))
    print("%s: gradient vs fp64 oracle: HIP rel. L2 %.2e (torch-CPU fp32: %.2e), cos %.7f, G %.2e" % (tag, l2, l2c, cos, G))
    for r in rows[:6]:
        print("   %.2f of bound: max err %.2e (cpu32 %.2e), max|g| %.2e  %s" % r)
    assert rows[0][0] < 1.0, "%s: gradient of %s exceeds its bound by a factor %.2f" % (tag, rows[0][4], rows[0][0])
    # global: within 2e-3 / cos 0.99999 — or, where the fp32 torch oracle itself is that far from the exact gradient (resenc with all
    # five levels weighted: torch-CPU 3.3e-3, HIP 5.0e-3), within 1.6x of what the reference's own arithmetic achieves
    cosc = float((gc * gt).sum() / (gc.norm() * gt.norm()))
    assert l2 < max(2e-3, 1.6 * l2c) and (1.0 - cos) < max(1e-5, 4.0 * (1.0 - cosc)), (tag, l2, l2c, cos, cosc)
    return rows[0]


def mixed_vs_reference_autocast(tag, net_name, logits, loss, grads, o32, o64):
    """The mixed-precision mode against the REFERENCE'S OWN fp16=True arithmetic.  tests/golden/autocast_fullsize_<net>.json
    holds how far the imported reference network + loss, run on the CPU under torch.autocast(float16) with the GradScaler's loss scale
    (tools/oracle_gen/make_golden_autocast.py: nnUNetTrainerV2.py:249-262, MultiTalent_Trainer_DDP.py:340-352), lands from its own fp32
    logits / loss and from the exact (fp64) gradient on THESE inputs (same seeds, same initial weights).  The reference publishes no
    tolerance for its AMP path, so that deviation is the yardstick: the HIP mixed mode (fp16 activations / forward products, bf16
    gradients / backward products) must be within 1.5x of it on every count — logits per level (relative L2 and largest error over
    largest logit), loss, gradient relative L2 and 1 - cos against fp64, and the worst large convolution weight's 1 - cos."""
    import json
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'autocast_fullsize_%s.json' % net_name)))['fp16']
    sd64, out32, l32 = o64[0], o32[1], o32[2]
    rows = []
    for i, (a, b) in enumerate(zip(logits, out32)):
        b = b.detach()
        l2, mx = float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())
        rows.append((l2, ref['logits_rel_l2_vs_fp32'][i], mx, ref['logits_max_err_over_max_vs_fp32'][i]))
    ga = torch.cat([grads[n].reshape(-1) for n in grads]).double()
    gr = torch.cat([(sd64[n].grad if sd64[n].grad is not None else torch.zeros_like(sd64[n])).reshape(-1) for n in grads]).double()
    gl2, cos = float((ga - gr).norm() / gr.norm()), float((ga * gr).sum() / (ga.norm() * gr.norm()))
    big = [n for n in grads if n.endswith('.weight') and grads[n].dim() == 5 and grads[n].numel() > 50000]
    worst = min((float((grads[n].double().reshape(-1) * sd64[n].grad.reshape(-1)).sum() / (grads[n].double().norm() * sd64[n].grad.norm() + 1e-30)), n) for n in big)
    print("%s mixed vs exact | reference fp16 autocast vs exact: logits rel. L2 %s | %s; gradient rel. L2 %.4f | %.4f, cos %.5f | %.5f, worst large conv weight cos %.4f (%s) | %.4f"
          % (tag, ['%.2e' % r[0] for r in rows], ['%.2e' % r[1] for r in rows], gl2, ref['grad_rel_l2_vs_fp64'], cos, ref['grad_cos_vs_fp64'],
             worst[0], worst[1], ref['worst_large_conv_weight_cos_vs_fp64']))
    for i, (l2, rl2, mx, rmx) in enumerate(rows):
        assert l2 <= 1.5 * rl2 and mx <= 1.5 * rmx, "%s: logits of level %d: rel. L2 %.3e (reference autocast %.3e), max %.3e (%.3e)" % (tag, i, l2, rl2, mx, rmx)
    for a, b, d in zip(loss, l32, ref['loss_abs_err_vs_fp32']):
        assert abs(a - float(b)) <= 1.5 * d + 2e-5 * max(1.0, abs(float(b))), (tag, loss, [float(r) for r in l32], ref['loss_abs_err_vs_fp32'])
    assert gl2 <= 1.5 * ref['grad_rel_l2_vs_fp64'] and (1.0 - cos) <= 1.5 * (1.0 - ref['grad_cos_vs_fp64']), (tag, gl2, cos)
    assert (1.0 - worst[0]) <= 1.5 * (1.0 - ref['worst_large_conv_weight_cos_vs_fp64']), (tag, worst)


def test_task009_fullsize_forward_loss_backward_vs_oracle(dev):
    import bench
    from oracle import reference_ops as R
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    torch.manual_seed(1234)
    net = bench.build_network('task009')
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.train()
    x = synthetic_ct(1, PATCH, 77, dev)
    tg = synthetic_targets(1, PATCH, ds_scales(bench.POOLS), [[1]], 77, dev)
    w = R.ds_loss_weights(len(bench.POOLS))
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, DC_and_CE_DS_loss(w, batch_dice=False), x, (tg,))

    def run(dt):
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
        out = R.generic_unet_forward(sd, x.cpu().to(dt), bench.POOLS, bench.KERNELS)
        l = R.multiple_output_loss(out, [t.cpu() for t in tg], w)
        l.backward()
        return sd, [o.detach().float() for o in out], [l.detach()]
    o32, o64 = oracle_two_precisions(run)
    compare('task009', logits, loss, grads, o32, o64)
    names32 = names
    # BASELINE's mixed mode on the same network and batch, against the reference's own fp16 autocast deviation
    from multitalent_amd import ops
    try:
        net.engine().set_precision('bf16')
        with recorded_kernels() as names16:
            lb, lossb, gb = hip_forward_backward(net, DC_and_CE_DS_loss(w, batch_dice=False), x, (tg,))
        mixed_vs_reference_autocast('task009', 'task009', lb, lossb, gb, o32, o64)
        assert any(n.startswith('conv_bwdw_tr16_kernel<3') for n in names16['bwdw']), names16['bwdw']
        assert sum(n.startswith(('conv_bf16', 'conv_x16')) for n in names16['fwd']) >= 10, names16['fwd']
    finally:
        net.engine().set_precision('fp32')
        ops.set_mma(0)
    names = names32
    # the kernels bench.py times are the ones checked here
    assert any(n.startswith('conv_wino') for n in names['fwd']), names['fwd']
    assert sum(n.startswith('conv_wino') for n in names['fwd']) >= 10          # forward + backward-data of the three top stages
    assert any(n.startswith('conv_fast_strided') for n in names['fwd'])
    assert any(n.startswith('conv_stem_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_tapsplit_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_gather_kernel') for n in names['fwd'])
    assert any(n.startswith('conv_bwdw_wino_kernel') for n in names['bwdw']), names['bwdw']
    assert any(n.startswith('conv_bwdw_stem_kernel') for n in names['bwdw'])
    assert any(n.startswith('conv_bwdw_fast_kernel<3, 3, 3, 2, 2, 2>') for n in names['bwdw'])
    # (the bottleneck's and the 240 <- 320 layer's backward-data: under-filled grids, K split over the waves)
    assert any(n.startswith('conv_bwdd_strided_kernel<2') for n in names['bwdd']) and any(n.startswith('conv_bwdd_strided_ks_kernel<1') for n in names['bwdd'])
    assert any(n.startswith('conv_bwdd_strided_ks_kernel<2') for n in names['bwdd'])


def test_task100_fullsize_multitalent_loss_vs_oracle(dev):
    _task100(dev, PATCH, 2, 'task100')


def test_task100_native_patch_96x192x192_vs_oracle(dev):
    """the patch a real Task100 run uses (MultiTalent_plans/MultiTalent_bs4_plans_3D.pkl stage 1, tests/golden/plans_stage1.json): 96x192x192,
    one sample (the same number of voxels as the B = 2 benchmark batch) — logits, loss and gradients against the fp32 / fp64 oracle"""
    _task100(dev, (96, 192, 192), 1, 'task100 96x192x192')


def _task100(dev, PATCH, B, tag):
    import bench
    from oracle import reference_ops as R
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                                        MultiTalent_valid_regions)
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    torch.manual_seed(4321)
    net = bench.build_network('task100')
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.train()
    valid = [MultiTalent_valid_regions['Task046_AbdOrgSegm2'], MultiTalent_valid_regions['Task003_Liver']][:B]
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    x = synthetic_ct(B, PATCH, 78, dev)
    tg = synthetic_targets(B, PATCH, ds_scales(bench.POOLS), label_sets, 78, dev)
    w = R.ds_loss_weights(len(bench.POOLS))
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, MultiTalentLoss(w, batch_dice=True), x, (tg, valid))

    def run(dt):
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
        out = R.generic_unet_forward(sd, x.cpu().to(dt), bench.POOLS, bench.KERNELS)
        rl = R.multitalent_loss(list(out), [t.cpu() for t in tg], valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
        rl[0].backward()
        return sd, [o.detach().float() for o in out], [r.detach() for r in rl]
    compare(tag, logits, loss, grads, *oracle_two_precisions(run))
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
    w = R.ds_loss_weights(len(bench.RESENC_POOLS))      # all five outputs weighted (MultiTalent_meets_resenc.py:157-170)
    with recorded_kernels() as names:
        logits, loss, grads = hip_forward_backward(net, MultiTalentLoss(w, batch_dice=True), x, (tg, valid))
    return sd0, x, tg, valid, w, logits, loss, grads, names


def _resenc_oracle(sd0, x, tg, valid, w):
    import bench
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_region_output_idx_mapping, MultiTalent_regions
    from oracle import reference_ops as R

    def run(dt):
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
        out = R.fabians_unet_forward(sd, x.cpu().to(dt), bench.RESENC_POOLS, bench.RESENC_KERNELS, bench.RESENC_BLOCKS)
        rl = R.multitalent_loss(list(out), [t.cpu() for t in tg], valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
        rl[0].backward()
        return sd, [o.detach().float() for o in out], [r.detach() for r in rl]
    return oracle_two_precisions(run)


def test_resenc_fullsize_fp32_and_bf16_vs_oracle(dev):
    """configs[3]: the residual-encoder network at full size.  fp32 within the fp32 tolerances; the mixed-precision mode configs[3] names
    within 1.5x of the deviation the REFERENCE'S OWN fp16 autocast shows on the same inputs (mixed_vs_reference_autocast: reference logits
    0.2 .. 1.6 % relative L2, gradient relative L2 0.227 / cos 0.974 against fp64, worst large conv weight cos 0.957;
    profiles/r05_reference_autocast_fullsize_resenc.json)."""
    from multitalent_amd import ops
    try:
        _resenc_fp32_and_bf16(dev)
    finally:
        ops.set_mma(0)               # the engine leaves its mode in the process-wide default of ops.fill_conv


def _resenc_fp32_and_bf16(dev):
    sd0, x, tg, valid, w, logits, loss, grads, names = _resenc(dev, 'fp32')
    o32, o64 = _resenc_oracle(sd0, x, tg, valid, w)
    compare('resenc fp32', logits, loss, grads, o32, o64)
    assert any(n.startswith('conv_wino') for n in names['fwd']), names['fwd']
    sd, out, rl = o64[0], o32[1], o32[2]            # bf16 below is judged against the exact gradient and the fp32 logits / loss
    del grads, logits
    torch.cuda.empty_cache()
    _, _, _, _, _, lb, lossb, gb, nb = _resenc(dev, 'bf16')
    assert sum(n.startswith(('conv_bf16', 'conv_x16')) for n in nb['fwd']) >= 20, nb['fwd']
    assert any(n.startswith('conv_bwdw_tr16_kernel<3') for n in nb['bwdw']) and any(n.startswith('conv_bwdw_tr16_kernel<1') for n in nb['bwdw']), nb['bwdw']
    assert any(n.startswith('bwdw_gemm_kernel') for n in nb['bwdw'])                      # the low-resolution stages
    mixed_vs_reference_autocast('resenc', 'resenc', lb, lossb, gb, o32, o64)
