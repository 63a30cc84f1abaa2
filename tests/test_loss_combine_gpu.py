"""mt_loss_combine (value + gradient of the loss combination in one launch) against the autograd spelling of the same arithmetic.

The autograd forms (`MultiTalentLoss.forward`, `DC_and_CE_DS_loss.forward`) are the ones pinned to the imported reference's
`compute_loss` (tests/golden/losses.npz, loss_ddp_batchdice: test_kernels_gpu.py / test_golden_gpu.py); the training step's
`fused_step` must give the same loss values and the same dLoss/dlogits.  Reference: MultiTalent_Trainer_DDP.py:544-623,
dice_loss.py:150-183, deep_supervision.py:37-42, nnUNetTrainerV2_DDP.py:249-282."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(8, 24, 24), (4, 12, 12), (2, 6, 6)]


def _levels(dev, B, C, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn((B,) + s + (C,), generator=g) * 2.0).to(dev) for s in SHAPES]


def _autograd(loss_fn, outs, args):
    leaves = [o.permute(0, 4, 1, 2, 3).requires_grad_(True) for o in outs]
    res = loss_fn(leaves, *args)
    (res[0] if isinstance(res, tuple) else res).backward()
    return res, [None if l.grad is None else l.grad.permute(0, 2, 3, 4, 1).contiguous() for l in leaves]


def _same(res_f, dl_f, res_a, dl_a):
    fa = [float(r) for r in (res_a if isinstance(res_a, tuple) else (res_a,))]
    ff = [float(r) for r in (res_f if isinstance(res_f, tuple) else (res_f,))]
    assert np.allclose(ff, fa, rtol=2e-6, atol=2e-6), (ff, fa)
    for a, b in zip(dl_f, dl_a):
        assert (a is None) == (b is None)
        if a is not None:
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-12, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("batch_dice", [True, False])
@pytest.mark.parametrize("B", [1, 2, 4])
def test_multitalent_fused_step_matches_autograd_form(dev, batch_dice, B):
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions
    names = list(MultiTalent_regions.keys())
    rng = np.random.RandomState(B)
    valid = [list(rng.choice(names, size=rng.randint(1, 9), replace=False)) for _ in range(B)]
    loss_fn = MultiTalentLoss([0.5, 0.3, 0.2], batch_dice=batch_dice)
    outs = _levels(dev, B, len(names), 3)
    target = [torch.from_numpy(rng.randint(0, 30, size=(B, 1) + s).astype(np.float32)).to(dev) for s in SHAPES]
    res_a, dl_a = _autograd(loss_fn, outs, (target, valid))
    res_f, dl_f = loss_fn.fused_step(outs, target, valid)
    _same(res_f, dl_f, res_a, dl_a)


@pytest.mark.parametrize("batch_dice,ddp,do_bg", [(False, False, False), (True, False, False), (False, True, False), (True, True, False),
                                                  (False, False, True), (True, False, True)])
@pytest.mark.parametrize("C,weights", [(2, [0.6, 0.4, 0.0]), (4, [0.5, 0.3, 0.2])])
def test_softmax_dice_ce_fused_step_matches_autograd_form(dev, batch_dice, ddp, do_bg, C, weights):
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    loss_fn = DC_and_CE_DS_loss(weights, batch_dice=batch_dice, do_bg=do_bg, ddp=ddp)
    B = 2
    rng = np.random.RandomState(C)
    outs = _levels(dev, B, C, 5)
    target = [torch.from_numpy(rng.randint(0, C, size=(B, 1) + s).astype(np.float32)).to(dev) for s in SHAPES]
    res_a, dl_a = _autograd(loss_fn, outs, (target,))
    fused = loss_fn.fused_step(outs, target)
    assert fused is not None                     # single process: every variant is covered
    _same(fused[0], fused[1], res_a, dl_a)


def test_training_step_takes_the_fused_loss(dev, monkeypatch):
    """FusedTrainStep calls fused_step (no autograd graph) and lands on the same parameters as with the autograd form."""
    from torch import nn
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    pools, kernels = [[2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 3

    def build():
        torch.manual_seed(0)
        net = Generic_UNet(1, 8, 3, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, None, pools, kernels, False, True, True).to(dev)
        net.train()
        return net

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 16, 32, 32, generator=g).to(dev)
    tg = [torch.randint(0, 3, (2, 1, 16 >> i, 32 >> i, 32 >> i), generator=g).float().to(dev) for i in range(2)]
    results = {}
    for fused in (True, False):
        net = build()
        loss_fn = DC_and_CE_DS_loss([0.7, 0.3], batch_dice=False)
        calls = []
        orig = loss_fn.fused_step
        loss_fn.fused_step = lambda *a, _o=orig, _c=calls: (_c.append(1), _o(*a))[1]
        step = FusedTrainStep(net, loss_fn, lr=1e-2)
        step.fused_loss = fused
        losses = [float(step(x, tg)) for _ in range(2)]
        assert len(calls) == (2 if fused else 0)
        results[fused] = (losses, net.engine().flat.detach().clone())
    assert np.allclose(results[True][0], results[False][0], rtol=1e-6, atol=1e-6), (results[True][0], results[False][0])
    d = float((results[True][1] - results[False][1]).abs().max())
    assert d <= 1e-6, d


def test_task100_plan_batch_4_full_resolution_level_vs_oracle(dev):
    """BASELINE configs[2] at its PLAN batch: B = 4 samples x 47 region channels at one 48x192x192 level, four
    different datasets (13, 2, 1 and 8 valid regions: 4 x 47 columns of statistics, most of them untouched) — loss, BCE and Dice terms and dLoss/dlogits of `fused_step` against the oracle's restatement of
    compute_loss (MultiTalent_Trainer_DDP.py:544-623) with autograd on the host."""
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                                        MultiTalent_valid_regions)
    from multitalent_amd.synthetic import synthetic_targets
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    from oracle import reference_ops as R
    B, patch, C = 4, (48, 192, 192), 47
    names = ['Task017_AbdominalOrganSegmentation', 'Task003_Liver', 'Task009_Spleen', 'Task046_AbdOrgSegm2']
    valid = [MultiTalent_valid_regions[n] for n in names]
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    tg = synthetic_targets(B, patch, [[1, 1, 1]], label_sets, 4242, dev)
    g = torch.Generator().manual_seed(77)
    logits = (torch.randn((B,) + patch + (C,), generator=g) * 1.5)           # NDHWC, what the head kernels write
    loss_fn = MultiTalentLoss([1.0], batch_dice=True)
    res, dl = loss_fn.fused_step([logits.to(dev)], tg, valid)
    torch.cuda.synchronize()
    leaf = logits.permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    rl = R.multitalent_loss([leaf], [t.cpu() for t in tg], valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, [1.0])
    rl[0].backward()
    got = [float(r) for r in res]
    want = [float(r) for r in rl]
    print("B=4 Task100 level: loss/ce/dice HIP %s oracle %s" % (got, want))
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4), (got, want)
    ref = leaf.grad.permute(0, 2, 3, 4, 1)
    d = (dl[0].cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print("dlogits: max |d| %.3e of max %.3e" % (d, scale))
    assert d <= 1e-4 * scale, (d, scale)
    # channels of regions that are not valid for a sample carry exactly zero gradient (the reference never touches them)
    for b in range(B):
        idx = sorted(MultiTalent_region_output_idx_mapping[r] for r in valid[b])
        rest = [c for c in range(C) if c not in idx]
        assert float(dl[0][b][..., rest].abs().max()) == 0.0
