"""The mixed-precision mode pinned to the REFERENCE's own `fp16=True` arithmetic.

tests/golden/autocast.npz holds what the imported reference networks + losses produce on CPU under `torch.autocast('cpu', float16)` (the
reference's mode: nnUNetTrainerV2.py:249-262, MultiTalent_Trainer_DDP.py:340-352, with the GradScaler's 65536 loss scale) and under
bfloat16 autocast, next to their fp32 logits / loss and fp64 gradient (tools/oracle_gen/make_golden_autocast.py).  The reference
publishes no tolerance for its AMP path, so its OWN deviation is the yardstick:

  * CPU: the fixture is self-consistent (the deviations in autocast_summary.json are recomputed from the arrays);
  * GPU: the HIP mixed mode (fp16 activations / forward products, bf16 gradients / backward products) deviates from the exact
    answers — fp32 logits and loss, fp64 gradient — by at most 1.5x what the reference's fp16 autocast deviates (plus the fp32 noise
    floor), and agrees with the autocast outputs themselves within the sum of both deviations."""
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _flat(z, prefix, names):
    return np.concatenate([z[prefix + n].astype(np.float64).reshape(-1) for n in names])


def _names(z, tag):
    p = '%s/fp64/grad/' % tag
    return [k[len(p):] for k in z.files if k.startswith(p)]


def _ref_dev(z, tag, variant, nlev=3):
    names = _names(z, tag)
    g64 = _flat(z, '%s/fp64/grad/' % tag, names)
    g = _flat(z, '%s/%s/grad/' % (tag, variant), names)
    lg = [float(np.linalg.norm(z['%s/%s/out%d' % (tag, variant, i)].astype(np.float64) - z['%s/fp32/out%d' % (tag, i)]) /
                np.linalg.norm(z['%s/fp32/out%d' % (tag, i)])) for i in range(nlev)]
    return {'logits': lg, 'loss': np.abs(z['%s/%s/loss' % (tag, variant)] - z['%s/fp32/loss' % tag]),
            'grad_l2': float(np.linalg.norm(g - g64) / np.linalg.norm(g64)),
            'grad_cos': float((g * g64).sum() / (np.linalg.norm(g) * np.linalg.norm(g64)))}


@pytest.mark.parametrize("tag", ["plain", "resenc"])
def test_autocast_fixture_is_self_consistent(tag):
    z = np.load(os.path.join(G, 'autocast.npz'))
    s = json.load(open(os.path.join(G, 'autocast_summary.json')))[tag]
    for v in ('fp16', 'bf16'):
        d = _ref_dev(z, tag, v)
        assert np.allclose(d['logits'], s[v]['logits_rel_l2_vs_fp32'], rtol=1e-5)
        assert abs(d['grad_l2'] - s[v]['grad_rel_l2_vs_fp64']) < 1e-6 and abs(d['grad_cos'] - s[v]['grad_cos_vs_fp64']) < 1e-6
        assert float(z['%s/%s/scale' % (tag, v)]) == 65536.0          # the GradScaler's initial scale survived: no overflow at this size
    # the reference's own ordering: fp16 autocast is the closer one (11 against 8 significand bits)
    assert _ref_dev(z, tag, 'fp16')['grad_l2'] < _ref_dev(z, tag, 'bf16')['grad_l2']


def _hip(tag, dev):
    from multitalent_amd.training.hot_loop import FusedTrainStep
    if tag == 'plain':
        from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
        from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
        z = np.load(os.path.join(G, 'plain_unet.npz'))
        net = Generic_UNet(1, 6, 4, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, None, z['pools'].tolist(), z['kernels'].tolist(), False, True, True)
        loss_fn, extra = DC_and_CE_DS_loss(z['weights'], batch_dice=False), ()
    else:
        from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
        from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
        z = np.load(os.path.join(G, 'resenc_unet.npz'))
        net = FabiansUNet(1, 6, z['blocks'].tolist(), 2, z['pools'].tolist(), z['kernels'].tolist(), get_default_network_config(3, None, norm_type="in"),
                          47, [1, 1, 1], True, False, 16, None)
        loss_fn = MultiTalentLoss(z['weights'], batch_dice=True)
        extra = (json.load(open(os.path.join(G, 'resenc_unet_valid.json')))['valid_regions'],)
    net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')})
    net.train()
    net.engine().set_precision('bf16')
    x = torch.from_numpy(z['x']).to(dev)
    tg = [torch.from_numpy(z['target%d' % i]).to(dev) for i in range(3)]
    step = FusedTrainStep(net, loss_fn, lr=0.0)
    leaves, res = step.forward_loss(x, (tg,) + extra)
    loss = res[0] if isinstance(res, (tuple, list)) else res
    loss.backward()
    net.engine().backward([None if l.grad is None else l.grad.permute(0, 2, 3, 4, 1).contiguous() for l in leaves])
    torch.cuda.synchronize()
    vals = [float(r.detach()) for r in res] if isinstance(res, (tuple, list)) else [float(res.detach())]
    return [l.detach().float().cpu().numpy() for l in leaves], vals, {n: net.engine().grad_of(p).cpu().numpy().astype(np.float64) for n, p in net.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "resenc"])
def test_hip_mixed_mode_within_the_reference_autocast_deviation(dev, tag):
    from multitalent_amd import ops
    z = np.load(os.path.join(G, 'autocast.npz'))
    ops.set_option('conv_bf16', 2)              # the 16-bit kernels also on the small grids of the toy networks
    try:
        logits, loss, grads = _hip(tag, dev)
    finally:
        ops.set_option('conv_bf16', 1)
        ops.set_mma(0)
    r16, rb = _ref_dev(z, tag, 'fp16'), _ref_dev(z, tag, 'bf16')
    names = _names(z, tag)
    g64 = _flat(z, '%s/fp64/grad/' % tag, names)
    g = np.concatenate([grads[n].reshape(-1) for n in names])
    gl2 = float(np.linalg.norm(g - g64) / np.linalg.norm(g64))
    gcos = float((g * g64).sum() / (np.linalg.norm(g) * np.linalg.norm(g64)))
    ll2 = [float(np.linalg.norm(a.astype(np.float64) - z['%s/fp32/out%d' % (tag, i)]) / np.linalg.norm(z['%s/fp32/out%d' % (tag, i)])) for i, a in enumerate(logits)]
    print("%s: HIP mixed vs exact: logits rel. L2 %s (reference fp16 autocast %s, bf16 %s); gradient rel. L2 %.4f cos %.5f (reference fp16 %.4f / %.5f, bf16 %.4f / %.5f)"
          % (tag, ['%.2e' % v for v in ll2], ['%.2e' % v for v in r16['logits']], ['%.2e' % v for v in rb['logits']], gl2, gcos,
             r16['grad_l2'], r16['grad_cos'], rb['grad_l2'], rb['grad_cos']))
    # forward: fp16 activations and products = the reference's autocast type
    for i, v in enumerate(ll2):
        assert v <= 1.5 * r16['logits'][i] + 1e-5, "level %d: %.3e against the reference autocast's %.3e" % (i, v, r16['logits'][i])
    for a, b, d in zip(loss, z['%s/fp32/loss' % tag], r16['loss']):
        assert abs(a - float(b)) <= 1.5 * float(d) + 2e-5 * max(1.0, abs(float(b)))
    # backward: within 1.5x the reference's fp16-autocast gradient deviation
    assert gl2 <= 1.5 * r16['grad_l2'] and (1.0 - gcos) <= 1.5 * (1.0 - r16['grad_cos']), (gl2, gcos, r16)
    # and against the autocast outputs themselves.  Two approximations of one exact answer with (nearly) independent rounding errors of
    # at most 1.5 r and r lie sqrt(1.5^2 + 1) r = 1.8 r apart; the triangle inequality (2.5 r) would hold for ANY pair that passed the
    # checks above and could never fail
    for i, a in enumerate(logits):
        b = z['%s/fp16/out%d' % (tag, i)]
        d = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        assert d <= 1.8 * r16['logits'][i] + 1e-5, "level %d: HIP mixed vs the reference's fp16-autocast logits %.3e, the autocast's own deviation %.3e" % (i, d, r16['logits'][i])
