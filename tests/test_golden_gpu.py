"""HIP path vs golden vectors produced by the REAL reference (tools/oracle_gen/make_golden.py): logits, losses,
gradients, and parameters after two full training iterations (fwd + loss + bwd + clip 12 + SGD-Nesterov).
Tolerance 1e-3 (north_star), tightened where fp32 allows."""
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

from mask_check import check_masks

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(G, name)))


def test_plain_unet_two_training_iterations(dev):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    z = load('plain_unet.npz')
    pools, kernels = z['pools'].tolist(), z['kernels'].tolist()
    net = Generic_UNet(1, 6, 4, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd0/')})
    net.train()
    x = torch.from_numpy(z['x']).to(dev)
    tg = [torch.from_numpy(z['target%d' % i]).to(dev) for i in range(3)]
    out = net(x)
    for i, o in enumerate(out):
        assert float((o.detach().cpu() - torch.from_numpy(z['out%d' % i])).abs().max()) < 1e-4
    # DDP-flavour loss at world size 1 (nnUNetTrainerV2_DDP.compute_loss)
    for bd, key in ((True, 'loss_ddp_batchdice'), (False, 'loss_ddp_nobatchdice')):
        l = DC_and_CE_DS_loss(z['weights'], batch_dice=bd, ddp=True)(out, tg)
        assert abs(float(l.detach()) - float(z[key])) < 1e-4
    step = FusedTrainStep(net, DC_and_CE_DS_loss(z['weights'], batch_dice=False), lr=1e-2)
    for it in range(2):
        l = step(x, tg)
        assert abs(float(l) - z['losses'][it]) < 1e-4, (it, float(l), z['losses'][it])
        if it == 0:
            eng = net.engine()
            for n, p in net.named_parameters():
                if ('grad0/' + n) in z:
                    # flat_grad holds the unclipped gradient of this iteration
                    g = eng.grad_of(p).cpu().numpy()
                    ref = z['grad0/' + n]
                    assert np.abs(g - ref).max() < 2e-3 * max(np.abs(ref).max(), 1e-3), n
    torch.cuda.synchronize()
    for k, v in net.state_dict().items():
        assert np.abs(v.cpu().numpy() - z['sd2/' + k]).max() < 1e-4, k
    net.eval(); net.do_ds = False
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd0/')})
    net.engine().mark_params_dirty()
    with torch.no_grad():
        o = net(x)
    assert float((o.cpu() - torch.from_numpy(z['out_infer'])).abs().max()) < 1e-4


def test_resenc_unet_logits_loss_gradients(dev):
    from multitalent_amd.network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    z = load('resenc_unet.npz')
    valid = json.load(open(os.path.join(G, 'resenc_unet_valid.json')))['valid_regions']
    pools, kernels, blocks = z['pools'].tolist(), z['kernels'].tolist(), z['blocks'].tolist()
    net = FabiansUNet(1, 6, blocks, 2, pools, kernels, get_default_network_config(3, None, norm_type="in"), 47, [1, 1, 1],
                      True, False, 16, None)
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd0/')})
    net.train()
    x = torch.from_numpy(z['x']).to(dev)
    tg = [torch.from_numpy(z['target%d' % i]).to(dev) for i in range(3)]
    out = net(x)
    for i, o in enumerate(out):
        assert float((o.detach().cpu() - torch.from_numpy(z['out%d' % i])).abs().max()) < 1e-4
    l, ce, dc = MultiTalentLoss(z['weights'], batch_dice=True)(out, tg, valid)
    assert np.allclose([float(l.detach()), float(ce.detach()), float(dc.detach())], z['loss'], rtol=1e-4)
    l.backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        ref = z['grad0/' + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 2e-3 * max(np.abs(ref).max(), 1e-3), n


def test_multitalent_loss_kernel_vs_reference_dlogits(dev):
    from multitalent_amd.training.loss_functions.fused_losses import MultiTalentLoss
    z = load('multitalent_loss.npz')
    valid = json.load(open(os.path.join(G, 'multitalent_loss_valid.json')))['valid_regions']
    logits = [torch.from_numpy(z['logits%d' % i]).to(dev).requires_grad_(True) for i in range(2)]
    tg = [torch.from_numpy(z['target%d' % i]).to(dev) for i in range(2)]
    l, ce, dc = MultiTalentLoss(z['weights'], batch_dice=True)(logits, tg, valid)
    assert np.allclose([float(l.detach()), float(ce.detach()), float(dc.detach())], z['bd1/loss'], rtol=1e-4)
    l.backward()
    for i in range(2):
        ref = z['bd1/dlogits%d' % i]
        assert np.abs(logits[i].grad.cpu().numpy() - ref).max() < 1e-3 * np.abs(ref).max()


def test_sliding_window_predict_3d_vs_reference(dev):
    """predict_3D (mirror TTA on/off, Gaussian blending, sigmoid+regions / softmax+argmax) vs the real reference's output."""
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.utilities.nd_softmax import softmax_helper
    z = load('sliding_window.npz')
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    for tag, nc, nonlin, order in (('mt', 5, nn.Sigmoid(), [3, 1, 4, 2, 5]), ('sm', 3, softmax_helper, None)):
        net = Generic_UNet(1, 6, nc, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                           {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                           lambda x: x, None, pools, kernels, False, True, True)
        pre = tag + '/sd/'
        net.load_state_dict({k[len(pre):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(pre)})
        net.to(dev)
        net.inference_apply_nonlin = nonlin
        net.eval(); net.do_ds = False
        for mirror in (True, False):
            seg, probs = net.predict_3D(z[tag + '/vol'], do_mirroring=mirror, mirror_axes=(0, 1, 2), use_sliding_window=True,
                                        step_size=0.5, patch_size=(8, 16, 16), regions_class_order=order, use_gaussian=True,
                                        pad_border_mode='constant', pad_kwargs={'constant_values': 0}, all_in_gpu=False,
                                        verbose=False, mixed_precision=False)
            ref_p, ref_s = z['%s/probs_m%d' % (tag, int(mirror))], z['%s/seg_m%d' % (tag, int(mirror))]
            assert probs.shape == ref_p.shape and seg.shape == ref_s.shape
            assert np.abs(probs - ref_p).max() < 1e-4
            # every voxel accounted for: identical away from ties, the reference's decision rule on our own probabilities
            # everywhere, identical wherever the probabilities are bit-identical; tie voxels counted and printed
            check_masks(seg, ref_s, probs, ref_p, order, 1e-4, 'predict_3D %s mirror=%d' % (tag, int(mirror)))


def test_plain_unet_nonuniform_kernel_sizes(dev):
    """conv_kernel_sizes [[1,3,3],[3,3,3],[3,3,3],[1,3,3]]: the first decoder stage takes the bottleneck's kernel
    (generic_UNet.py:338-339).  Logits, loss and every gradient vs the real reference."""
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    z = load('plain_unet_aniso.npz')
    pools, kernels = z['pools'].tolist(), z['kernels'].tolist()
    net = Generic_UNet(1, 6, 3, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd0/')}, strict=True)
    net.train()
    x = torch.from_numpy(z['x']).to(dev)
    tg = [torch.from_numpy(z['target%d' % i]).to(dev) for i in range(3)]
    out = net(x)
    for i, o in enumerate(out):
        assert float((o.detach().cpu() - torch.from_numpy(z['out%d' % i])).abs().max()) < 1e-4
    l = DC_and_CE_DS_loss(z['weights'], batch_dice=False)(out, tg)
    assert abs(float(l.detach()) - float(z['loss'])) < 1e-4
    l.backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        ref = z['grad0/' + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 2e-3 * max(np.abs(ref).max(), 1e-3), n


def test_sliding_window_volume_smaller_than_patch(dev):
    """pad_nd_image path (neural_network.py:301): volumes smaller than the patch along one or two axes, odd differences; host and
    device-resident input."""
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    z = load('sliding_window_pad.npz')
    pools, kernels = [[2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 3
    net = Generic_UNet(1, 6, 5, 2, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, pools, kernels, False, True, True)
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd/')})
    net.to(dev)
    net.inference_apply_nonlin = nn.Sigmoid()
    net.eval(); net.do_ds = False
    for tag in ('a', 'b'):
        for mirror in (True, False):
            for on_device in (False, True):
                vol = torch.from_numpy(z[tag + '/vol']).to(dev) if on_device else z[tag + '/vol']
                seg, probs = net.predict_3D(vol, do_mirroring=mirror, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5,
                                            patch_size=(8, 16, 16), regions_class_order=[3, 1, 4, 2, 5], use_gaussian=True,
                                            pad_border_mode='constant', pad_kwargs={'constant_values': 0}, all_in_gpu=False,
                                            verbose=False, mixed_precision=False)
                ref_p, ref_s = z['%s/probs_m%d' % (tag, int(mirror))], z['%s/seg_m%d' % (tag, int(mirror))]
                assert probs.shape == ref_p.shape and seg.shape == ref_s.shape
                assert np.abs(probs - ref_p).max() < 1e-4
                check_masks(seg, ref_s, probs, ref_p, [3, 1, 4, 2, 5], 1e-4, 'predict_3D pad %s mirror=%d device=%d' % (tag, int(mirror), int(on_device)))
