"""Fine-tuning path against golden vectors from the REAL reference (tools/oracle_gen/make_golden_finetune.py):
learning-rate schedules and load_pretrained_weights on CPU; heads-only AdamW iterations and the switch to whole-network SGD on
the GPU through the fused hot loop."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _plans(z):
    from multitalent_amd import plans as P
    sp = {'batch_size': 2, 'patch_size': np.array(z['x'].shape[2:]), 'pool_op_kernel_sizes': z['pools'].tolist(),
          'conv_kernel_sizes': z['kernels'].tolist(), 'do_dummy_2D_data_aug': False}
    return P.make_plans(sp, base_num_features=6, num_classes=3, stage=0)


def test_warmup_lr_schedules_match_reference():
    from multitalent_amd.training.model_restore import find_trainer_class
    z = np.load(os.path.join(G, 'finetune.npz'))
    p = dict(np.load(os.path.join(G, 'plain_unet.npz')))
    for name, key in (('nnUNetTrainerV2_warmup_increasing_lr', 'lr_warmup_increasing'), ('nnUNetTrainerV2_warmupsegheads', 'lr_warmupsegheads'),
                      ('nnUNetTrainerV2_warmupsegheads_resenc', 'lr_warmupsegheads')):
        tr = find_trainer_class(name)(_plans(p), 0, output_folder=None, stage=0)
        tr.train_step = SimpleNamespace(lr=None)
        tr.log_file = os.devnull
        got = []
        for ep in z['epochs'][:len(z[key])]:
            tr.epoch = int(ep)
            tr.maybe_update_lr()
            assert tr.train_step.lr == tr.optimizer_lr
            got.append(tr.optimizer_lr)
        assert np.allclose(got, z[key], rtol=1e-12, atol=0), (name, got, z[key])
    assert tr.max_num_epochs == 1060 and tr.warmup_duration == 10 and tr.num_epochs_sgd_warmup == 50


def test_load_pretrained_weights_matches_reference(tmp_path):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.run.load_pretrained_weights import load_pretrained_weights
    z = np.load(os.path.join(G, 'finetune.npz'))
    p = dict(np.load(os.path.join(G, 'plain_unet.npz')))
    build = lambda nc, base=6: Generic_UNet(1, base, nc, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                                            nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                                            True, False, lambda x: x, None, p['pools'].tolist(), p['kernels'].tolist(), False, True, True)
    torch.manual_seed(5)
    src = build(7)
    for q in src.parameters():
        q.data.add_(0.25 * torch.randn(q.shape))
    f = str(tmp_path / 'pre.model')
    torch.save({'state_dict': {'module.' + k: v for k, v in src.state_dict().items()}}, f)
    dst = build(4)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    load_pretrained_weights(dst, f)
    after = dst.state_dict()
    assert [k for k in after if not torch.equal(after[k], before[k])] == [str(k) for k in z['pre_transferred']]
    assert [k for k in after if torch.equal(after[k], before[k])] == [str(k) for k in z['pre_kept']]
    assert all(torch.equal(after[str(k)], src.state_dict()[str(k)]) for k in z['pre_transferred'])
    with pytest.raises(RuntimeError):
        load_pretrained_weights(build(4, base=8), f)


@pytest.mark.gpu
def test_heads_only_then_whole_network_matches_reference(dev):
    """nnUNetTrainerV2_warmupsegheads: 3 heads-only iterations (AdamW amsgrad, lr ramp, clip by the WHOLE network's gradient norm),
    the switch at epoch == warmup_duration, 2 SGD iterations with the linear warm-up lr.  Losses 1e-4, parameters 1e-5 / 1e-4."""
    from multitalent_amd.training.model_restore import find_trainer_class
    z = np.load(os.path.join(G, 'finetune.npz'))
    p = dict(np.load(os.path.join(G, 'plain_unet.npz')))
    tr = find_trainer_class('nnUNetTrainerV2_warmupsegheads')(_plans(p), 0, output_folder=None, stage=0, batch_dice=False)
    tr.log_file = os.devnull
    tr.initialize(True)
    assert np.allclose(tr.ds_loss_weights, p['weights'])
    sd0 = {k[4:]: torch.from_numpy(v) for k, v in p.items() if k.startswith('sd0/')}
    tr.network.load_state_dict(sd0)
    tr.network.engine().mark_params_dirty()
    tr.network.train()
    batch = {'data': torch.from_numpy(p['x']).to(dev), 'target': [torch.from_numpy(p['target%d' % i]).to(dev) for i in range(3)]}
    gen = iter(lambda: batch, None)
    for it in range(3):
        tr.epoch = it
        tr.maybe_update_lr()
        l = float(tr.run_iteration(gen, True))
        assert abs(l - z['heads_losses'][it]) < 1e-4, (it, l, z['heads_losses'][it])
    moved = []
    for k, v in tr.network.state_dict().items():
        if k.startswith('seg_outputs'):
            assert np.abs(v.cpu().numpy() - z['sd_heads/' + k]).max() < 1e-5, k
            moved.append(float(np.abs(v.cpu().numpy() - sd0[k].numpy()).max()))
        else:
            assert torch.equal(v.cpu(), sd0[k]), k                                  # nothing else did
    assert max(moved) > 1e-4 and min(moved) < 1e-6        # the weighted heads moved; the zero-weight lowest level only decays
    tr.epoch = 10
    tr.on_epoch_end()                                                               # "now train whole network"
    assert tr.train_step.head_opt is None
    for it in range(2):
        tr.epoch = 10 + it
        tr.maybe_update_lr()
        l = float(tr.run_iteration(gen, True))
        assert abs(l - z['sgd_losses'][it]) < 1e-4, (it, l, z['sgd_losses'][it])
    for k, v in tr.network.state_dict().items():
        assert np.abs(v.cpu().numpy() - z['sd_sgd/' + k]).max() < 1e-4, k
