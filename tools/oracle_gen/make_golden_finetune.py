"""Golden vectors for the fine-tuning path from the REAL reference, build container only:
  * the learning-rate schedules of nnUNetTrainerV2_warmup_increasing_lr / nnUNetTrainerV2_warmupsegheads (maybe_update_lr),
  * three heads-only iterations (AdamW amsgrad on network.seg_outputs, gradient clipping over ALL parameters) of
    nnUNetTrainerV2_warmupsegheads on the plain_unet.npz inputs, then two whole-network SGD iterations after the switch,
  * load_pretrained_weights between networks with different numbers of classes.
Writes tests/golden/finetune.npz.  Run: python tools/oracle_gen/make_golden_finetune.py"""
import os, sys, tempfile
from types import SimpleNamespace
import numpy as np
import torch
from torch import nn
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import
ref_import.install()
from nnunet.network_architecture.generic_UNet import Generic_UNet
from nnunet.training.loss_functions.deep_supervision import MultipleOutputLoss2
from nnunet.training.loss_functions.dice_loss import DC_and_CE_loss
import nnunet.training.network_training.nnUNet_variants.pretraining.nnUNetTrainerV2_warmup as W
from nnunet.run.load_pretrained_weights import load_pretrained_weights

G = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden'))


def build(nc, pools, kernels):
    return Generic_UNet(1, 6, nc, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                        {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                        lambda x: x, None, pools, kernels, False, True, True)


def lr_table(cls, attrs, epochs):
    out = []
    for ep in epochs:
        opt = SimpleNamespace(param_groups=[{'lr': -1.0}])
        fake = SimpleNamespace(epoch=ep, optimizer=opt, initial_lr=1e-2, print_to_log_file=lambda *a, **k: None, **attrs)
        if cls is W.nnUNetTrainerV2_warmup_increasing_lr and ep >= attrs['warmup_duration']:
            # delegates to nnUNetTrainerV2.maybe_update_lr(ep) = poly_lr(ep, max_num_epochs, initial_lr, 0.9)
            from nnunet.training.learning_rate.poly_lr import poly_lr
            out.append(poly_lr(ep - (attrs['warmup_duration'] - 1), attrs['max_num_epochs'], 1e-2, 0.9))
            continue
        cls.maybe_update_lr(fake)
        out.append(opt.param_groups[0]['lr'])
    return np.array(out, dtype=np.float64)


def main():
    z = dict(np.load(os.path.join(G, 'plain_unet.npz')))
    pools, kernels = z['pools'].tolist(), z['kernels'].tolist()
    rec = {}
    epochs = [0, 1, 4, 9, 10, 11, 30, 49, 50, 59, 60, 61, 100, 500, 1049, 1059]
    rec['epochs'] = np.array(epochs)
    rec['lr_warmup_increasing'] = lr_table(W.nnUNetTrainerV2_warmup_increasing_lr, dict(warmup_duration=50, max_num_epochs=1050),
                                           [e for e in epochs if e < 1050])
    rec['lr_warmupsegheads'] = lr_table(W.nnUNetTrainerV2_warmupsegheads,
                                        dict(warmup_duration=10, num_epochs_sgd_warmup=50, warmup_max_lr=5e-4, max_num_epochs=1060), epochs)

    # ---- heads-only AdamW iterations, then the switch to whole-network SGD
    torch.manual_seed(0)
    net = build(4, pools, kernels)
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd0/')})
    net.train()
    x = torch.from_numpy(z['x']); tg = [torch.from_numpy(z['target%d' % i]) for i in range(3)]
    loss = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), z['weights'])
    fake = SimpleNamespace(network=net, weight_decay=3e-5, initial_lr=1e-2, lr_scheduler=None, optimizer=None,
                           epoch=0, warmup_duration=10, num_epochs_sgd_warmup=50, warmup_max_lr=5e-4, max_num_epochs=1060,
                           print_to_log_file=lambda *a, **k: None)
    W.nnUNetTrainerV2_warmupsegheads.initialize_optimizer_and_scheduler(fake, True)
    assert type(fake.optimizer).__name__ == 'AdamW'
    losses = []
    for it in range(3):
        fake.epoch = it                        # one iteration per "epoch" so that the lr ramp is exercised: 5e-5, 1e-4, 1.5e-4
        W.nnUNetTrainerV2_warmupsegheads.maybe_update_lr(fake)
        fake.optimizer.zero_grad()
        l = loss(net(x), tg)
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)          # nnUNetTrainerV2.run_iteration :262-264
        fake.optimizer.step()
        losses.append(float(l))
    rec['heads_losses'] = np.array(losses)
    sd0 = {k[4:]: v for k, v in z.items() if k.startswith('sd0/')}
    for k, v in net.state_dict().items():
        if k.startswith('seg_outputs'):
            rec['sd_heads/' + k] = v.detach().numpy().copy()
        else:
            assert np.array_equal(v.detach().numpy(), sd0[k]), k        # everything but the heads is untouched (not stored)
    fake.epoch = 10
    W.nnUNetTrainerV2_warmupsegheads.initialize_optimizer_and_scheduler(fake, False)     # on_epoch_end at epoch == warmup_duration
    losses = []
    for it in range(2):
        fake.epoch = 10 + it
        W.nnUNetTrainerV2_warmupsegheads.maybe_update_lr(fake)        # 1/50, 2/50 of initial_lr
        fake.optimizer.zero_grad()
        l = loss(net(x), tg)
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        fake.optimizer.step()
        losses.append(float(l))
    rec['sgd_losses'] = np.array(losses)
    for k, v in net.state_dict().items():
        rec['sd_sgd/' + k] = v.detach().numpy().copy()

    # ---- load_pretrained_weights: 7-class checkpoint (DDP-style 'module.' prefix) into the 4-class network
    torch.manual_seed(5)
    src = build(7, pools, kernels)
    for p in src.parameters():
        p.data.add_(0.25 * torch.randn(p.shape))          # norm weights/biases too, so that every transferred tensor differs
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, 'pre.model')
        torch.save({'state_dict': {'module.' + k: v for k, v in src.state_dict().items()}}, f)
        dst = build(4, pools, kernels)
        before = {k: v.clone() for k, v in dst.state_dict().items()}
        load_pretrained_weights(dst, f)
    rec['pre_transferred'] = np.array([k for k, v in dst.state_dict().items() if not torch.equal(v, before[k])])
    rec['pre_kept'] = np.array([k for k, v in dst.state_dict().items() if torch.equal(v, before[k])])
    np.savez_compressed(os.path.join(G, 'finetune.npz'), **rec)
    print('wrote finetune.npz', os.path.getsize(os.path.join(G, 'finetune.npz')) // 1024, 'KiB')
    print(rec['lr_warmupsegheads'][:8], rec['heads_losses'], rec['sgd_losses'], rec['pre_kept'])


if __name__ == '__main__':
    main()
