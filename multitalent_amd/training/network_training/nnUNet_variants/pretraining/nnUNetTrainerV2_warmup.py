"""Fine-tuning trainers (reference nnUNet_variants/pretraining/nnUNetTrainerV2_warmup.py:38-199,441-621; the
readme's downstream use-case, readme.md:51-64): after `load_pretrained_weights` the new segmentation heads are trained alone for
`warmup_duration` epochs (AdamW 3e-3 amsgrad, lr ramped 0 -> warmup_max_lr), then the whole network with SGD-Nesterov whose lr
ramps linearly for `num_epochs_sgd_warmup` epochs before the usual poly schedule.

Every phase runs on the same fused hot loop: forward, loss, FULL backward and the clip norm over ALL parameters are identical
(the reference clips the heads' gradients by the whole network's norm, nnUNetTrainerV2.py:262-264), only the update differs —
`FusedTrainStep.set_head_optimizer` swaps the fused SGD kernel for AdamW on the (tiny) head tensors."""
import numpy as np
import torch
from torch import nn

from ...nnUNetTrainer import nnUNetTrainerV2, poly_lr
from .....network_architecture.generic_modular_residual_UNet import FabiansUNet, get_default_network_config
from .....network_architecture.initialization import InitWeights_He
from .....utilities.nd_softmax import softmax_helper


class nnUNetTrainerV2_warmup_increasing_lr(nnUNetTrainerV2):
    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False):
        super().__init__(plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data, deterministic, fp16)
        self.warmup_duration = 50
        self.max_num_epochs = 1000 + self.warmup_duration

    def _set_lr(self, lr):
        self.optimizer_lr = lr
        self.train_step.lr = lr

    def maybe_update_lr(self, epoch=None):
        if self.epoch < self.warmup_duration:                       # linear 0 -> initial_lr, epoch 49 reaches it (reference :46-52)
            lr = (self.epoch + 1) / self.warmup_duration * self.initial_lr
            self._set_lr(lr)
            self.print_to_log_file("epoch:", self.epoch, "lr:", lr)
        else:
            ep = (epoch if epoch is not None else self.epoch) - (self.warmup_duration - 1)
            assert ep > 0, "epoch must be >0"
            return super().maybe_update_lr(ep)


class nnUNetTrainerV2_warmupsegheads(nnUNetTrainerV2_warmup_increasing_lr):
    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False):
        super().__init__(plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data, deterministic, fp16)
        self.num_epochs_sgd_warmup = 50          # linear warm-up of the whole network
        self.warmup_max_lr = 5e-4                # for the heads
        self.warmup_duration = 10                # epochs of heads-only training
        self.max_num_epochs = 1000 + self.num_epochs_sgd_warmup + self.warmup_duration

    def head_parameters(self):
        return list(self.network.seg_outputs.parameters())                       # reference :123

    def initialize(self, training=True, force_load_plans=False):
        super().initialize(training, force_load_plans)
        if training:
            self.initialize_optimizer_and_scheduler(True)

    def initialize_optimizer_and_scheduler(self, seg_heads_only=False):
        """reference :119-132: AdamW(3e-3, amsgrad) on the heads, or SGD(initial_lr, 0.99, nesterov) on everything."""
        if self.train_step is None:
            super().initialize_optimizer_and_scheduler()
        if seg_heads_only:
            self.train_step.set_head_optimizer(self.head_parameters(), lr=3e-3, weight_decay=self.weight_decay)
        else:
            self.train_step.set_head_optimizer(None)
            self.train_step.reset_momentum()                                     # a NEW SGD instance in the reference
            self.train_step.lr = self.initial_lr
        self.lr_scheduler = None

    def maybe_update_lr(self, epoch=None):
        if self.epoch < self.warmup_duration:                                     # reference :91-96
            lr = (self.epoch + 1) / self.warmup_duration * self.warmup_max_lr
            self._set_lr(lr)
            self.lr = lr
            self.print_to_log_file("epoch:", self.epoch, "lr for heads:", lr)
        elif self.warmup_duration <= self.epoch < self.warmup_duration + self.num_epochs_sgd_warmup:
            lr = (self.epoch - self.warmup_duration + 1) / self.num_epochs_sgd_warmup * self.initial_lr
            self._set_lr(lr)
            self.print_to_log_file("epoch:", self.epoch, "lr now lin increasing whole network:", lr)
        else:
            ep = (epoch if epoch is not None else self.epoch) - (self.warmup_duration + self.num_epochs_sgd_warmup - 1)
            assert ep > 0, "epoch must be >0"
            self._set_lr(poly_lr(ep, self.max_num_epochs - self.num_epochs_sgd_warmup - self.warmup_duration, self.initial_lr, 0.9))
            self.print_to_log_file("lr was set to:", np.round(self.optimizer_lr, decimals=6))

    def on_epoch_end(self):
        if self.epoch == self.warmup_duration:                                    # reference :112-116
            self.print_to_log_file("now train whole network")
            self.initialize_optimizer_and_scheduler(seg_heads_only=False)
        return super().on_epoch_end()

    def load_checkpoint_ram(self, checkpoint, train=True):
        """the optimizer that matches the checkpoint's epoch must exist before its state is restored (reference :134-199)."""
        if not self.was_initialized:
            self.initialize(train)
        if train and checkpoint.get('epoch', 0) > self.warmup_duration:
            self.initialize_optimizer_and_scheduler(seg_heads_only=False)
        return super().load_checkpoint_ram(checkpoint, train)


class nnUNetTrainerV2_warmupsegheads_resenc(nnUNetTrainerV2_warmupsegheads):
    """Residual-encoder variant (reference :441-551): FabiansUNet, heads = decoder.deep_supervision_outputs."""

    def initialize_network(self):
        cfg = get_default_network_config(3, None, norm_type="in")
        sp = self.plans['plans_per_stage'][self.stage]
        self.network = FabiansUNet(self.num_input_channels, self.base_num_features, sp['num_blocks_encoder'], 2,
                                   sp['pool_op_kernel_sizes'], sp['conv_kernel_sizes'], cfg, self.num_classes,
                                   sp['num_blocks_decoder'], True, False, 320, InitWeights_He(1e-2))
        if torch.cuda.is_available():
            self.network.cuda()
        self.network.inference_apply_nonlin = softmax_helper
        from ...custom_trainers.MultiTalent.MultiTalent.MultiTalent_meets_resenc import init_last_bn_before_add_to_0
        self.network.apply(init_last_bn_before_add_to_0)

    def head_parameters(self):
        return list(self.network.decoder.deep_supervision_outputs.parameters())   # reference :477

    def setup_DA_params(self):
        self.deep_supervision_scales = [[1, 1, 1]] + list(list(i) for i in 1 / np.cumprod(
            np.vstack(self.net_num_pool_op_kernel_sizes[1:]), axis=0))[:-1]
