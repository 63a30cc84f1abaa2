"""MultiTalent trainers with the residual-encoder U-Net (reference MultiTalent_meets_resenc.py:31-116,1021-1052)."""
import numpy as np
import torch
from torch import nn

from .MultiTalent_Trainer_DDP import MultiTalent_trainer_ddp
from ......network_architecture.generic_modular_residual_UNet import BasicResidualBlock, FabiansUNet, get_default_network_config
from ......network_architecture.initialization import InitWeights_He


def init_last_bn_before_add_to_0(module):
    """reference :31-34 — norm2 of every residual block starts at gamma = beta = 0."""
    if isinstance(module, BasicResidualBlock):
        nn.init.constant_(module.norm2.weight, 0)
        nn.init.constant_(module.norm2.bias, 0)


class MultiTalent_trainer_resenc_ddp(MultiTalent_trainer_ddp):
    def initialize_network(self):
        """reference :72-104 — same positional FabiansUNet call."""
        cfg = get_default_network_config(3, None, norm_type="in")
        sp = self.plans["plans_per_stage"][self.stage]
        torch.manual_seed(1234)
        self.network = FabiansUNet(self.num_input_channels, self.base_num_features, sp["num_blocks_encoder"], 2,
                                   sp["pool_op_kernel_sizes"], sp["conv_kernel_sizes"], cfg, self.num_classes,
                                   sp["num_blocks_decoder"], True, False, 320, InitWeights_He(1e-2))
        self.network.apply(init_last_bn_before_add_to_0)
        if torch.cuda.is_available():
            self.network.cuda()
        self.network.inference_apply_nonlin = nn.Sigmoid()

    def setup_DA_params(self):
        """net_num_pool_op_kernel_sizes includes the stem's [1,1,1]: scales skip it (reference :107-116)."""
        self.deep_supervision_scales = [[1, 1, 1]] + list(list(i) for i in 1 / np.cumprod(
            np.vstack(self.net_num_pool_op_kernel_sizes[1:]), axis=0))[:-1]

    @staticmethod
    def _get_ds(net):
        return net.decoder.deep_supervision                                   # reference :458-460

    @staticmethod
    def _set_ds(net, v):
        net.decoder.deep_supervision = v


class MultiTalent_trainer_resenc_ddp_2000ep(MultiTalent_trainer_resenc_ddp):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_num_epochs = 2000
