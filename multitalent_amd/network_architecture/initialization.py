"""Weight initialisation matching the reference's InitWeights_He (initialization.py:19-27):
kaiming_normal_(a=neg_slope) on conv / transposed-conv weights, zero biases.  Host-side torch glue."""
from torch import nn


class InitWeights_He(object):
    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module):
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)
