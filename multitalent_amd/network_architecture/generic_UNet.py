"""Generic_UNet — drop-in for the reference's plain conv U-Net (generic_UNet.py:156-401).

Same constructor signature (called positionally at nnUNetTrainerV2.py:156-161), same nn.Module
hierarchy and therefore the same `state_dict()` keys / shapes / attribute paths
(conv_blocks_context.{s}.blocks.{j}.{conv,instnorm}, tu.{u}, conv_blocks_localization.{u}.{0,1}.blocks.0,
seg_outputs.{u}).  The sub-modules are PARAMETER HOLDERS: `forward` hands the whole network to the
HIP engine (multitalent_amd.engine) — fused conv+InstanceNorm+LeakyReLU blocks, transposed convs, heads
and their backward all run in libmtseg_hip.so.  Only the 3D, convolutional_pooling +
convolutional_upsampling configuration the north-star trainers use is implemented; anything else raises.
"""
import numpy as np
import torch
from torch import nn

from .initialization import InitWeights_He
from .neural_network import SegmentationNetwork


class ConvDropoutNormNonlin(nn.Module):
    """conv -> [dropout p=0: absent] -> InstanceNorm3d -> LeakyReLU (reference generic_UNet.py:28-70)."""

    def __init__(self, input_channels, output_channels, conv_op=nn.Conv3d, conv_kwargs=None, norm_op=nn.InstanceNorm3d,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None):
        super().__init__()
        if conv_op is not nn.Conv3d or norm_op is not nn.InstanceNorm3d or nonlin is not nn.LeakyReLU:
            raise NotImplementedError("the HIP engine implements Conv3d + InstanceNorm3d + LeakyReLU blocks only")
        if dropout_op is not None and dropout_op_kwargs is not None and dropout_op_kwargs.get('p', 0) > 0:
            raise NotImplementedError("dropout p > 0 is not used by the supported trainers (nnUNetTrainerV2.py:153)")
        self.conv_kwargs = dict(conv_kwargs)
        self.norm_op_kwargs = dict(norm_op_kwargs)
        self.nonlin_kwargs = dict(nonlin_kwargs)
        self.conv = nn.Conv3d(input_channels, output_channels, **self.conv_kwargs)
        self.dropout = None
        self.instnorm = nn.InstanceNorm3d(output_channels, **self.norm_op_kwargs)
        self.lrelu = nn.LeakyReLU(**self.nonlin_kwargs)


class StackedConvLayers(nn.Module):
    """num_convs blocks, the first with first_stride (reference generic_UNet.py:81-144)."""

    def __init__(self, input_feature_channels, output_feature_channels, num_convs, conv_op, conv_kwargs, norm_op,
                 norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs, first_stride=None,
                 basic_block=ConvDropoutNormNonlin):
        super().__init__()
        self.input_channels = input_feature_channels
        self.output_channels = output_feature_channels
        first_kwargs = dict(conv_kwargs)
        if first_stride is not None:
            first_kwargs['stride'] = first_stride
        blocks = [basic_block(input_feature_channels, output_feature_channels, conv_op, first_kwargs, norm_op,
                              norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs)]
        for _ in range(num_convs - 1):
            blocks.append(basic_block(output_feature_channels, output_feature_channels, conv_op, conv_kwargs, norm_op,
                                      norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs))
        self.blocks = nn.Sequential(*blocks)


class Generic_UNet(SegmentationNetwork):
    DEFAULT_BATCH_SIZE_3D = 2
    DEFAULT_PATCH_SIZE_3D = (64, 192, 160)
    SPACING_FACTOR_BETWEEN_STAGES = 2
    BASE_NUM_FEATURES_3D = 30
    MAX_NUMPOOL_3D = 999
    MAX_NUM_FILTERS_3D = 320

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2,
                 feat_map_mul_on_downscale=2, conv_op=nn.Conv3d, norm_op=nn.InstanceNorm3d, norm_op_kwargs=None,
                 dropout_op=nn.Dropout3d, dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None,
                 deep_supervision=True, dropout_in_localization=False, final_nonlin=lambda x: x,
                 weightInitializer=InitWeights_He(1e-2), pool_op_kernel_sizes=None, conv_kernel_sizes=None,
                 upscale_logits=False, convolutional_pooling=False, convolutional_upsampling=False,
                 max_num_features=None, basic_block=ConvDropoutNormNonlin, seg_output_use_bias=False,
                 internal_conv_bias=True):
        super().__init__()
        if conv_op is not nn.Conv3d:
            raise NotImplementedError("only the 3D network is on the hot path (MultiTalent is 3D only)")
        if not (convolutional_pooling and convolutional_upsampling):
            raise NotImplementedError("only convolutional_pooling=True, convolutional_upsampling=True "
                                      "(nnUNetTrainerV2.py:156-161) is implemented")
        if upscale_logits or num_conv_per_stage != 2 or basic_block is not ConvDropoutNormNonlin:
            raise NotImplementedError("unsupported Generic_UNet option for the HIP engine")
        nonlin_kwargs = {'negative_slope': 1e-2, 'inplace': True} if nonlin_kwargs is None else nonlin_kwargs
        dropout_op_kwargs = {'p': 0.5, 'inplace': True} if dropout_op_kwargs is None else dropout_op_kwargs
        norm_op_kwargs = {'eps': 1e-5, 'affine': True, 'momentum': 0.1} if norm_op_kwargs is None else norm_op_kwargs
        if dropout_op_kwargs.get('p', 0) != 0:
            raise NotImplementedError("dropout p > 0 unsupported")

        self.convolutional_upsampling = convolutional_upsampling
        self.convolutional_pooling = convolutional_pooling
        self.upscale_logits = upscale_logits
        self.conv_kwargs = {'stride': 1, 'dilation': 1, 'bias': internal_conv_bias}
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.norm_op, self.norm_op_kwargs = norm_op, norm_op_kwargs
        self.weightInitializer = weightInitializer
        self.conv_op = conv_op
        self.num_classes = num_classes
        self.final_nonlin = final_nonlin
        self._deep_supervision = deep_supervision
        self.do_ds = deep_supervision

        if pool_op_kernel_sizes is None:
            pool_op_kernel_sizes = [(2, 2, 2)] * num_pool
        if conv_kernel_sizes is None:
            conv_kernel_sizes = [(3, 3, 3)] * (num_pool + 1)
        self.input_shape_must_be_divisible_by = np.prod(pool_op_kernel_sizes, 0, dtype=np.int64)
        self.pool_op_kernel_sizes = pool_op_kernel_sizes
        self.conv_kernel_sizes = conv_kernel_sizes
        self.conv_pad_sizes = [[1 if i == 3 else 0 for i in k] for k in conv_kernel_sizes]
        self.max_num_features = self.MAX_NUM_FILTERS_3D if max_num_features is None else max_num_features

        def stacked(cin, cout, n, level, first_stride=None):
            kw = dict(self.conv_kwargs)
            kw['kernel_size'] = self.conv_kernel_sizes[level]
            kw['padding'] = self.conv_pad_sizes[level]
            return StackedConvLayers(cin, cout, n, conv_op, kw, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs,
                                     nonlin, nonlin_kwargs, first_stride, basic_block=basic_block)

        context, localization, tu, seg = [], [], [], []
        cin, cout = input_channels, base_num_features
        for d in range(num_pool):
            context.append(stacked(cin, cout, num_conv_per_stage, d, pool_op_kernel_sizes[d - 1] if d != 0 else None))
            cin = cout
            cout = min(int(np.round(cout * feat_map_mul_on_downscale)), self.max_num_features)
        final_num_features = cout
        context.append(nn.Sequential(stacked(cin, cout, num_conv_per_stage - 1, num_pool, pool_op_kernel_sizes[-1]),
                                     stacked(cout, final_num_features, 1, num_pool)))
        for u in range(num_pool):
            nfeatures_from_down = final_num_features
            nfeatures_from_skip = context[-(2 + u)].output_channels
            final_num_features = nfeatures_from_skip
            tu.append(nn.ConvTranspose3d(nfeatures_from_down, nfeatures_from_skip, pool_op_kernel_sizes[-(u + 1)],
                                         pool_op_kernel_sizes[-(u + 1)], bias=False))
            lvl = num_pool - u      # reference generic_UNet.py:338-339 indexes conv_kernel_sizes[-(u + 1)] (num_pool + 1 entries): the
                                    # first decoder stage reuses the bottleneck's kernel — an nnU-Net v1 quirk that fixes weight shapes
            localization.append(nn.Sequential(stacked(nfeatures_from_skip * 2, nfeatures_from_skip, num_conv_per_stage - 1, lvl),
                                              stacked(nfeatures_from_skip, final_num_features, 1, lvl)))
        for ds in range(len(localization)):
            seg.append(nn.Conv3d(localization[ds][-1].output_channels, num_classes, 1, 1, 0, 1, 1, seg_output_use_bias))
        self.upscale_logits_ops = [lambda x: x for _ in range(num_pool - 1)]

        # registration order = reference order (generic_UNet.py:365-369): keeps parameter iteration order identical
        self.conv_blocks_localization = nn.ModuleList(localization)
        self.conv_blocks_context = nn.ModuleList(context)
        self.td = nn.ModuleList([])
        self.tu = nn.ModuleList(tu)
        self.seg_outputs = nn.ModuleList(seg)
        if self.weightInitializer is not None:
            self.apply(self.weightInitializer)
        self._engine = None

    # ---- engine hand-off ---------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            from ..engine import build_plain_unet_engine
            self._engine = build_plain_unet_engine(self)
        return self._engine

    def forward(self, x):
        outs = self.engine().apply(x, all_heads=bool(self._deep_supervision and self.do_ds))
        outs = [self.final_nonlin(o) for o in outs]
        if self._deep_supervision and self.do_ds:
            return tuple(outs)
        return outs[0]
