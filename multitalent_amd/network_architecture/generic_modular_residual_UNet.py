"""FabiansUNet — drop-in for the reference's residual-encoder U-Net used by
MultiTalent_trainer_resenc_ddp (generic_modular_residual_UNet.py:28-118,320-358; conv_blocks.py:49-85,
116-213,330-357; generic_modular_UNet.py:31-78,185-291).  Parameter-holder modules with the reference's
hierarchy/`state_dict` keys (encoder.initial_conv, encoder.stages.{s}.convs.{b}.{conv1,norm1,conv2,norm2,
downsample_skip.{0,1}}, decoder.tus.{i}, decoder.stages.{i}.convs.0.{conv,norm},
decoder.deep_supervision_outputs.{i}); forward runs on the HIP engine."""
from copy import deepcopy

import numpy as np
import torch
from torch import nn

from .initialization import InitWeights_He
from .neural_network import SegmentationNetwork


def get_default_network_config(dim=3, dropout_p=None, nonlin="LeakyReLU", norm_type="bn"):
    """Same dict the reference builds (generic_modular_UNet.py:31-78) for dim=3."""
    if dim != 3:
        raise NotImplementedError("3D only")
    props = {'conv_op': nn.Conv3d, 'dropout_op': nn.Dropout3d}
    if norm_type == "bn":
        props['norm_op'] = nn.BatchNorm3d
    elif norm_type == "in":
        props['norm_op'] = nn.InstanceNorm3d
    else:
        raise NotImplementedError
    props['conv_op_kwargs'] = {'stride': 1, 'dilation': 1, 'bias': False}   # generic_modular_UNet.py:67
    props['norm_op_kwargs'] = {'eps': 1e-5, 'affine': True}
    props['dropout_op'] = None if dropout_p is None else props['dropout_op']
    props['dropout_op_kwargs'] = {'p': 0 if dropout_p is None else dropout_p, 'inplace': True}
    if nonlin == "LeakyReLU":
        props['nonlin'] = nn.LeakyReLU
        props['nonlin_kwargs'] = {'negative_slope': 1e-2, 'inplace': True}
    elif nonlin == "ReLU":
        props['nonlin'] = nn.ReLU
        props['nonlin_kwargs'] = {'inplace': True}
    else:
        raise ValueError
    return props


def _check_props(props):
    if props['conv_op'] is not nn.Conv3d or props['norm_op'] is not nn.InstanceNorm3d or props['nonlin'] is not nn.LeakyReLU:
        raise NotImplementedError("HIP engine: Conv3d + InstanceNorm3d + LeakyReLU only")
    if props['dropout_op_kwargs']['p'] != 0:
        raise NotImplementedError("dropout unsupported")


def _conv_kwargs(props):
    kw = dict(props['conv_op_kwargs'])
    kw.pop('stride', None)
    return kw


class ConvDropoutNormReLU(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size, network_props):
        super().__init__()
        network_props = deepcopy(network_props)
        _check_props(network_props)
        self.kernel_size = list(kernel_size)
        self.conv = nn.Conv3d(input_channels, output_channels, kernel_size, padding=[(i - 1) // 2 for i in kernel_size],
                              **network_props['conv_op_kwargs'])
        self.do = nn.Identity()
        self.norm = nn.InstanceNorm3d(output_channels, **network_props['norm_op_kwargs'])
        self.nonlin = nn.LeakyReLU(**network_props['nonlin_kwargs'])
        # the reference also registers the same modules as `all` (conv_blocks.py:81), which duplicates the
        # state_dict entries (all.0.weight, all.2.{weight,bias} alias conv/norm) — kept for checkpoint parity
        self.all = nn.Sequential(self.conv, self.do, self.norm, self.nonlin)


class StackedConvLayers(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size, network_props, num_convs, first_stride=None):
        super().__init__()
        props = deepcopy(network_props)
        first = deepcopy(props)
        if first_stride is not None:
            first['conv_op_kwargs']['stride'] = first_stride
        self.convs = nn.Sequential(ConvDropoutNormReLU(input_channels, output_channels, kernel_size, first),
                                   *[ConvDropoutNormReLU(output_channels, output_channels, kernel_size, props)
                                     for _ in range(num_convs - 1)])


class BasicResidualBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, props, stride=None, use_avgpool_in_skip=False):
        super().__init__()
        if use_avgpool_in_skip:
            raise NotImplementedError
        props = deepcopy(props)
        _check_props(props)
        kw = _conv_kwargs(props)
        kernel_size = list(kernel_size) if isinstance(kernel_size, (list, tuple)) else [kernel_size] * 3
        if stride is None:
            stride = [1, 1, 1]
        elif isinstance(stride, (tuple, list)):
            stride = [i if i is not None else 1 for i in stride]
        else:
            stride = [stride] * 3
        self.stride, self.kernel_size, self.props = stride, kernel_size, props
        self.in_planes, self.out_planes = in_planes, out_planes
        pad = [(i - 1) // 2 for i in kernel_size]
        self.conv1 = nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad, stride=stride, **kw)
        self.norm1 = nn.InstanceNorm3d(out_planes, **props['norm_op_kwargs'])
        self.nonlin1 = nn.LeakyReLU(**props['nonlin_kwargs'])
        self.conv2 = nn.Conv3d(out_planes, out_planes, kernel_size=kernel_size, padding=pad, stride=1, **kw)
        self.norm2 = nn.InstanceNorm3d(out_planes, **props['norm_op_kwargs'])
        self.nonlin2 = nn.LeakyReLU(**props['nonlin_kwargs'])
        if any(i != 1 for i in stride) or in_planes != out_planes:
            self.downsample_skip = nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=1, padding=0, stride=stride, bias=False),
                                                 nn.InstanceNorm3d(out_planes, **props['norm_op_kwargs']))
        else:
            self.downsample_skip = None


class ResidualLayer(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size, network_props, num_blocks, first_stride=None,
                 block=BasicResidualBlock, block_kwargs=None):
        super().__init__()
        block_kwargs = {} if block_kwargs is None else block_kwargs
        props = deepcopy(network_props)
        self.convs = nn.Sequential(block(input_channels, output_channels, kernel_size, props, first_stride, **block_kwargs),
                                   *[block(output_channels, output_channels, kernel_size, props, **block_kwargs)
                                     for _ in range(num_blocks - 1)])
        self.output_channels = output_channels


class ResidualUNetEncoder(nn.Module):
    def __init__(self, input_channels, base_num_features, num_blocks_per_stage, feat_map_mul_on_downscale,
                 pool_op_kernel_sizes, conv_kernel_sizes, props, default_return_skips=True, max_num_features=480,
                 block=BasicResidualBlock, block_kwargs=None):
        super().__init__()
        _check_props(props)
        self.default_return_skips = default_return_skips
        self.props = props
        assert len(pool_op_kernel_sizes) == len(conv_kernel_sizes)
        num_stages = len(conv_kernel_sizes)
        if not isinstance(num_blocks_per_stage, (list, tuple)):
            num_blocks_per_stage = [num_blocks_per_stage] * num_stages
        assert len(num_blocks_per_stage) == num_stages
        self.num_blocks_per_stage = num_blocks_per_stage
        self.initial_conv = nn.Conv3d(input_channels, base_num_features, 3, padding=1, **props['conv_op_kwargs'])
        self.initial_norm = nn.InstanceNorm3d(base_num_features, **props['norm_op_kwargs'])
        self.initial_nonlin = nn.LeakyReLU(**props['nonlin_kwargs'])
        self.stage_output_features, self.stage_pool_kernel_size, self.stage_conv_op_kernel_size = [], [], []
        stages = []
        cin = base_num_features
        for stage in range(num_stages):
            cout = min(base_num_features * feat_map_mul_on_downscale ** stage, max_num_features)
            layer = ResidualLayer(cin, cout, conv_kernel_sizes[stage], props, num_blocks_per_stage[stage],
                                  pool_op_kernel_sizes[stage], block, block_kwargs)
            stages.append(layer)
            self.stage_output_features.append(layer.output_channels)
            self.stage_conv_op_kernel_size.append(conv_kernel_sizes[stage])
            self.stage_pool_kernel_size.append(pool_op_kernel_sizes[stage])
            cin = layer.output_channels
        self.output_features = cin
        self.stages = nn.ModuleList(stages)


class PlainConvUNetDecoder(nn.Module):
    def __init__(self, previous, num_classes, num_blocks_per_stage=None, network_props=None, deep_supervision=False,
                 upscale_logits=False):
        super().__init__()
        if upscale_logits:
            raise NotImplementedError
        self.num_classes = num_classes
        self.deep_supervision = deep_supervision
        self.props = previous.props if network_props is None else network_props
        _check_props(self.props)
        feats, pools, kernels = previous.stage_output_features, previous.stage_pool_kernel_size, previous.stage_conv_op_kernel_size
        if num_blocks_per_stage is None:
            num_blocks_per_stage = previous.num_blocks_per_stage[:-1][::-1]
        assert len(num_blocks_per_stage) == len(previous.num_blocks_per_stage) - 1
        self.stage_pool_kernel_size, self.stage_output_features, self.stage_conv_op_kernel_size = pools, feats, kernels
        num_stages = len(previous.stages) - 1
        tus, stages, heads = [], [], []
        for i, s in enumerate(np.arange(num_stages)[::-1]):
            if num_blocks_per_stage[i] != 1:
                raise NotImplementedError("HIP engine: one conv block per decoder stage (MultiTalent_meets_resenc.py:81)")
            tus.append(nn.ConvTranspose3d(feats[s + 1], feats[s], pools[s + 1], pools[s + 1], bias=False))
            stages.append(StackedConvLayers(2 * feats[s], feats[s], kernels[s], self.props, num_blocks_per_stage[i]))
            if deep_supervision and s != 0:
                heads.append(nn.Conv3d(feats[s], num_classes, 1, 1, 0, 1, 1, bias=True))
        heads.append(nn.Conv3d(feats[0], num_classes, 1, 1, 0, 1, 1, bias=True))
        self.tus = nn.ModuleList(tus)
        self.stages = nn.ModuleList(stages)
        self.deep_supervision_outputs = nn.ModuleList(heads)


class FabiansUNet(SegmentationNetwork):
    use_this_for_2D_configuration = 1244233721.0
    use_this_for_3D_configuration = 1230348801.0
    default_blocks_per_stage_encoder = (1, 2, 3, 4, 4, 4, 4, 4, 4, 4, 4)
    default_blocks_per_stage_decoder = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1)
    default_min_batch_size = 2

    def __init__(self, input_channels, base_num_features, num_blocks_per_stage_encoder, feat_map_mul_on_downscale,
                 pool_op_kernel_sizes, conv_kernel_sizes, props, num_classes, num_blocks_per_stage_decoder,
                 deep_supervision=False, upscale_logits=False, max_features=512, initializer=None,
                 block=BasicResidualBlock, props_decoder=None, block_kwargs=None):
        super().__init__()
        self.do_ds = deep_supervision
        self.conv_op = props['conv_op']
        self.num_classes = num_classes
        self.encoder = ResidualUNetEncoder(input_channels, base_num_features, num_blocks_per_stage_encoder,
                                           feat_map_mul_on_downscale, pool_op_kernel_sizes, conv_kernel_sizes, props,
                                           default_return_skips=True, max_num_features=max_features, block=block,
                                           block_kwargs=block_kwargs)
        props['dropout_op_kwargs']['p'] = 0
        if props_decoder is None:
            props_decoder = props
        self.decoder = PlainConvUNetDecoder(self.encoder, num_classes, num_blocks_per_stage_decoder, props_decoder,
                                            deep_supervision, upscale_logits)
        self.pool_op_kernel_sizes = pool_op_kernel_sizes
        self.conv_kernel_sizes = conv_kernel_sizes
        self.input_shape_must_be_divisible_by = np.prod(pool_op_kernel_sizes, 0, dtype=np.int64)
        if initializer is not None:
            self.apply(initializer)
        self._engine = None

    def engine(self):
        if self._engine is None:
            from ..engine import build_resenc_unet_engine
            self._engine = build_resenc_unet_engine(self)
        return self._engine

    def forward(self, x):
        ds = bool(self.decoder.deep_supervision)
        outs = self.engine().apply(x, all_heads=ds)
        return list(outs) if ds else outs[0]
