"""CPU experiment (oracle only, no GPU): how far is the gradient of the residual-encoder network from the exact (fp64) gradient when
the convolutions' matrix operands — and optionally the stored activations — are rounded to bf16 (8 mantissa bits) or fp16 (11 bits,
the reference's autocast type)?  Straight-through rounding inside the torch oracle; everything else fp32.
usage: python tools/emulate_precision_cpu.py D H W"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from oracle import reference_ops as R
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                                        MultiTalent_valid_regions)
    from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
    patch = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 128)
    torch.manual_seed(4321)
    torch.set_num_threads(os.cpu_count())
    net = bench.build_network('resenc')
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    dev = torch.device('cpu')
    B = 1
    valid = [MultiTalent_valid_regions['Task064_KiTS_labelsFixed']] * B
    label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
    x = synthetic_ct(B, patch, 79, dev)
    tg = synthetic_targets(B, patch, ds_scales(bench.RESENC_POOLS, skip_first=True), label_sets, 79, dev)
    w = R.ds_loss_weights(len(bench.RESENC_POOLS))
    real_conv = F.conv3d
    mode = {'op': None, 'store': None}

    def rnd(t, dt):
        return t + (t.to(dt).to(t.dtype) - t).detach()            # straight-through

    def conv(xx, ww, bb=None, stride=1, padding=0, **kw):
        if mode['op'] is not None and ww.shape[1] >= 16 and ww.shape[2:] != (1, 1, 1):
            xx, ww = rnd(xx, mode['op']), rnd(ww, mode['op'])
        y = real_conv(xx, ww, bb, stride=stride, padding=padding, **kw)
        if mode['store'] is not None and ww.shape[2:] != (1, 1, 1):
            y = rnd(y, mode['store'])
        return y

    def run(dt):
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
        out = R.fabians_unet_forward(sd, x.to(dt), bench.RESENC_POOLS, bench.RESENC_KERNELS, bench.RESENC_BLOCKS)
        rl = R.multitalent_loss(list(out), tg, valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping, w)
        rl[0].backward()
        g = torch.cat([(v.grad if v.grad is not None else torch.zeros_like(v)).reshape(-1) for v in sd.values()]).double()
        return g, [o.detach().double() for o in out]

    g64, o64 = run(torch.float64)
    R.F.conv3d = conv
    try:
        for tag, op, st in (('fp32', None, None), ('bf16 operands', torch.bfloat16, None), ('fp16 operands', torch.float16, None),
                            ('bf16 operands + bf16 stored y', torch.bfloat16, torch.bfloat16),
                            ('fp16 operands + fp16 stored y', torch.float16, torch.float16),
                            ('fp16 operands + bf16 stored y', torch.float16, torch.bfloat16)):
            mode['op'], mode['store'] = op, st
            g, o = run(torch.float32)
            cos = float((g * g64).sum() / (g.norm() * g64.norm()))
            lrel = ['%.4f' % float((a - b).norm() / b.norm()) for a, b in zip(o, o64)]
            print("%-34s gradient cos %.5f rel.L2 %.3e   logits rel.L2 %s" % (tag, cos, float((g - g64).norm() / g64.norm()), lrel), flush=True)
    finally:
        R.F.conv3d = real_conv


if __name__ == '__main__':
    main()
