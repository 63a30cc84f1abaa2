"""Synthetic, device-resident training batches in the style of the reference's dummyLoad benchmarking trainer
(nnUNet_variants/benchmarking/nnUNetTrainerV2_dummyLoad.py:26-64): CT-like intensities drawn from the Task100 plan's
global statistics, clipped and z-scored like the CT preprocessing (preprocessing.py:275-285), and blocky integer label
maps stored as float32 with a nearest-neighbour deep-supervision pyramid (downsampling.py:70-104)."""
import numpy as np
import torch

from .plans import TASK100_CT_STATS


def ds_scales(pool_op_kernel_sizes, skip_first=False):
    """deep_supervision_scales (nnUNetTrainerV2.py:107-108; resenc variant MultiTalent_meets_resenc.py:107-116)."""
    pools = np.vstack(pool_op_kernel_sizes[1:] if skip_first else pool_op_kernel_sizes)
    return [[1, 1, 1]] + [list(i) for i in 1 / np.cumprod(pools, axis=0)][:-1]


def synthetic_ct(B, patch, seed, device):
    g = torch.Generator(device='cpu').manual_seed(seed)
    s = TASK100_CT_STATS
    hu = torch.randn((B, 1) + tuple(patch), generator=g) * s['sd'] + s['mean']
    hu = hu.clamp_(s['percentile_00_5'], s['percentile_99_5'])
    return ((hu - s['mean']) / s['sd']).float().to(device)


def synthetic_targets(B, patch, scales, label_sets, seed, device, block=8):
    """label_sets[b]: label values present in sample b.  Returns list of [B,1,...] float32 maps, highest res first."""
    g = torch.Generator(device='cpu').manual_seed(seed + 7)
    coarse_shape = tuple(max(p // block, 1) for p in patch)
    maps = []
    for b in range(B):
        labs = torch.tensor([0] + list(label_sets[b]), dtype=torch.float32)
        idx = torch.randint(0, len(labs), coarse_shape, generator=g)
        keep = torch.rand(coarse_shape, generator=g) < 0.35      # mostly background, like real CT crops
        maps.append(torch.where(keep, labs[idx], torch.zeros(())))
    coarse = torch.stack(maps, 0)[:, None]
    full = torch.nn.functional.interpolate(coarse, size=tuple(patch), mode='nearest')
    out = []
    for sc in scales:
        size = tuple(int(np.round(p * f)) for p, f in zip(patch, sc))
        out.append(torch.nn.functional.interpolate(full, size=size, mode='nearest').contiguous().to(device))
    return out


class SyntheticBatchGenerator:
    """Fixed device-resident batch, reference dummyLoad style (nnUNetTrainerV2_dummyLoad.py:26-64)."""

    def __init__(self, trainer, seed=1234):
        from .dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
        dev = torch.device('cuda', torch.cuda.current_device())
        B, patch = int(trainer.batch_size), tuple(int(i) for i in trainer.patch_size)
        rank = getattr(trainer, 'local_rank', 0)
        scales = trainer.deep_supervision_scales
        self.data = synthetic_ct(B, patch, seed + rank, dev)
        names = list(MultiTalent_valid_regions.keys())
        if getattr(trainer, 'regions', None) is not None:
            valid = [MultiTalent_valid_regions[names[(rank * B + b) % len(names)]] for b in range(B)]
            label_sets = [sorted({l for r in v for l in MultiTalent_regions[r]}) for v in valid]
        else:
            valid = None
            label_sets = [list(range(1, trainer.num_classes))] * B
        self.target = synthetic_targets(B, patch, scales, label_sets, seed + rank, dev)
        self.properties = [{'valid_regions': v} for v in valid] if valid is not None else [{} for _ in range(B)]

    def __iter__(self):
        return self

    def __next__(self):
        return {'data': self.data, 'target': self.target, 'properties': self.properties, 'keys': None}
