"""Fused segmentation losses on the HIP path.

Each deep-supervision level is ONE kernel pass producing a tiny statistics tensor stats[B, C, 4] =
(bce/ce sum, tp, fp, fn); the scalar loss is then combined from these statistics with ordinary torch
ops on [B, C]-sized tensors (host-side glue, a few hundred floats), and the backward kernel is the exact
vector-Jacobian product of the statistics w.r.t. the logits — dlogits is written once, NDHWC, and
handed to the network engine without a copy.

Reference semantics:
  * MultiTalent: MultiTalent_Trainer_DDP.py:544-623 (== MultiTalent_meets_resenc.py:713-798) — BCE is the
    MEAN over voxels, SUMMED over valid regions and samples; Dice has NO smooth term, clamp 1e-7, summed;
    with batch_dice the statistics are all_gathered and summed over RANKS only (distributed.py:63).
  * softmax Dice+CE: dice_loss.py:100-195,488-545, crossentropy.py:4-11, deep_supervision.py:19-43,
    DDP variant nnUNetTrainerV2_DDP.py:249-282.
"""
import numpy as np
import torch
from torch import nn

from .. import distributed_utils
from ... import ops
from ...dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                       region_label_lut, valid_mask)
from ...ops import Act

_ws_cache = {}
_const_cache = {}


def _const(values, dev):
    """Small constant vector on the device, uploaded once (a per-step torch.tensor(list, device=...) is a blocking copy)."""
    key = (tuple(float(v) for v in values), str(dev))
    t = _const_cache.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=torch.float32, device=dev)
        _const_cache[key] = t
    return t


def _workspace(dev, nbytes):
    n = (int(nbytes) + 3) // 4 + 16
    t = _ws_cache.get(dev)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 18), dtype=torch.float32, device=dev)
        _ws_cache[dev] = t
    return t


def _as_ndhwc(logits):
    """logits: logical [B,C,D,H,W].  Returns a dense NDHWC tensor sharing memory when the strides are channels-last
    (which is what the engine returns)."""
    if not logits.is_cuda:
        raise RuntimeError("fused losses run on the HIP device only (no CPU fallback)")
    return logits.permute(0, 2, 3, 4, 1).contiguous()


def _target_flat(target):
    """[B,1,D,H,W] (or [B,D,H,W]) float label map -> contiguous float32 [B, V]."""
    t = target
    if t.dim() == 5:
        t = t[:, 0]
    return t.reshape(t.shape[0], -1).contiguous().float()


class _MultiTalentStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, valid, lut):
        x = _as_ndhwc(logits)
        a = Act(x)
        stats = torch.empty((a.N, a.C, 4), dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, ops.loss_workspace(a.N, a.V, a.C))
        ops.multitalent_loss_fwd(a, target, valid, lut, stats, ws)
        ctx.save_for_backward(x, target, valid, lut)
        return stats

    @staticmethod
    def backward(ctx, gstats):
        x, target, valid, lut = ctx.saved_tensors
        d = torch.empty_like(x)
        ops.multitalent_loss_bwd(Act(x), target, valid, lut, gstats.contiguous().float(), Act(d))
        return d.permute(0, 4, 1, 2, 3), None, None, None


class _SoftmaxStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        x = _as_ndhwc(logits)
        a = Act(x)
        stats = torch.empty((a.N, a.C, 4), dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, ops.loss_workspace(a.N, a.V, a.C))
        ops.softmax_dice_ce_fwd(a, target, stats, ws)
        ctx.save_for_backward(x, target)
        return stats

    @staticmethod
    def backward(ctx, gstats):
        x, target = ctx.saved_tensors
        d = torch.empty_like(x)
        ops.softmax_dice_ce_bwd(Act(x), target, gstats.contiguous().float(), Act(d))
        return d.permute(0, 4, 1, 2, 3), None


class MultiTalentLoss(nn.Module):
    """Callable with the signature of MultiTalent_trainer_ddp.compute_loss(output, target, valid_regions)
    -> (total_loss, total_ce, total_dc)."""

    def __init__(self, ds_loss_weights, batch_dice=True, regions=None, region_idx=None):
        super().__init__()
        self.ds_loss_weights = [float(w) for w in ds_loss_weights]
        self.batch_dice = batch_dice
        self.regions = MultiTalent_regions if regions is None else regions
        self.region_idx = MultiTalent_region_output_idx_mapping if region_idx is None else region_idx
        lut = [0] * len(self.regions)
        for name, labels in self.regions.items():
            m = 0
            for l in labels:
                m |= (1 << int(l))
            lut[self.region_idx[name]] = m
        self._lut_host = np.array(lut, dtype=np.uint64).view(np.int64)
        self._lut = {}
        self._valid_cache = {}

    def _masks(self, valid_regions, dev):
        if dev not in self._lut:
            self._lut[dev] = torch.from_numpy(self._lut_host.copy()).to(dev)
        v = []
        for names in valid_regions:
            m = 0
            for r in names:
                m |= (1 << self.region_idx[r])
            v.append(m)
        key = (tuple(v), str(dev))                    # the per-sample masks repeat (one per dataset): upload each combination once
        valid = self._valid_cache.get(key)
        if valid is None:
            if len(self._valid_cache) > 4096:
                self._valid_cache.clear()
            valid = torch.from_numpy(np.array(v, dtype=np.uint64).view(np.int64)).to(dev)
            self._valid_cache[key] = valid
        return valid, self._lut[dev]

    def forward(self, output, target, valid_regions):
        dev = output[0].device
        valid, lut = self._masks(valid_regions, dev)
        total_loss = total_ce = total_dc = None
        stats_all, nvox = [], []
        for i in range(len(output)):
            t = _target_flat(target[i])
            stats_all.append(_MultiTalentStats.apply(output[i], t, valid, lut))   # [B, C, 4]
            nvox.append(t.shape[1])
        stats_all = torch.stack(stats_all, 0)                                  # [L, B, C, 4]
        dice_stats = stats_all[..., 1:]
        if self.batch_dice:
            dice_stats = distributed_utils.sum_over_ranks(dice_stats)
        # all levels at once ([L]-vectors): the per-level Python loop of the reference costs ~40 tiny launches per level
        w = _const(self.ds_loss_weights[:len(output)], dev)
        inv_nvox = _const([1.0 / n for n in nvox], dev)
        ce = stats_all[..., 0].sum((1, 2)) * inv_nvox                          # BCE: mean over voxels, summed over (b, region)
        tp, fp, fn = dice_stats[..., 0], dice_stats[..., 1], dice_stats[..., 2]
        dc = (2 * tp / torch.clamp(2 * tp + fp + fn, min=1e-7)).sum((1, 2))
        total_ce = (w * ce).sum()
        total_dc = (w * dc).sum()
        return total_ce - total_dc, total_ce, total_dc


    def fused_step(self, outs, target, valid_regions):
        """Value AND dLoss/dlogits of `forward` without autograd (the training step's path): per level one statistics pass, ONE
        combination launch (mt_loss_combine: value + gradient of the [L, B, C] arithmetic the reference spells as ~40 autograd
        operations), per level one backward pass.  outs = the engine's NDHWC logits of every level.
        Returns ((total, ce, dc), [dlogits NDHWC]), or None where only the autograd form applies."""
        dev = outs[0].device
        if isinstance(valid_regions, torch.Tensor):        # the per-sample region masks as a device tensor (static_args: the captured step)
            valid, lut = valid_regions, self._masks([], dev)[1]
        else:
            valid, lut = self._masks(valid_regions, dev)
        acts = [Act(o) for o in outs]
        L, B, Cn = len(acts), acts[0].N, acts[0].C
        if any((a.N, a.C) != (B, Cn) for a in acts):
            return None                      # (levels of different (B, C): the autograd form)
        st = torch.empty((L, B, Cn, 4), dtype=torch.float32, device=dev)
        tg = []
        for i, a in enumerate(acts):
            t = _target_flat(target[i])
            tg.append(t)
            ops.multitalent_loss_fwd(a, t, valid, lut, st[i], _workspace(dev, ops.loss_workspace(a.N, a.V, a.C)))
        dice, stride, gscale = st.view(-1)[1:], 4, 1.0
        if self.batch_dice and distributed_utils.active():
            dice, stride = distributed_utils.all_reduce_sum(st[..., 1:].contiguous()), 3
            gscale = float(distributed_utils.world_size())
        w = _const(self.ds_loss_weights[:L], dev)
        ce_coef = _const([self.ds_loss_weights[i] / tg[i].shape[1] for i in range(L)], dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        g = torch.empty_like(st)
        ops.loss_combine(st, dice, stride, ce_coef, w, ops.LOSS_CE_ALL_CHANNELS, 0, 0.0, 0.0, 0.0, 1e-7, out3, g, dice_grad_scale=gscale)
        dl = []
        for i, a in enumerate(acts):
            d = torch.empty_like(outs[i])
            ops.multitalent_loss_bwd(a, tg[i], valid, lut, g[i], Act(d))
            dl.append(d)
        return (out3[0], out3[1], out3[2]), dl


    def static_args(self, target, valid_regions):
        """For the HIP-graph form of the training step (hot_loop.FusedTrainStep): (device tensors that change from step to step, a function
        that rebuilds fused_step's arguments from static copies of them, a hashable key of everything else).  The valid regions travel as
        their mask tensor."""
        if distributed_utils.active():
            return None
        tg = list(target)
        dev = tg[0].device
        valid = self._masks(valid_regions, dev)[0]
        return tg + [valid], (lambda ts: (ts[:-1], ts[-1])), ('mt', len(tg), self.batch_dice, tuple(self.ds_loss_weights))


class DC_and_CE_DS_loss(nn.Module):
    """MultipleOutputLoss2(DC_and_CE_loss({'batch_dice', 'smooth': 1e-5, 'do_bg': False}, {}), weights)
    (nnUNetTrainer.py:108, nnUNetTrainerV2.py:78-90) fused per level.  `ddp=True` reproduces
    nnUNetTrainerV2_DDP.compute_loss (no +1e-8 in the denominator, batch-dice statistics gathered over ranks)."""

    def __init__(self, ds_loss_weights, batch_dice=False, smooth=1e-5, do_bg=False, ddp=False):
        super().__init__()
        self.ds_loss_weights = [float(w) for w in ds_loss_weights]
        self.batch_dice, self.smooth, self.do_bg, self.ddp = batch_dice, smooth, do_bg, ddp

    def level_loss(self, logits, target):
        t = _target_flat(target)
        stats = _SoftmaxStats.apply(logits, t)                                 # [B, C, 4]
        B, V = t.shape
        ce = stats[:, 0, 0].sum() / (B * V)                                    # CrossEntropyLoss mean over all voxels
        tp, fp, fn = stats[:, :, 1], stats[:, :, 2], stats[:, :, 3]
        if not self.do_bg:
            tp, fp, fn = tp[:, 1:], fp[:, 1:], fn[:, 1:]
        if self.ddp:
            # nnUNetTrainerV2_DDP.py:262-279
            nominator = 2 * tp
            denominator = 2 * tp + fp + fn
            if self.batch_dice:   # gathered [W,B,C-1] summed over the rank axis only -> still [B, C-1]
                nd = distributed_utils.sum_over_ranks(torch.stack((nominator, denominator), 0))
                nominator, denominator = nd[0], nd[1]
            dice_loss = (-(nominator + self.smooth) / (denominator + self.smooth)).mean()
            return ce + dice_loss
        if self.batch_dice:
            tp, fp, fn = tp.sum(0), fp.sum(0), fn.sum(0)
        dc = (2 * tp + self.smooth) / (2 * tp + fp + fn + self.smooth + 1e-8)  # dice_loss.py:180-183
        return ce - dc.mean()

    def forward(self, output, target):
        active = [i for i in range(len(output)) if i == 0 or self.ds_loss_weights[i] != 0]     # deep_supervision.py:37-42
        if len({tuple(output[i].shape[:2]) for i in active}) != 1:
            l = self.ds_loss_weights[0] * self.level_loss(output[0], target[0])
            for i in active[1:]:
                l = l + self.ds_loss_weights[i] * self.level_loss(output[i], target[i])
            return l
        # same (B, C) at every level: combine all levels with [L]-vector math (the per-level loop is ~40 tiny launches each)
        dev = output[0].device
        stats, bv = [], []
        for i in active:
            t = _target_flat(target[i])
            stats.append(_SoftmaxStats.apply(output[i], t))
            bv.append(t.shape[0] * t.shape[1])
        st = torch.stack(stats, 0)                                             # [L, B, C, 4]
        w = _const([self.ds_loss_weights[i] for i in active], dev)
        inv_bv = _const([1.0 / n for n in bv], dev)
        ce = st[:, :, 0, 0].sum(1) * inv_bv                                    # CrossEntropyLoss mean over all voxels
        tp, fp, fn = st[..., 1], st[..., 2], st[..., 3]
        if not self.do_bg:
            tp, fp, fn = tp[:, :, 1:], fp[:, :, 1:], fn[:, :, 1:]
        if self.ddp:
            nominator = 2 * tp
            denominator = 2 * tp + fp + fn
            if self.batch_dice:
                nd = distributed_utils.sum_over_ranks(torch.stack((nominator, denominator), 0))
                nominator, denominator = nd[0], nd[1]
            dice_loss = (-(nominator + self.smooth) / (denominator + self.smooth)).mean((1, 2))
            return (w * (ce + dice_loss)).sum()
        if self.batch_dice:
            tp, fp, fn = tp.sum(1), fp.sum(1), fn.sum(1)
            dc = ((2 * tp + self.smooth) / (2 * tp + fp + fn + self.smooth + 1e-8)).mean(1)
        else:
            dc = ((2 * tp + self.smooth) / (2 * tp + fp + fn + self.smooth + 1e-8)).mean((1, 2))
        return (w * (ce - dc)).sum()

    def fused_step(self, outs, target):
        """As MultiTalentLoss.fused_step.  Returns None where the autograd form has to run: levels of different (B, C), or the DDP
        variant's batch Dice over ranks (numerator / denominator gathered, nnUNetTrainerV2_DDP.py:267-270)."""
        active = [i for i in range(len(outs)) if i == 0 or self.ds_loss_weights[i] != 0]
        acts = {i: Act(outs[i]) for i in active}
        if len({(a.N, a.C) for a in acts.values()}) != 1 or (self.ddp and self.batch_dice and distributed_utils.active()):
            return None
        dev = outs[0].device
        a0 = acts[active[0]]
        L, B, Cn = len(active), a0.N, a0.C
        st = torch.empty((L, B, Cn, 4), dtype=torch.float32, device=dev)
        tg = {}
        for k, i in enumerate(active):
            t = _target_flat(target[i])
            tg[i] = t
            ops.softmax_dice_ce_fwd(acts[i], t, st[k], _workspace(dev, ops.loss_workspace(B, acts[i].V, Cn)))
        c0 = 0 if self.do_bg else 1
        over_b = self.batch_dice and not self.ddp                  # nnUNetTrainerV2_DDP.compute_loss keeps [B, C-1] entries
        count = (Cn - c0) * (1 if over_b else B)
        ce_coef = _const([self.ds_loss_weights[i] / (B * tg[i].shape[1]) for i in active], dev)
        dice_coef = _const([self.ds_loss_weights[i] / count for i in active], dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        g = torch.empty_like(st)
        ops.loss_combine(st, st.view(-1)[1:], 4, ce_coef, dice_coef, ops.LOSS_DICE_OVER_BATCH if over_b else 0, c0, self.smooth,
                         self.smooth, 0.0 if self.ddp else 1e-8, -3.0e38, out3, g)
        dl = [None] * len(outs)
        for k, i in enumerate(active):
            d = torch.empty_like(outs[i])
            ops.softmax_dice_ce_bwd(acts[i], tg[i], g[k], Act(d))
            dl[i] = d
        return out3[0], dl

    def static_args(self, target):
        """As MultiTalentLoss.static_args: the targets are the only per-step tensors."""
        if self.ddp and distributed_utils.active():
            return None
        tg = list(target)
        return tg, (lambda ts: (ts,)), ('dcce', len(tg), self.batch_dice, self.do_bg, self.ddp, float(self.smooth), tuple(self.ds_loss_weights))
