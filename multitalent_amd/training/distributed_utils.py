"""Cross-rank exchange of the Dice statistics.

Reference: awesome_allgather_function (utilities/distributed.py:28-73): forward all_gather + stack -> [W,B,C];
the callers immediately `.sum(0)` over the rank axis (MultiTalent_Trainer_DDP.py:598-604,
nnUNetTrainerV2_DDP.py:267-270); backward all_reduce(SUM) of the gathered gradient and select the own slice.
gather-then-sum == all_reduce(SUM), and its backward (sum over ranks of dL_r/dy) is again one all_reduce, so
the 3 x 5 tiny collectives per direction of the reference collapse into ONE RCCL all_reduce per direction
(the caller stacks tp/fp/fn of all deep-supervision levels into one tensor)."""
import torch
import torch.distributed as dist


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x.contiguous().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, gy):
        g = gy.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


def sum_over_ranks(x):
    """y = sum_r x_r on every rank, differentiable w.r.t. the local x exactly like
    awesome_allgather_function(x).sum(0)."""
    if not _active():
        return x
    return _AllReduceSum.apply(x)


def gather_sum_over_ranks(tp, fp, fn):
    """[B,C] x3 -> [1,B,C] x3 summed over ranks (one collective for the three tensors)."""
    s = sum_over_ranks(torch.stack((tp, fp, fn), 0))
    return s[0][None], s[1][None], s[2][None]
