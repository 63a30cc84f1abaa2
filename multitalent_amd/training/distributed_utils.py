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


def active():
    return _active()


def world_size():
    return dist.get_world_size() if _active() else 1


def all_reduce_sum(t):
    """Plain (no autograd) in-place SUM over ranks: the fused loss step's exchange of the Dice statistics."""
    return _all_reduce_sum(t)


def _all_reduce_sum(t):
    """in-place SUM over ranks of a small tensor.  RCCL ('nccl') reduces device tensors directly; under gloo (CPU tests, or two
    ranks sharing one GPU in the single-GPU test box) a device tensor travels through the host — transport only, a few KB."""
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def assert_same_on_all_ranks(values, what):
    """One MAX all-reduce of (v, -v): raises on EVERY rank when the ranks disagree on `values` (a short list of numbers).  Used once per
    training-step object to pin decisions that change the NUMBER of collectives a rank issues (fused loss step or autograd form): a
    disagreement would otherwise show up as a deadlock."""
    if not active():
        return
    v = torch.tensor([float(x) for x in values], dtype=torch.float64)
    t = torch.cat([v, -v])
    if dist.get_backend() != 'gloo':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu()
    hi, lo = t[:len(v)], -t[len(v):]
    if not torch.equal(hi, lo):
        raise RuntimeError("%s differs between the ranks: max %s, min %s (this rank: %s)" % (what, hi.tolist(), lo.tolist(), v.tolist()))


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _all_reduce_sum(x.contiguous().clone())

    @staticmethod
    def backward(ctx, gy):
        return _all_reduce_sum(gy.contiguous().clone())


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


def gather_over_ranks(x):
    """[...] -> [W, ...] stacked over ranks (no gradient): awesome_allgather_function's forward as run_online_evaluation uses it
    (MultiTalent_Trainer_DDP.py:399-401).  World size 1 (or no process group): x[None]."""
    if not _active():
        return x[None]
    x = x.contiguous()
    if x.is_cuda and dist.get_backend() == 'gloo':
        h = x.cpu()
        out = [torch.zeros_like(h) for _ in range(dist.get_world_size())]
        dist.all_gather(out, h)
        return torch.stack(out, 0).to(x.device)
    out = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x)
    return torch.stack(out, 0)
