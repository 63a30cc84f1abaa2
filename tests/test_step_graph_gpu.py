"""The HIP-graph form of the training step (hot_loop.FusedTrainStep, MT_STEP_GRAPH): the captured and replayed step must be the eager
step — same launches, same order per stream — so the parameters after N iterations are BIT-identical, with changing batches, a learning
rate that changes on the way (a new capture) and both losses.  Reference: run_iteration, MultiTalent_Trainer_DDP.py:324-370."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _net(nc):
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    from multitalent_amd.network_architecture.initialization import InitWeights_He
    pools, kernels = [[2, 2, 2], [2, 2, 2], [1, 2, 2]], [[3, 3, 3]] * 4
    torch.manual_seed(3)
    return Generic_UNet(1, 8, nc, 3, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                        {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                        lambda x: x, InitWeights_He(1e-2), pools, kernels, False, True, True)


@pytest.mark.parametrize("kind,precision", [('softmax', 'fp32'), ('multitalent', 'fp32'), ('multitalent', 'bf16')])
def test_graph_replay_equals_eager_step(dev, kind, precision):
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss, MultiTalentLoss
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_regions, MultiTalent_valid_regions
    names = list(MultiTalent_valid_regions.keys())
    nc = 3 if kind == 'softmax' else 47
    B, patch = 2, (8, 32, 32)
    shapes = [patch, (4, 16, 16), (2, 8, 8)]
    w = [4 / 7, 2 / 7, 1 / 7]
    g = torch.Generator().manual_seed(11)
    batches = []
    for it in range(10):
        x = torch.randn((B, 1) + patch, generator=g).to(dev)
        if kind == 'softmax':
            tg = [torch.randint(0, nc, (B, 1) + s, generator=g).float().to(dev) for s in shapes]
            batches.append((x, (tg,)))
        else:
            valid = [MultiTalent_valid_regions[names[(it + b) % len(names)]] for b in range(B)]
            tg = [torch.randint(0, 30, (B, 1) + s, generator=g).float().to(dev) for s in shapes]
            batches.append((x, (tg, valid)))
    results = []
    for use_graph in (False, True):
        net = _net(nc).to(dev)
        net.train()
        net.engine().set_precision(precision)
        loss = DC_and_CE_DS_loss(w, batch_dice=False) if kind == 'softmax' else MultiTalentLoss(w, batch_dice=True)
        step = FusedTrainStep(net, loss, lr=1e-2)
        step.use_graph = use_graph
        losses = []
        for it, (x, largs) in enumerate(batches):
            if it == 5:
                step.lr = 5e-3                          # (the poly schedule: a new learning rate = a new capture)
            r = step(x, *largs)
            losses.append(float(r[0] if isinstance(r, tuple) else r))
        torch.cuda.synchronize()
        assert (step._graph is not None) == use_graph
        results.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, step.last_logits.detach().clone()))
    (l0, sd0, lg0), (l1, sd1, lg1) = results
    assert l0 == l1, (l0, l1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    assert torch.equal(lg0, lg1)
    assert np.isfinite(l0).all() and l0[-1] != l0[0]


def test_graph_is_dropped_for_validation_and_other_shapes(dev):
    """do_backprop=False (validation) runs eagerly between replays and sees the replayed steps' weights; a batch of another shape
    captures its own graph."""
    from multitalent_amd.training.hot_loop import FusedTrainStep
    from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
    net = _net(3).to(dev)
    net.train()
    step = FusedTrainStep(net, DC_and_CE_DS_loss([4 / 7, 2 / 7, 1 / 7], batch_dice=False), lr=1e-2)
    step.use_graph = True
    g = torch.Generator().manual_seed(5)

    def batch(patch):
        shapes = [patch, tuple(p // 2 for p in patch), tuple(p // 4 for p in patch)]
        return torch.randn((2, 1) + patch, generator=g).to(dev), [torch.randint(0, 3, (2, 1) + s, generator=g).float().to(dev) for s in shapes]
    x, tg = batch((8, 32, 32))
    for _ in range(4):
        step(x, tg)
    assert step._graph is not None
    v0 = float(step(x, tg, do_backprop=False))
    step(x, tg)
    v1 = float(step(x, tg, do_backprop=False))
    assert v0 != v1                                    # the replayed step changed the weights the eager validation pass reads
    x2, tg2 = batch((8, 16, 32))
    k0 = step._graph['key']
    for _ in range(4):                                 # eager steps with the new shape (re-planning, then two with the settled plan), then its own capture
        a = float(step(x2, tg2))
    assert step._graph['key'] != k0 and np.isfinite(a)
