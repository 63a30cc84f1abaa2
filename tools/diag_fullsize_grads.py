"""Per-tensor gradient error of the HIP path at full size (Task009 network, B = 1, 48x192x192) against the fp32 AND the fp64 CPU
oracle: separates kernel error from the conditioning of the problem (the fp32 oracle itself is ~1e-3..1e-2 away from fp64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from oracle import reference_ops as R
from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets
from multitalent_amd.training.loss_functions.fused_losses import DC_and_CE_DS_loss
from test_fullsize_oracle_gpu import hip_forward_backward
from multitalent_amd import ops

dev = torch.device('cuda:0')
torch.set_num_threads(32)
for opt in (sys.argv[1:] or ['default']):
    if opt != 'default':
        k, v = opt.split('=')
        ops.set_option(k, int(v))
    torch.manual_seed(1234)
    net = bench.build_network('task009')
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.train()
    x = synthetic_ct(1, bench.PATCH, 77, dev)
    tg = synthetic_targets(1, bench.PATCH, ds_scales(bench.POOLS), [[1]], 77, dev)
    w = R.ds_loss_weights(len(bench.POOLS))
    logits, loss, grads = hip_forward_backward(net, DC_and_CE_DS_loss(w, batch_dice=False), x, (tg,))
    print('== option', opt, 'hip loss', loss, flush=True)
    if opt == (sys.argv[1:] or ['default'])[0]:
        ref = {}
        for dt in (torch.float32, torch.float64):
            t0 = time.time()
            sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
            out = R.generic_unet_forward(sd, x.cpu().to(dt), bench.POOLS, bench.KERNELS)
            l = R.multiple_output_loss(out, [t.cpu() for t in tg], w)
            l.backward()
            ref[dt] = ({k: (v.grad.double() if v.grad is not None else None) for k, v in sd.items()}, [o.detach().double() for o in out], float(l))
            print(dt, 'loss', float(l), '%.1f s' % (time.time() - t0), flush=True)
    g32, g64 = ref[torch.float32][0], ref[torch.float64][0]
    print('logit max diff vs fp64:', [float((a.double() - b).abs().max()) for a, b in zip(logits, ref[torch.float64][1])])
    rows = []
    for k, g in grads.items():
        if g64[k] is None:
            continue
        g = g.double(); t = g64[k]
        rows.append((float((g - t).norm() / (t.norm() + 1e-30)), float((g32[k] - t).norm() / (t.norm() + 1e-30)),
                     float((g - t).abs().max()), float((g32[k] - t).abs().max()), float(t.abs().max()), float(t.norm()), k))
    rows.sort(reverse=True)
    print('%-11s %-11s %-11s %-11s %-10s %-10s' % ('hip l2rel', 'cpu32 l2rel', 'hip maxabs', 'cpu32 maxabs', 'max|ref|', '|ref|_2'))
    for r in rows[:40]:
        print('%.3e   %.3e   %.3e   %.3e   %.2e   %.2e  %s' % r)
    ga = torch.cat([grads[k].double().reshape(-1) for k in grads if g64[k] is not None])
    gt = torch.cat([g64[k].reshape(-1) for k in grads if g64[k] is not None])
    gc = torch.cat([g32[k].reshape(-1) for k in grads if g64[k] is not None])
    print('global: hip l2rel %.3e cos %.8f | cpu32 l2rel %.3e' % (float((ga - gt).norm() / gt.norm()), float((ga * gt).sum() / (ga.norm() * gt.norm())), float((gc - gt).norm() / gt.norm())))
