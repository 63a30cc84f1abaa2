"""Is the HIP path's fp32 gradient error at full size (2.1e-3 against the fp64 oracle, where torch fp32 on the CPU has 0.96e-3:
tools/diag_fullsize_grads.py) explained by HOW the InstanceNorm is written?  CPU-only emulation: the oracle's fp32 run repeated with
F.instance_norm replaced by the product's forward formulation — per-(n, c) sums of y and y^2 taken in fp32 over blocks of voxels and
combined in fp64 (mt_inorm_finalize), variance = E[y^2] - mean^2, then ONE multiply-add y * scale + shift with fp32 scale = gamma * rstd,
shift = beta - mean * scale (the lazy activation every consumer applies on load) — gradients by autograd through that expression.
usage: python tools/emulate_lazy_norm_cpu.py [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from oracle import reference_ops as R
from multitalent_amd.synthetic import ds_scales, synthetic_ct, synthetic_targets

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
dev = torch.device('cpu')
torch.manual_seed(1234)
net = bench.build_network('task009')
sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
x = synthetic_ct(1, bench.PATCH, 77, dev)
tg = synthetic_targets(1, bench.PATCH, ds_scales(bench.POOLS), [[1]], 77, dev)
w = R.ds_loss_weights(len(bench.POOLS))
orig = R.F.instance_norm
BLK = 2048


MODE = 'both'          # 'both' | 'stats' (blocked E[y^2] - mean^2 statistics, textbook (y - mean) * rstd * gamma + beta) | 'fma' (two-pass statistics, single multiply-add)


def lazy_instance_norm(x, running_mean=None, running_var=None, weight=None, bias=None, use_input_stats=True, momentum=0.1, eps=1e-5):
    if x.dtype != torch.float32:
        return orig(x, weight=weight, bias=bias, eps=eps)
    if MODE == 'sub':       # what a consumer could apply instead: (y - mean) * scale + beta with the blocked statistics (one more instruction per element)
        N, C = x.shape[:2]
        xf = x.reshape(N, C, -1)
        V = xf.shape[2]
        nb = (V + BLK - 1) // BLK
        pad = nb * BLK - V
        xb = (torch.nn.functional.pad(xf, (0, pad)) if pad else xf).reshape(N, C, nb, BLK)
        mean = xb.sum(-1).double().sum(-1) / V
        var = (xb * xb).sum(-1).double().sum(-1) / V - mean * mean
        scale = (weight.double()[None] / torch.sqrt(var + eps)).float()
        sh = (N, C) + (1,) * (x.dim() - 2)
        return torch.addcmul(bias.reshape((1, C) + (1,) * (x.dim() - 2)), x - mean.float().reshape(sh), scale.reshape(sh))
    if MODE != 'both':
        N, C = x.shape[:2]
        xf = x.reshape(N, C, -1)
        V = xf.shape[2]
        sh = (N, C) + (1,) * (x.dim() - 2)
        if MODE == 'stats':
            nb = (V + BLK - 1) // BLK
            pad = nb * BLK - V
            xb = (torch.nn.functional.pad(xf, (0, pad)) if pad else xf).reshape(N, C, nb, BLK)
            mean = xb.sum(-1).double().sum(-1) / V
            var = (xb * xb).sum(-1).double().sum(-1) / V - mean * mean
            rstd = (1.0 / torch.sqrt(var + eps)).float()
            return (x - mean.float().reshape(sh)) * rstd.reshape(sh) * weight.reshape((1, C) + (1,) * (x.dim() - 2)) + bias.reshape((1, C) + (1,) * (x.dim() - 2))
        mean = xf.double().mean(-1)
        var = ((xf.double() - mean[..., None]) ** 2).mean(-1)
        rstd = 1.0 / torch.sqrt(var + eps)
        scale = (weight.double()[None] * rstd).float()
        shift = (bias.double()[None] - mean * scale.double()).float()
        return torch.addcmul(shift.reshape(sh), x, scale.reshape(sh))
    N, C = x.shape[:2]
    xf = x.reshape(N, C, -1)
    V = xf.shape[2]
    nb = (V + BLK - 1) // BLK
    pad = nb * BLK - V
    xp = torch.nn.functional.pad(xf, (0, pad)) if pad else xf
    xb = xp.reshape(N, C, nb, BLK)
    s1 = xb.sum(-1).double().sum(-1)                       # fp32 partial per block, fp64 over the blocks
    s2 = (xb * xb).sum(-1).double().sum(-1)
    mean = s1 / V
    var = s2 / V - mean * mean
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = (weight.double()[None] * rstd).float()
    shift = (bias.double()[None] - mean * scale.double()).float()
    sh = (N, C) + (1,) * (x.dim() - 2)
    return torch.addcmul(shift.reshape(sh), x, scale.reshape(sh))


def run(dt, patched):
    R.F.instance_norm = lazy_instance_norm if patched else orig
    try:
        t0 = time.time()
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
        out = R.generic_unet_forward(sd, x.to(dt), bench.POOLS, bench.KERNELS)
        l = R.multiple_output_loss(out, list(tg), w)
        l.backward()
        print(dt, 'patched' if patched else 'reference', 'loss %.10f' % float(l), '%.1f s' % (time.time() - t0), flush=True)
        return {k: v.grad.double() for k, v in sd.items() if v.grad is not None}
    finally:
        R.F.instance_norm = orig


g64 = run(torch.float64, False)
g32 = run(torch.float32, False)
res = {}
for MODE in ('both', 'stats', 'fma', 'sub'):
    res[MODE] = run(torch.float32, True)
glz = res['both']
tot = float(torch.cat([g.reshape(-1) for g in g64.values()]).norm())
keys = [k for k in g64 if k in g32 and k in glz and float(g64[k].norm()) > 1e-6 * tot]       # (conv biases in front of a norm have an exactly zero gradient)
cat = lambda g: torch.cat([g[k].reshape(-1) for k in keys])
t = cat(g64)
err = lambda g: float((cat(g) - t).norm() / t.norm())
print('global relative L2 error of all gradients against fp64: torch fp32 %.3e | blocked E[y^2] - mean^2 statistics + single multiply-add %.3e | '
      'statistics only %.3e | single multiply-add only %.3e | (y - mean) * scale + beta on the blocked statistics %.3e'
      % (err(g32), err(res['both']), err(res['stats']), err(res['fma']), err(res['sub'])))
rows = sorted(((float((glz[k] - g64[k]).norm() / g64[k].norm()), float((g32[k] - g64[k]).norm() / g64[k].norm()), k) for k in keys), reverse=True)
for r in rows[:10]:
    print('  lazy %.3e   torch %.3e   %s' % r)
