"""Gradient / logit accuracy of the mixed-precision mode on the full-size residual-encoder network (BASELINE configs[3]) against the
fp64 oracle, for several (storage, compute-threshold) settings.  usage: python tools/bf16_accuracy.py "storage,minvox,act" ...   (storage 0|1, act fp16|bf16)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import test_fullsize_oracle_gpu as T
    dev = torch.device('cuda:0')
    settings = [tuple(a.split(',')) for a in sys.argv[1:]] or [('0', '0', 'fp16'), ('1', '0', 'fp16'), ('1', '0', 'bf16')]
    os.environ['MT_BF16_STORAGE'] = '0'
    sd0, x, tg, valid, w, logits, loss, grads, names = T._resenc(dev, 'fp32')
    o32, o64 = T._resenc_oracle(sd0, x, tg, valid, w)
    sd, out = o64[0], o32[1]
    names_g = list(grads.keys())
    gr = torch.cat([(sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])).reshape(-1) for n in names_g]).double()

    def report(tag, lg, gd):
        ga = torch.cat([gd[n].reshape(-1) for n in names_g]).double()
        cos = float((ga * gr).sum() / (ga.norm() * gr.norm()))
        lrel = ['%.4f' % float((a - b.detach()).norm() / b.detach().norm()) for a, b in zip(lg, out)]
        worst = min(((float((gd[n].double().reshape(-1) * sd[n].grad.reshape(-1)).sum() / (gd[n].double().norm() * sd[n].grad.norm() + 1e-30)), n)
                     for n in gd if n.endswith('.weight') and gd[n].dim() == 5 and gd[n].numel() > 50000), key=lambda t: t[0])
        print("%-28s cos %.5f  rel.L2 %.3e  logits rel.L2 %s  worst tensor cos %.4f (%s)" % (
            tag, cos, float((ga - gr).norm() / gr.norm()), lrel, worst[0], worst[1]), flush=True)
    report('fp32', logits, grads)
    del logits, grads
    for st, mv, act in settings:
        os.environ['MT_BF16_STORAGE'] = str(st)
        os.environ['MT_BF16_MIN_VOXELS'] = str(mv)
        os.environ['MT_ACT_STORAGE'] = act
        torch.cuda.empty_cache()
        _, _, _, _, _, lb, lossb, gb, nb = T._resenc(dev, 'bf16')
        report('mixed storage=%s minvox=%s act=%s' % (st, mv, act), lb, gb)
        del lb, gb


if __name__ == '__main__':
    main()
