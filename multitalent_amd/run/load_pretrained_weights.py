"""Transfer of pretrained MultiTalent weights into a new network (reference run/load_pretrained_weights.py:18-62; readme.md:51-64).

Every tensor whose key exists in both state dicts with the same shape is copied ('module.' prefixes of DDP checkpoints are
stripped); all `conv_blocks*` keys of the target MUST be present with matching shapes, otherwise the checkpoint is incompatible
and a RuntimeError is raised.  Segmentation heads with a different number of classes are therefore left at their initialisation —
the fine-tuning trainers (`nnUNetTrainerV2_warmupsegheads*`) train them first."""
import torch


def load_pretrained_weights(network, fname, verbose=False):
    ckpt = torch.load(fname, map_location='cpu', weights_only=False)
    source = {(name[7:] if name.startswith('module.') else name): tensor for name, tensor in ckpt['state_dict'].items()}
    target = network.state_dict()
    missing = [k for k in target if 'conv_blocks' in k and (k not in source or source[k].shape != target[k].shape)]
    if missing:
        raise RuntimeError("Pretrained weights are not compatible with the current network architecture (first mismatch: %s)" % missing[0])
    transferred = [k for k in source if k in target and target[k].shape == source[k].shape]
    for k in transferred:
        target[k] = source[k]
    print("loading pretrained weights from %s: %d of %d tensors transferred" % (fname, len(transferred), len(target)))
    if verbose:
        for k in transferred:
            print("  ", k)
    network.load_state_dict(target)
    if hasattr(network, 'engine'):
        network.engine().mark_params_dirty()          # packed weights are re-derived on the next step
    return transferred
