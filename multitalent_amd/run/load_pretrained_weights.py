"""Transfer of pretrained MultiTalent weights into a new network (reference run/load_pretrained_weights.py:18-62; readme.md:51-64).

Every tensor whose key exists in both state dicts with the same shape is copied ('module.' prefixes of DDP checkpoints are
stripped); all `conv_blocks*` keys of the target MUST be present with matching shapes, otherwise the checkpoint is incompatible
and a RuntimeError is raised.  Segmentation heads with a different number of classes are therefore left at their initialisation —
the fine-tuning trainers (`nnUNetTrainerV2_warmupsegheads*`) train them first."""
import torch


def load_pretrained_weights(network, fname, verbose=False):
    saved_model = torch.load(fname, map_location='cpu', weights_only=False)
    pretrained = {(k[7:] if k.startswith('module.') else k): v for k, v in saved_model['state_dict'].items()}
    model_dict = network.state_dict()
    for key in model_dict:
        if 'conv_blocks' in key and not (key in pretrained and model_dict[key].shape == pretrained[key].shape):
            raise RuntimeError("Pretrained weights are not compatible with the current network architecture")
    pretrained = {k: v for k, v in pretrained.items() if k in model_dict and model_dict[k].shape == v.shape}
    model_dict.update(pretrained)
    print("################### Loading pretrained weights from file ", fname, '###################')
    if verbose:
        print("Below is the list of overlapping blocks in pretrained model and nnUNet architecture:")
        for key in pretrained:
            print(key)
    print("################### Done ###################")
    network.load_state_dict(model_dict)
    if hasattr(network, 'engine'):
        network.engine().mark_params_dirty()          # packed weights are re-derived on the next step
    return list(pretrained.keys())
