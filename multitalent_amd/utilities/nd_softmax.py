"""softmax_helper = F.softmax(x, 1) (reference utilities/nd_softmax.py); recognised by the sliding-window engine so the
softmax is fused into the tile accumulation kernel."""
import torch.nn.functional as F


def softmax_helper(x):
    return F.softmax(x, 1)
