"""Deep-supervision loss weights (nnUNetTrainerV2.py:78-90): product-side copy so bench.py's timed path never
imports the oracle."""
import numpy as np


def ds_loss_weights(net_numpool):
    w = np.array([1 / (2 ** i) for i in range(net_numpool)])
    w[net_numpool - 1:] = 0
    return w / w.sum()
