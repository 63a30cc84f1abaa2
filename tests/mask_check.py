"""Mask comparison used by the sliding-window parity tests: instead of silently dropping voxels near a
decision boundary, every voxel is accounted for —
  * away from ties (reference margin > tol) the masks must be identical;
  * EVERYWHERE the mask must be the reference's decision rule (neural_network.py:404-412: thresholds painted in
    regions_class_order, or argmax) applied to the probabilities this path produced — so a differing voxel can only come from a
    probability difference below `tol`, never from a different rule;
  * wherever the probabilities are bit-identical to the reference's, the masks must be identical, ties included;
  * the number of tie voxels and how many of them actually differ is printed and bounded."""
import numpy as np


def decide(probs, order):
    if order is None:
        return probs.argmax(0)
    seg = np.zeros(probs.shape[1:], dtype=np.float32)
    for i, c in enumerate(order):
        seg[probs[i] > 0.5] = c
    return seg


def check_masks(seg, ref_seg, probs, ref_probs, order, tol=1e-4, what='', max_tie_frac=1e-2):
    seg, ref_seg = np.asarray(seg), np.asarray(ref_seg)
    assert seg.shape == ref_seg.shape and probs.shape == ref_probs.shape
    if order is None:
        srt = np.sort(ref_probs, 0)
        margin = srt[-1] - srt[-2]
    else:
        margin = np.abs(ref_probs - 0.5).min(0)
    safe = margin > tol
    assert np.array_equal(seg[safe].astype(np.int64), ref_seg[safe].astype(np.int64)), "%s: masks differ away from ties" % what
    assert np.array_equal(seg.astype(np.int64), decide(probs, order).astype(np.int64)), "%s: mask is not the decision rule of its own probabilities" % what
    same = (probs == ref_probs).all(0)
    assert np.array_equal(seg[same].astype(np.int64), ref_seg[same].astype(np.int64)), "%s: identical probabilities, different masks" % what
    ties = ~safe
    ndiff = int((seg[ties].astype(np.int64) != ref_seg[ties].astype(np.int64)).sum())
    print("%s: %d voxels, %d within %g of a decision boundary (%.4f %%), %d of those differ from the reference; %d voxels with bit-identical "
          "probabilities" % (what, seg.size, int(ties.sum()), tol, 100.0 * ties.mean(), ndiff, int(same.sum())))
    assert ties.mean() <= max_tie_frac, "%s: %.4f of the voxels are ties" % (what, ties.mean())
    return int(ties.sum()), ndiff
