"""Mask comparison used by the sliding-window parity tests: instead of silently dropping voxels near a
decision boundary, every voxel is accounted for —
  * away from ties (reference margin > tol) the masks must be identical;
  * EVERYWHERE the mask must be the reference's decision rule (neural_network.py:404-412: thresholds painted in
    regions_class_order, or argmax) applied to the probabilities this path produced — so a differing voxel can only come from a
    probability difference below `tol`, never from a different rule;
  * wherever the probabilities are bit-identical to the reference's, the masks must be identical, ties included;
  * the number of tie voxels (a property of the reference's probabilities alone: it must EQUAL the recorded count of the golden case) and
    how many of them actually differ (at most the recorded bound) are asserted against tests/golden/mask_tie_bounds.json — one entry
    per golden case, recorded with MT_RECORD_MASK_TIES=1 (counted, bounded and tracked, not printed)."""
import json
import os

import numpy as np

_BOUNDS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mask_tie_bounds.json')


def decide(probs, order):
    if order is None:
        return probs.argmax(0)
    seg = np.zeros(probs.shape[1:], dtype=np.float32)
    for i, c in enumerate(order):
        seg[probs[i] > 0.5] = c
    return seg


def check_masks(seg, ref_seg, probs, ref_probs, order, tol=1e-4, what='', max_tie_frac=1e-2, key=None, live=False):
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
    key = key or what
    if live:          # the reference side was computed in this run (no golden case, no recorded tie count)
        return int(ties.sum()), ndiff
    if os.environ.get('MT_RECORD_MASK_TIES') == '1':
        out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'mask_tie_bounds.json')
        rec = json.load(open(out)) if os.path.exists(out) else {}
        rec[key] = {'voxels': int(seg.size), 'ties': int(ties.sum()), 'max_differing': max(ndiff, rec.get(key, {}).get('max_differing', 0))}
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(rec, open(out, 'w'), indent=1, sort_keys=True)
    else:
        bounds = json.load(open(_BOUNDS))
        assert key in bounds, "%s: no recorded tie bound for this golden case (measured: %d ties, %d differing)" % (key, int(ties.sum()), ndiff)
        b = bounds[key]
        assert int(ties.sum()) == b['ties'] and seg.size == b['voxels'], "%s: %d tie voxels of %d, the golden case has %d of %d" % (key, int(ties.sum()), seg.size, b['ties'], b['voxels'])
        assert ndiff <= b['max_differing'], "%s: %d tie voxels differ from the reference, recorded bound %d" % (key, ndiff, b['max_differing'])
    return int(ties.sum()), ndiff
