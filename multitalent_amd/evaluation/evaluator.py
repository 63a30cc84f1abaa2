"""`aggregate_scores` as `trainer.validate()` calls it (reference evaluation/evaluator.py:30-56,312-400 with the confusion-matrix
metrics of evaluation/metrics.py:105-311): per case and label the thirteen default metrics, their nan-mean over the cases, and the
reference's summary.json layout.  Host bookkeeping after the device work is done; the medpy surface distances ("advanced"
metrics, off by default in the reference too) are not provided."""
import hashlib
import json
from collections import OrderedDict
from datetime import datetime

import numpy as np

from ..utilities.nifti_io import read_image

DEFAULT_METRICS = ["False Positive Rate", "Dice", "Jaccard", "Precision", "Recall", "Accuracy", "False Omission Rate",
                   "Negative Predictive Value", "False Negative Rate", "True Negative Rate", "False Discovery Rate",
                   "Total Positives Test", "Total Positives Reference"]


def confusion_metrics(test, reference):
    """test / reference: boolean masks of one label -> OrderedDict of DEFAULT_METRICS (NaN where the reference returns NaN)."""
    tp = int((test & reference).sum())
    fp = int((test & ~reference).sum())
    fn = int((~test & reference).sum())
    tn = int(test.size) - tp - fp - fn
    test_empty, test_full = (tp + fp) == 0, (tn + fn) == 0
    ref_empty, ref_full = (tp + fn) == 0, (tn + fp) == 0
    nan = float("NaN")
    spec = nan if ref_full else tn / (tn + fp)
    prec = nan if test_empty else tp / (tp + fp)
    sens = nan if ref_empty else tp / (tp + fn)
    fom = nan if test_full else fn / (fn + tn)
    both_empty = test_empty and ref_empty
    m = OrderedDict()
    m["False Positive Rate"] = 1 - spec
    m["Dice"] = nan if both_empty else 2. * tp / (2 * tp + fp + fn)
    m["Jaccard"] = nan if both_empty else tp / (tp + fp + fn)
    m["Precision"] = prec
    m["Recall"] = sens
    m["Accuracy"] = (tp + tn) / (tp + fp + tn + fn)
    m["False Omission Rate"] = fom
    m["Negative Predictive Value"] = 1 - fom
    m["False Negative Rate"] = 1 - sens
    m["True Negative Rate"] = spec
    m["False Discovery Rate"] = 1 - prec
    m["Total Positives Test"] = tp + fp
    m["Total Positives Reference"] = tp + fn
    return OrderedDict((k, float(v)) for k, v in m.items())


def evaluate_case(test_file, ref_file, labels):
    """labels: iterable of ints or tuples of ints (a tuple = the union of its members, evaluator.py:140-160)."""
    test = np.asarray(read_image(test_file).array) if isinstance(test_file, str) else np.asarray(test_file)
    ref = np.asarray(read_image(ref_file).array) if isinstance(ref_file, str) else np.asarray(ref_file)
    if test.shape != ref.shape:
        raise ValueError("Shape mismatch: %s and %s" % (test.shape, ref.shape))
    res = OrderedDict()
    for l in labels:
        members = l if isinstance(l, (tuple, list)) else (l,)
        t = np.isin(test, list(members))
        r = np.isin(ref, list(members))
        res[str(l)] = confusion_metrics(t, r)
    res["reference"] = ref_file if isinstance(ref_file, str) else None
    res["test"] = test_file if isinstance(test_file, str) else None
    return res


def aggregate_scores(test_ref_pairs, labels=None, nanmean=True, json_output_file=None, json_name="", json_description="",
                     json_author="Fabian", json_task="", num_threads=2, **_):
    if labels is None:
        raise ValueError("labels must be given")
    scores = OrderedDict(all=[], mean=OrderedDict())
    for test, ref in test_ref_pairs:
        scores["all"].append(evaluate_case(test, ref, labels))
    for res in scores["all"]:
        for label, sd in res.items():
            if label in ("test", "reference"):
                continue
            dst = scores["mean"].setdefault(label, OrderedDict())
            for k, v in sd.items():
                dst.setdefault(k, []).append(v)
    for label in scores["mean"]:
        for k in scores["mean"][label]:
            v = scores["mean"][label][k]
            with np.errstate(all='ignore'):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    scores["mean"][label][k] = float(np.nanmean(v) if nanmean else np.mean(v))
    if json_output_file is not None:
        d = OrderedDict()
        d["name"], d["description"], d["timestamp"] = json_name, json_description, str(datetime.today())
        d["task"], d["author"], d["results"] = json_task, json_author, scores
        d["id"] = hashlib.md5(json.dumps(d).encode("utf-8")).hexdigest()[:12]
        with open(json_output_file, 'w') as f:
            json.dump(d, f, sort_keys=True, indent=4)
    return scores
