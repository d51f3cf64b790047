"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of the evaluation path, SURVEY.md 8f
rank 3: regda/utils/eval.py:14-56 (argmax, the `cls_gt >= 0` mask, the confusion matrix) and
regda/gast/metrics.py:19-65 (`PixelMetricIgnore.summary_all`: per-class IoU / F1 / precision / recall rounded to
`dec` places, the classes in `ignore_labels` dropped, then the means).

Third-party arithmetic: `ever.api.metric.pixel.PixelMetric` (package `ever`, un-vendored, not installed) -- its
confusion matrix (scipy coo_matrix of (y_true, y_pred) pairs, rows = truth) and the textbook per-class formulas are
restated here; PARITY WITH `ever` ITSELF IS UNPINNED (no source, no vector).  What is pinned: the restatement equals
scikit-learn's confusion_matrix / jaccard / precision / recall / f1 on random labels with rows = truth
(tests/test_oracle_golden.py::test_eval_metrics_against_scikit_learn_definitions); IoU and mIoU, which is all the
training driver consumes (train_ssl_reg.py:256-259), are symmetric in the two axes anyway.
"""
import numpy as np


def confusion_matrix(y_true, y_pred, num_classes):
    """rows = true class, columns = predicted class, only pixels with y_true >= 0 (eval.py:45-50)."""
    y_true, y_pred = np.asarray(y_true).ravel(), np.asarray(y_pred).ravel()
    m = y_true >= 0
    cm = np.zeros((num_classes, num_classes), np.int64)
    np.add.at(cm, (y_true[m], y_pred[m]), 1)
    return cm


def per_class(cm):
    cm = cm.astype(np.float64)
    diag = np.diag(cm)
    true_cnt, pred_cnt = cm.sum(axis=1), cm.sum(axis=0)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = diag / (true_cnt + pred_cnt - diag)
        precision = diag / pred_cnt
        recall = diag / true_cnt
        f1 = 2 * precision * recall / (precision + recall)
    return iou, f1, precision, recall


def summary(cm, ignore_labels=(), dec=5):
    """metrics.py:25-45: round per class, pop the ignored classes (descending index), then round the means."""
    cols = [np.round(v, dec).tolist() for v in per_class(cm)]
    for idx in sorted(ignore_labels, reverse=True):
        for c in cols:
            c.pop(idx)
    iou, f1, prec, rec = cols
    means = [np.round(np.array(c).mean(), dec) for c in (iou, f1, prec, rec)]
    return dict(iou=iou, f1=f1, precision=prec, recall=rec, miou=means[0], mf1=means[1], mprecision=means[2],
                mrecall=means[3])
