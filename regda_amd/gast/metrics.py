"""PixelMetricIgnore -- mirror of regda/gast/metrics.py:19-65 (on top of ever's PixelMetric, restated: `ever` is not
vendored with the reference).  The confusion matrix lives on the GPU (int64 [C][C], rgda_confusion_accumulate)."""
import numpy as np
import torch

from .. import ops


class PixelMetricIgnore:
    def __init__(self, num_classes, logdir=None, logger=None, class_names=None, ignore_labels=list()):
        self.num_classes = num_classes
        self.logdir, self.logger = logdir, logger
        self._class_names = list(class_names) if class_names else None
        self.ignore_labels = sorted(ignore_labels, reverse=True)
        self._total = torch.zeros(num_classes, num_classes, dtype=torch.int64, device='cuda')
        self._flag = torch.zeros(1, dtype=torch.int32, device='cuda')

    def forward(self, y_true, y_pred):
        """Accumulate one batch; pixels with y_true < 0 are skipped (eval.py:45-50 masks them before the call --
        doing it here is the same count).  numpy arrays or tensors on any device."""
        yt = torch.as_tensor(y_true).to('cuda', torch.int64)
        yp = torch.as_tensor(y_pred).to('cuda', torch.int64)
        ops.confusion_accumulate(yt, yp, self._total, self._flag)

    __call__ = forward

    def confusion_matrix(self):
        if int(self._flag.item()) != 0:
            raise ValueError('PixelMetricIgnore: a label >= num_classes or a prediction outside [0, num_classes) was seen')
        return self._total.cpu().numpy()

    def summary_all(self, dec=5):
        """-> (table text, mIoU): per-class values rounded to `dec`, `ignore_labels` dropped, then the rounded means
        (metrics.py:25-45).  The table is plain text (prettytable is not a dependency here)."""
        cm = self.confusion_matrix().astype(np.float64)
        diag, true_cnt, pred_cnt = np.diag(cm), cm.sum(axis=1), cm.sum(axis=0)
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = diag / (true_cnt + pred_cnt - diag)
            precision, recall = diag / pred_cnt, diag / true_cnt
            f1 = 2 * precision * recall / (precision + recall)
        cols = [np.round(v, dec).tolist() for v in (iou, f1, precision, recall)]
        names = list(self._class_names) if self._class_names else None
        for idx in self.ignore_labels:
            for c in cols:
                c.pop(idx)
            if names:
                names.pop(idx)
        iou, f1, precision, recall = cols
        miou, mf1, mprec, mrec = (np.round(np.array(c).mean(), dec) for c in (iou, f1, precision, recall))
        head = (['name'] if names else []) + ['class', 'iou', 'f1', 'precision', 'recall']
        rows = [head]
        for i in range(len(iou)):
            rows.append(([names[i]] if names else []) + [i, iou[i], f1[i], precision[i], recall[i]])
        rows.append(([''] if names else []) + ['mean', miou, mf1, mprec, mrec])
        table = '\n'.join(' | '.join(str(v) for v in r) for r in rows)
        if self.logger is not None:
            self.logger.info('\n' + table)
        return table, miou
