"""Aligner (prototype EMA + online soft-label re-weighting) -- mirror of the parts of
regda/gast/alignment.py that sit on the SSL path: label_refine (:194-265, with
label_t_sup=None as tools/train_ssl_reg.py:214 calls it; every `mode`), update_prototype (:86-90),
DownscaleLabel (:456-481).  Stage-2 alignment losses are out of scope.
"""
import torch

from .. import ops


class DownscaleLabel(torch.nn.Module):
    def __init__(self, scale_factor=16, n_classes=7, ignore_label=-1, min_ratio=0.75):
        super().__init__()
        assert scale_factor > 1
        self.scale_factor, self.n_classes = scale_factor, n_classes
        self.ignore_label, self.min_ratio = ignore_label, min_ratio

    def forward(self, label):
        if label.dim() == 4:
            label = label.squeeze(1)
        b, H, W = label.shape
        s = self.scale_factor
        # the kernel fuses the downscale with the prototype pass; run it against a dummy feature
        feat = torch.zeros((b, 4, H // s, W // s), device=label.device)
        protos = torch.zeros((self.n_classes, 4), device=label.device)
        return ops.proto_update(feat, label, protos, s, self.ignore_label, self.min_ratio, 0.5)


class Aligner:
    def __init__(self, logger, feat_channels=64, class_num=7, ignore_label=-1, decay=0.999, topk=32, resume=None):
        self.feat_channels = feat_channels
        self.class_num = class_num
        self.ignore_label = ignore_label
        self.decay = decay
        self.logger = logger
        self.eps = 1e-7
        if resume:
            self.prototypes = torch.load(resume, map_location='cpu').float().cuda().contiguous()
            if logger is not None:
                logger.info('finish init prototypes!')
        else:
            self.prototypes = torch.zeros([class_num, feat_channels], device='cuda')
        self.downscale_gt = DownscaleLabel(scale_factor=16, n_classes=class_num, ignore_label=ignore_label,
                                           min_ratio=0.75)
        self._classmax_ws = None

    def update_prototype(self, feat, label):
        """Update global prototypes by source features and labels (alignment.py:86-90)."""
        return ops.proto_update(feat.detach(), label, self.prototypes, 16, self.ignore_label, 0.75, self.decay)

    def label_refine(self, label_t_sup, feat_t, preds_t, label_t_soft, refine=True, mode='all', temp=2.0):
        """alignment.py:194-265.  Built: every `mode` with label_t_sup=None (the SSL path passes None,
        tools/train_ssl_reg.py:214) and one or two prediction tensors.  The superpixel view (label_t_sup given, modes
        'all' / 's') is not."""
        assert mode in ['all', 's', 'p', 'n', 'l']
        if not refine:
            return label_t_soft
        if label_t_sup is not None and mode in ('all', 's'):
            raise NotImplementedError('the superpixel view of label_refine (label_t_sup given) is not built; '
                                      'see DESIGN.md "out of scope"')
        if mode in ('s', 'n'):
            return label_t_soft                  # no view contributes: `weight` stays the int 0 (alignment.py:260-261)
        if isinstance(preds_t, (list, tuple)):
            assert len(preds_t) == 2
            p1, p2 = preds_t
        else:
            p1 = p2 = preds_t                    # (s + s) * 0.5 == s: the single-tensor branch, alignment.py:232-234
        views = {'all': 3, 'p': 1, 'l': 2}[mode]
        out, cm = ops.label_refine(feat_t.detach(), self.prototypes, p1.detach() if views & 2 else None,
                                   p2.detach() if views & 2 else None, label_t_soft, temp, return_ws=True, views=views)
        self._classmax_ws = cm       # per-image per-class maxima of the result (reused by the fused trainer)
        return out
