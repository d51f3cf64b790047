"""Aligner (prototype EMA + online soft-label re-weighting) -- mirror of the parts of
regda/gast/alignment.py that sit on the SSL path: label_refine (:194-265,
with and without the superpixel view; every `mode`), update_prototype (:86-90),
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
        """alignment.py:194-265: every `mode`, one or two prediction tensors, with or without the superpixel view
        (label_t_sup (b,1,H,W) int64 given and mode 'all' / 's', :238-258; the SSL path passes None,
        tools/train_ssl_reg.py:214).  `max_superpixels` (attribute, default 65536) bounds the ids: the table of
        per-superpixel maxima is sized by it instead of by a read-back of label_t_sup.max() (the reference has no bound:
        torch_scatter sizes its output by the largest id); ids beyond it raise ValueError -- that check reads a flag back
        (one host sync per call); `check_superpixel_range = False` (attribute) skips it and keeps the call enqueue-only."""
        assert mode in ['all', 's', 'p', 'n', 'l']
        if not refine:
            return label_t_soft
        sup = label_t_sup is not None and mode in ('all', 's')
        if mode == 'n' or (mode == 's' and not sup):
            return label_t_soft                  # no view contributes: `weight` stays the int 0 (alignment.py:260-261)
        views = {'all': 3, 'p': 1, 'l': 2, 's': 0}[mode]
        p1 = p2 = None
        if views & 2:
            if isinstance(preds_t, (list, tuple)):
                assert len(preds_t) == 2
                p1, p2 = preds_t
            else:
                p1 = p2 = preds_t                # (s + s) * 0.5 == s: the single-tensor branch, alignment.py:232-234
            p1, p2 = p1.detach(), p2.detach()
        feat = feat_t.detach() if views & 1 else None
        if sup:
            out, cm = ops.label_refine_sup(feat, self.prototypes, p1, p2, label_t_soft, label_t_sup.long(), temp, views,
                                           max_regions=getattr(self, 'max_superpixels', 65536), return_ws=True,
                                           check=getattr(self, 'check_superpixel_range', True))
        else:
            out, cm = ops.label_refine(feat, self.prototypes, p1, p2, label_t_soft, temp, return_ws=True, views=views)
        self._classmax_ws = cm       # per-image per-class maxima of the result (reused by the fused trainer)
        return out
