"""evaluate -- mirror of regda/utils/eval.py:14-56: eval-mode sliding-window inference over a loader, argmax, confusion
matrix, mIoU with class 0 dropped for IsprsDA.  Datasets / loaders and the colour visualisations are outside the
path: the loader is passed in (`dataloader=`), anything yielding `(image (1,3,H,W), {'cls': (1,H,W)})`."""
import torch

from .. import ops
from ..gast.metrics import PixelMetricIgnore
from .tools import pre_slide


def evaluate(model, cfg, is_training=False, ckpt_path=None, logger=None, slide=True, tta=False, test=False,
             dataloader=None, class_names=None):
    ignore_labels = [0] if getattr(cfg, 'DATASETS', None) == 'IsprsDA' else []
    if dataloader is None:
        raise ValueError('regda_amd.utils.eval.evaluate needs dataloader=: the dataset classes are not part of this build')
    if not is_training:
        model.load_state_dict(torch.load(ckpt_path), strict=True)
        if logger is not None:
            logger.info('[Load params] from {}'.format(ckpt_path))
    num_class = getattr(cfg, 'NUM_CLASSES', None) or model.num_classes
    model.eval()
    names = list(class_names) if class_names is not None else [str(i) for i in range(num_class)]
    metric_op = PixelMetricIgnore(len(names), class_names=names, logdir=getattr(cfg, 'SNAPSHOT_DIR', None), logger=logger,
                                  ignore_labels=ignore_labels)
    with torch.no_grad():
        for ret, ret_gt in dataloader:
            ret = ret.cuda()
            cls = pre_slide(model, ret, num_classes=num_class, tta=tta) if slide else model(ret)
            pred = ops.argmax_nchw(cls)
            metric_op.forward(ret_gt['cls'].to('cuda', torch.int64), pred)      # y_true < 0 is masked in the kernel
    return metric_op.summary_all()
