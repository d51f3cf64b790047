"""pseudo_selection + gener_target_pseudo -- mirror of regda/gast/pseudo_generation.py:59-141."""
import os

import torch

from .. import ops


def pseudo_selection(mask, cutoff_top=0.8, cutoff_low=0.6, return_type='ndarray', ignore_label=-1):
    """soft (b,c,h,w) probabilities -> hard labels (b,h,w) int64; same arguments, asserts and
    return types as the reference."""
    assert return_type in ['ndarray', 'tensor']
    ret = ops.pseudo_select(mask, cutoff_top, cutoff_low, ignore_label, check=True)
    if return_type == 'ndarray':
        return ret.cpu().numpy()
    return ret


def gener_target_pseudo(_cfg, model, pseudo_loader, save_pseudo_label_path, slide=True, save_prob=False,
                        size=(1024, 1024), ignore_label=-1):
    """Teacher pass over the target set (pseudo_generation.py:96-141): eval-mode sliding-window + 8-view TTA
    inference per tile; `save_prob` writes the soft labels the SSL loader reads back -- a (C, h, w) fp32 CPU tensor
    `torch.save`d as `<fname>.pt` (pseudo_generation.py:135-136, basedata.py:86).  The reference's colour visualisation (VisualizeSegmm)
    and its cv2 hard-label images (save_prob=False) are not part of the path and are not reproduced."""
    from ..utils.tools import pre_slide
    model.eval()
    os.makedirs(save_pseudo_label_path, exist_ok=True)
    num_classes = getattr(_cfg, 'NUM_CLASSES', None) or model.num_classes
    with torch.no_grad():
        for ret, ret_gt in pseudo_loader:
            ret = ret.cuda()
            cls = pre_slide(model, ret, num_classes=num_classes, tta=True) if slide else model(ret)
            if save_prob:
                out = ops.resize_bilinear_ac(cls, size) if tuple(cls.shape[-2:]) != tuple(size) else cls
                torch.save(out.squeeze(dim=0).cpu(), os.path.join(save_pseudo_label_path, ret_gt['fname'][0] + '.pt'))
            else:
                # hard labels are written by the reference as uint8 images through cv2 (pseudo_generation.py:149-150);
                # image file formats are outside the path, and the SSL driver never takes this branch
                # (train_ssl_reg.py:188-189 passes save_prob=True)
                raise NotImplementedError('gener_target_pseudo(save_prob=False): hard-label image output is out of scope')
