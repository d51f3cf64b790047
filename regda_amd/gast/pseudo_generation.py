"""pseudo_selection + gener_target_pseudo -- mirror of regda/gast/pseudo_generation.py:59-141."""
import os

import numpy as np

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
    """Teacher pass over the target set (pseudo_generation.py:96-153): eval-mode sliding-window + 8-view TTA
    inference per tile.  `save_prob=True` writes the soft labels the SSL loader reads back -- a (C, h, w) fp32 CPU tensor
    `torch.save`d as `<fname>.pt` (pseudo_generation.py:135-136, basedata.py:86).  `save_prob=False` writes the HARD
    labels as the reference does (pseudo_generation.py:143-150): pseudo_selection (or, with `_cfg.PSEUDO_SELECT` false,
    the argmax) of the tile's probabilities, + 1, reshaped to `size`, as a single-channel uint8 image named `<fname>` --
    the reference hands that array to cv2.imwrite; here Pillow writes it (the container's bytes differ, the decoded
    pixels are the same array).  The colour visualisations (VisualizeSegmm) are not reproduced."""
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
                if _cfg.PSEUDO_SELECT:                # required attribute, as in the reference (:144)
                    lab = pseudo_selection(cls, ignore_label=ignore_label)      # the reference's call: default cut-offs, ndarray (:145)
                else:
                    lab = ops.argmax_nchw(cls).cpu().numpy()
                from PIL import Image
                arr = (np.asarray(lab) + 1).reshape(*size).astype(np.uint8)      # -1 .. C-1  ->  0 .. C  (:149-150)
                # (a uint8 2-D array is mode 'L' by itself; the `mode=` argument is deprecated in Pillow >= 11.3)
                Image.fromarray(arr).save(os.path.join(save_pseudo_label_path, ret_gt['fname'][0]))
