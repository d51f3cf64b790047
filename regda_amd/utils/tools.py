"""loss_calc / learning-rate schedule / config import -- mirror of regda/utils/tools.py:173-207,240-260."""
import importlib
import os
import shutil

import torch.nn.functional as tnf


def import_config(config_name, prefix='configs', copy=True, create=True, postfix=''):
    """UPPER_CASE python-module configs addressed by dotted name, e.g. 'st.regda.2potsdam'
    (tools.py:173-181)."""
    cfg_path = '{}.{}'.format(prefix, config_name)
    m = importlib.import_module(name=cfg_path)
    m.SNAPSHOT_DIR += postfix
    if create:
        os.makedirs(m.SNAPSHOT_DIR, exist_ok=True)
    if copy:
        src = os.path.abspath(m.__file__)
        shutil.copy(src, os.path.join(m.SNAPSHOT_DIR, 'config.py'))
    return m


def lr_poly(base_lr, i_iter, max_iter, power):
    return base_lr * ((1 - float(i_iter) / max_iter) ** power)


def lr_warmup(base_lr, i_iter, warmup_iter):
    return base_lr * (float(i_iter) / warmup_iter)


def adjust_learning_rate(optimizer, i_iter, cfg):
    if i_iter < cfg.PREHEAT_STEPS:
        lr = lr_warmup(cfg.LEARNING_RATE, i_iter, cfg.PREHEAT_STEPS)
    else:
        lr = lr_poly(cfg.LEARNING_RATE, i_iter, cfg.NUM_STEPS, cfg.POWER)
    optimizer.param_groups[0]['lr'] = lr
    if len(optimizer.param_groups) > 1:
        optimizer.param_groups[1]['lr'] = lr * 10
    return lr


def loss_calc(pred, label, loss_fn, multi=False):
    """Cross entropy for segmentation (tools.py:240-260).  With the fused CrossEntropy of
    regda_amd.gast.balance the upsample and both heads run in one kernel."""
    if multi is True:
        if hasattr(loss_fn, 'forward_multi') and len(pred) == 2:
            return loss_fn.forward_multi(pred, label.long())
        loss, num = 0, 0
        for p in pred:
            if p.size()[-2:] != label.size()[-2:] and not hasattr(loss_fn, 'forward_multi'):
                p = tnf.interpolate(p, size=label.size()[-2:], mode='bilinear', align_corners=True)
            loss += loss_fn(p, label.long())
            num += 1
        return loss / num
    if pred.size()[-2:] != label.size()[-2:] and not hasattr(loss_fn, 'forward_multi'):
        pred = tnf.interpolate(pred, size=label.size()[-2:], mode='bilinear', align_corners=True)
    return loss_fn(pred, label.long())
