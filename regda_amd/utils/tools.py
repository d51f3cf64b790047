"""loss_calc / learning-rate schedule / config import / sliding-window + TTA inference -- mirror of
regda/utils/tools.py:51-97,132-152,173-207,240-260."""
import importlib
import os
import shutil
from math import ceil

import torch
import torch.nn.functional as tnf

from .. import ops


def import_config(config_name, prefix='configs', copy=True, create=True, postfix=''):
    """Load the UPPER_CASE config module `<prefix>.<config_name>` (e.g. 'st.regda.2potsdam'), append `postfix` to its
    SNAPSHOT_DIR, optionally create that directory and drop a copy of the module there as config.py
    (same contract as tools.py:173-181)."""
    module = importlib.import_module(f'{prefix}.{config_name}')
    module.SNAPSHOT_DIR = module.SNAPSHOT_DIR + postfix
    if create:
        os.makedirs(module.SNAPSHOT_DIR, exist_ok=True)
    if copy:
        shutil.copy(os.path.abspath(module.__file__), os.path.join(module.SNAPSHOT_DIR, 'config.py'))
    return module


def learning_rate_at(i_iter, base_lr, warmup_iter, max_iter, power):
    """The one schedule of the st.regda.* runs (tools.py:184-207): linear warm-up from 0 over `warmup_iter`
    iterations, then polynomial decay base * (1 - it / max_iter) ** power.  Plain Python floats, like the reference."""
    it = float(i_iter)
    if i_iter < warmup_iter:
        return base_lr * (it / warmup_iter)
    return base_lr * (1 - it / max_iter) ** power


def lr_poly(base_lr, i_iter, max_iter, power):
    """Decay branch alone (tools.py:184-185)."""
    return learning_rate_at(i_iter, base_lr, 0, max_iter, power)


def lr_warmup(base_lr, i_iter, warmup_iter):
    """Warm-up branch alone (tools.py:188-189); not clamped past `warmup_iter`, like the reference."""
    return base_lr * (float(i_iter) / warmup_iter)


def adjust_learning_rate(optimizer, i_iter, cfg):
    """Set the lr of `optimizer` for iteration `i_iter` from cfg.{LEARNING_RATE, PREHEAT_STEPS, NUM_STEPS, POWER};
    a second parameter group, when present, runs at ten times the rate (tools.py:191-207).  Returns the lr."""
    lr = learning_rate_at(i_iter, cfg.LEARNING_RATE, cfg.PREHEAT_STEPS, cfg.NUM_STEPS, cfg.POWER)
    for group, mult in zip(optimizer.param_groups[:2], (1, 10)):
        group['lr'] = lr * mult
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


# ----------------------------------------------------------------------------- teacher inference
def pad_image(img, target_size):
    """"Pad an image up to the target size" -- literally what tools.py:51-58 computes:
    `tnf.pad(img, (0, 0, rows_missing, cols_missing))`, i.e. the ROW dimension gets rows_missing zeros on top and
    cols_missing at the bottom (negative = crop) and the width is untouched.  A no-op for windows that already have
    the tile size, which is every case where the image is at least as large as the tile."""
    rows_missing = target_size[0] - img.shape[2]
    cols_missing = target_size[1] - img.shape[3]
    if rows_missing == 0 and cols_missing == 0:
        return img
    return ops.pad_rows(img.contiguous().float(), rows_missing, cols_missing)


def tta_predict(model, img):
    """8-view test-time augmentation (tools.py:132-152; ttach 0.0.3 Compose([HorizontalFlip(), Rotate90(0/90/180/
    270)]), merge = mean).  The reference runs eight batch-1 forwards; the views are independent and the model is in
    eval mode (running BatchNorm statistics, per-sample InstanceNorm and pooling), so they go through the network
    as ONE batch of eight -- same arithmetic per view, one eighth of the launches.  Square tiles only when batched
    (rot90 of a non-square tile changes its shape): other shapes fall back to one forward per view."""
    if img.shape[0] != 1:
        raise ValueError('tta_predict averages over dim 0 like the reference: batch size must be 1')
    img = img.contiguous().float()
    views = [(f, k) for f in (False, True) for k in (0, 1, 2, 3)]
    _, c, h, w = img.shape
    out = None
    if h == w:
        batch = torch.empty(len(views), c, h, w, device=img.device)
        for i, (f, k) in enumerate(views):
            ops.dihedral(img, f, k, True, dst=batch[i:i + 1])
        pred = model(batch)
        for i, (f, k) in enumerate(views):
            out = ops.dihedral(pred[i:i + 1].contiguous(), f, (4 - k) % 4, False, dst=out, scale=1.0 / len(views),
                               accumulate=out is not None)
        return out
    for f, k in views:
        pred = model(ops.dihedral(img, f, k, True))
        out = ops.dihedral(pred.contiguous(), f, (4 - k) % 4, False, dst=out, scale=1.0 / len(views),
                           accumulate=out is not None)
    return out


def pre_slide(model, image, num_classes=7, tile_size=(512, 512), tta=False):
    """Sliding-window inference with overlap 1/2 (tools.py:61-97); returns the visit-count average (n, C, H, W)."""
    image = image.contiguous().float()
    n, c, H, W = image.shape
    stride = ceil(tile_size[0] * (1 - 1 / 2))
    tile_rows = int(ceil((H - tile_size[0]) / stride) + 1)
    tile_cols = int(ceil((W - tile_size[1]) / stride) + 1)
    full_probs = torch.zeros(n, num_classes, H, W, device=image.device)
    count = torch.zeros(n, 1, H, W, device=image.device)
    for row in range(tile_rows):
        for col in range(tile_cols):
            x1, y1 = int(col * stride), int(row * stride)
            x2, y2 = min(x1 + tile_size[1], W), min(y1 + tile_size[0], H)
            x1, y1 = max(int(x2 - tile_size[1]), 0), max(int(y2 - tile_size[0]), 0)
            h, w = y2 - y1, x2 - x1
            img = image if (h, w) == (H, W) else ops.window_crop(image, y1, x1, h, w, h, w)
            padded_img = pad_image(img, tile_size)
            padded = tta_predict(model, padded_img) if tta else model(padded_img)
            ops.window_accumulate(padded.contiguous(), full_probs, count, y1, x1, h, w)
    ops.window_normalise(full_probs, count)
    return full_probs
