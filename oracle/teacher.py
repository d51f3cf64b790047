"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of the teacher / pseudo-label
harness, SURVEY.md 8f rank 1.

  tta_predict           regda/utils/tools.py:132-152
  pre_slide, pad_image  regda/utils/tools.py:51-97
  soft-label resize     regda/gast/pseudo_generation.py:135 (tnf.interpolate(..., align_corners=True))

Third-party arithmetic on this path: `ttach==0.0.3` (requirement.txt:165), NOT vendored under /root/reference and
not installed here.  Its published behaviour for the two transforms the reference composes is restated in
`tta_views`: `Compose([HorizontalFlip(), Rotate90(angles=[0, 90, 180, 270])])` iterates the product of the
parameters in order (flip off/on outer, angle inner); `augment_image` applies HorizontalFlip (x.flip(3)) and then
Rotate90 (torch.rot90(x, angle // 90, (2, 3))); `deaugment_mask` undoes them in reverse order (rot90 by -k, then
the flip).  PARITY OF THE VIEW SET IS UNPINNED against the real ttach (golden vectors in tests/golden/tta.npz are
minted from the reference's own pre_slide / tta_predict driving this same restatement installed as `ttach`);
everything else on the path (window arithmetic, padding, averaging, resize) is pinned by that fixture.
"""
from math import ceil

import torch
import torch.nn.functional as F


def tta_views():
    """[(hflip, k)] in ttach's iteration order."""
    return [(f, k) for f in (False, True) for k in (0, 1, 2, 3)]


def augment(img, hflip, k):
    x = img.flip(3) if hflip else img
    return torch.rot90(x, k, (2, 3))


def deaugment(mask, hflip, k):
    x = torch.rot90(mask, -k, (2, 3))
    return x.flip(3) if hflip else x


def tta_predict(model, img):
    """tools.py:132-152: mean over the 8 de-augmented predictions (cat along dim 0, mean keepdim -> batch 1)."""
    xs = [deaugment(model(augment(img, f, k)), f, k) for f, k in tta_views()]
    return torch.mean(torch.cat(xs, 0), dim=0, keepdim=True)


def pad_image(img, target_size):
    """tools.py:51-58, as written: `tnf.pad(img, (0, 0, rows_missing, cols_missing))`.  torch's pad tuple starts at the
    LAST dimension, so this leaves W alone and pads H by rows_missing on TOP and cols_missing at the BOTTOM (a
    negative value crops).  A no-op whenever the window already has the tile size -- every production case, tiles
    are cut from images at least as large -- and reproduced literally otherwise."""
    rows_missing = target_size[0] - img.shape[2]
    cols_missing = target_size[1] - img.shape[3]
    return F.pad(img, (0, 0, rows_missing, cols_missing), 'constant', 0)


def windows(H, W, tile_size):
    """tools.py:62-77: the (y1, y2, x1, x2) windows of pre_slide, overlap 1/2."""
    stride = ceil(tile_size[0] * (1 - 1 / 2))
    rows = int(ceil((H - tile_size[0]) / stride) + 1)
    cols = int(ceil((W - tile_size[1]) / stride) + 1)
    out = []
    for row in range(rows):
        for col in range(cols):
            x1, y1 = int(col * stride), int(row * stride)
            x2, y2 = min(x1 + tile_size[1], W), min(y1 + tile_size[0], H)
            x1, y1 = max(int(x2 - tile_size[1]), 0), max(int(y2 - tile_size[0]), 0)
            out.append((y1, y2, x1, x2))
    return out


def pre_slide(model, image, num_classes=7, tile_size=(512, 512), tta=False):
    """tools.py:61-97."""
    n, _, H, W = image.shape
    full = torch.zeros(n, num_classes, H, W)
    count = torch.zeros(n, 1, H, W)
    for (y1, y2, x1, x2) in windows(H, W, tile_size):
        img = image[:, :, y1:y2, x1:x2]
        padded = pad_image(img, tile_size)
        pred = tta_predict(model, padded) if tta else model(padded)
        full[:, :, y1:y2, x1:x2] += pred[:, :, 0:img.shape[2], 0:img.shape[3]]
        count[:, :, y1:y2, x1:x2] += 1
    return full / count


def soft_label(cls, size):
    """pseudo_generation.py:135: the (C, h, w) fp32 tensor that is torch.save()d as <fname>.pt"""
    return F.interpolate(cls, size, mode='bilinear', align_corners=True).squeeze(dim=0)
