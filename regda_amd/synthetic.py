"""Synthetic SSL batches: 512x512 IRRG tiles, block-constant source labels, soft target
labels and random SAM-like region maps.  Used by bench.py, the CLI driver and the tests; generated on the
host with a seeded numpy/torch generator, then moved to the GPU once."""
import numpy as np
import torch

MEAN = (123.675, 116.28, 103.53)      # configs/ToPotsdam.py:51-52
STD = (58.395, 57.12, 57.375)


def region_maps(rng, b, h, w, kmin=20, kmax=250, min_area=1024):
    """Axis-aligned regions of area >= min_area painted over a zero background, later over earlier
    (mirrors area_thrshold=1024 and the overwrite order of local_region_homog.py:51-56)."""
    regs = np.zeros((b, h, w), np.int64)
    for i in range(b):
        k = int(rng.integers(kmin, kmax + 1))
        for r in range(1, k + 1):
            hh = int(rng.integers(max(8, h // 32), max(9, h // 4)))
            ww = max(int(np.ceil(min_area / hh)), int(rng.integers(max(8, w // 32), max(9, w // 4))))
            y0, x0 = int(rng.integers(0, max(1, h - hh))), int(rng.integers(0, max(1, w - ww)))
            regs[i, y0:y0 + hh, x0:x0 + ww] = r
    return regs


def make_batch(b=8, size=512, classes=6, seed=2333, device='cuda', with_soft=True):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor(MEAN).view(1, 3, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1)
    img_s = (torch.randint(0, 256, (b, 3, size, size), generator=g).float() - mean) / std
    img_t = ((torch.randint(0, 256, (b, 3, size, size), generator=g).float() - mean) / std).clamp(max=1.0)
    blk = 32
    nb = (size + blk - 1) // blk
    lab_s = torch.from_numpy(np.kron(rng.integers(-1, classes, size=(b, nb, nb)),
                                     np.ones((blk, blk), np.int64))[:, :size, :size].copy())
    out = dict(images_s=img_s, label_s=lab_s, images_t=img_t,
               regs_t=torch.from_numpy(region_maps(rng, b, size, size))[:, None])
    if with_soft:
        out['soft_t'] = torch.softmax(3.0 * torch.randn(b, classes, size, size, generator=g), dim=1)
    return {k: v.to(device) for k, v in out.items()}
