"""Integer label arithmetic of the SSL step, restated in numpy.  TEST INFRASTRUCTURE ONLY.

Bit-exact contracts:
  pseudo_selection  <- regda/gast/pseudo_generation.py:59-93
  homogenize (LRH)  <- regda/utils/local_region_homog.py:107-152
                       (+ third-party torch_scatter.scatter(reduce='sum'), :140)
  downscale_label   <- regda/gast/alignment.py:456-481
"""
import numpy as np


def pseudo_selection(mask, cutoff_top=0.8, cutoff_low=0.6, ignore_label=-1):
    """soft (b,c,h,w) f32 -> hard (b,h,w) int64.

    pseudo_generation.py:70-88: per image, per class threshold
    t = max(f32(max_hw p) * cutoff_top, cutoff_low) computed in fp32; a pixel
    passes a class iff p > t (strict); it is labelled with that class iff
    exactly one class passes, else ignore_label.
    """
    mask = np.asarray(mask, dtype=np.float32)
    assert mask.ndim == 4
    if mask.size:
        assert mask.max() <= 1 and mask.min() >= 0          # :71
    b, c, h, w = mask.shape
    m = mask.reshape(b, c, -1)
    mx = m.max(-1, keepdims=True) if m.shape[-1] else np.zeros((b, c, 1), np.float32)
    thr = (mx * np.float32(cutoff_top)).astype(np.float32)  # mask_max *= cutoff_top (:77)
    thr = np.maximum(thr, np.float32(cutoff_low))           # .max(lowest) (:80-81)
    g = m > thr                                             # :83
    cnt = g.sum(1)                                          # :85
    lab = g.argmax(1).astype(np.int64)                      # first True (:87)
    lab[cnt != 1] = ignore_label                            # :88
    return lab.reshape(b, h, w)


def region_histogram(labels, regions, class_num, ignore_label=-1):
    """(b,h,w) labels, (b,h,w) region ids >= 0 -> (b, R, class_num) int64 counts,
    R = regions.max()+1 (what torch_scatter sizes its output to,
    local_region_homog.py:140)."""
    labels = np.asarray(labels, dtype=np.int64)
    regions = np.asarray(regions, dtype=np.int64)
    b = labels.shape[0]
    R = int(regions.max()) + 1 if regions.size else 1
    hist = np.zeros((b, R, class_num), dtype=np.int64)
    for i in range(b):
        lab = labels[i].reshape(-1)
        reg = regions[i].reshape(-1)
        keep = lab != ignore_label                          # one-hot drops column C (:119-121)
        np.add.at(hist[i], (reg[keep], lab[keep]), 1)
    return hist


def homogenize(pseudo_labels, regions, percent=0.5, class_num=6, ignore_label=-1):
    """Local Region Homogenizing.  local_region_homog.py:125-152.

    Per (image, region): n = sum_c hist, m = max_c hist, id = FIRST argmax;
    ratio = f32(m) / (f32(n) + 1e-5f) in fp32 (:143, int64 -> float promotion
    gives float32); id = ignore if ratio < percent (:144); every pixel takes its
    region's id (:147); region 0 -> ignore (:149); pixels whose id is ignore
    keep their original label (:151).
    """
    pseudo_labels = np.asarray(pseudo_labels, dtype=np.int64)
    regions = np.asarray(regions, dtype=np.int64)
    assert pseudo_labels.ndim == 3                           # :133
    hist = region_histogram(pseudo_labels, regions, class_num, ignore_label)
    n = hist.sum(-1)
    m = hist.max(-1)
    idx = hist.argmax(-1).astype(np.int64)
    ratio = m.astype(np.float32) / (n.astype(np.float32) + np.float32(1e-5))
    # `percent` is a python float; torch compares the f32 tensor against it as a
    # double scalar promoted to f32 semantics: (f32 < python_scalar) casts the
    # scalar to f32.  0.5 / 0.9 etc. are compared in f32.
    idx[ratio < np.float32(percent)] = ignore_label
    b = pseudo_labels.shape[0]
    out = np.empty_like(pseudo_labels)
    for i in range(b):
        o = idx[i][regions[i]]
        o[regions[i] == 0] = ignore_label
        out[i] = np.where(o == ignore_label, pseudo_labels[i], o)
    return out


def downscale_label(label, scale_factor=16, n_classes=6, ignore_label=-1, min_ratio=0.75):
    """(b,H,W) int64 -> (b,1,H/s,W/s) int64.  alignment.py:466-481.

    one-hot over C+1 (ignore -> C), avg_pool s x s (f32 mean = count/s^2),
    max ratio + first argmax; class C or ratio < min_ratio -> ignore.
    """
    label = np.asarray(label, dtype=np.int64)
    if label.ndim == 4:
        label = label[:, 0]
    b, H, W = label.shape
    s = scale_factor
    th, tw = H // s, W // s
    lab = np.where(label == ignore_label, n_classes, label)[:, :th * s, :tw * s]
    blocks = lab.reshape(b, th, s, tw, s).transpose(0, 1, 3, 2, 4).reshape(b, th, tw, s * s)
    cnt = np.zeros((b, th, tw, n_classes + 1), dtype=np.int64)
    for c in range(n_classes + 1):
        cnt[..., c] = (blocks == c).sum(-1)
    # avg_pool2d of a 0/1 float tensor: sum (exact in f32) * (1/s^2); for s=16
    # the division is exact, so the ratio test equals an integer test.
    ratio = cnt.astype(np.float32) / np.float32(s * s)
    mx = ratio.max(-1)
    out = ratio.argmax(-1).astype(np.int64)
    out[out == n_classes] = ignore_label
    out[mx < np.float32(min_ratio)] = ignore_label
    return out[:, None]
