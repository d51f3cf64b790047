"""Thin tensor-level wrappers over the C ABI (include/rgda_hip.h).

Everything here takes/returns CUDA(ROCm) torch tensors, enqueues on
`torch.cuda.current_stream()` and never synchronises.  PyTorch is plumbing
(device memory + streams); the arithmetic is in librgda_hip.so.
"""
import ctypes

import torch

from ._lib import lib


_STREAM = None      # cached raw hipStream_t of the stream selected with use_stream() (saves ~8 us per launch)


def _stream():
    return _STREAM if _STREAM is not None else torch.cuda.current_stream().cuda_stream


class use_stream:
    """`with ops.use_stream(s):` = `with torch.cuda.stream(s):` + caches the raw handle for the launches inside."""

    def __init__(self, stream):
        self.stream = stream
        self.ctx = torch.cuda.stream(stream)

    def __enter__(self):
        global _STREAM
        self.prev = _STREAM
        self.ctx.__enter__()
        _STREAM = self.stream.cuda_stream
        return self.stream

    def __exit__(self, *a):
        global _STREAM
        _STREAM = self.prev
        return self.ctx.__exit__(*a)


def _p(t):
    return 0 if t is None else t.data_ptr()


STAT_FRAC_FWD, STAT_FRAC_BWD = 26, 40       # RGDA_STAT_FRAC_FWD / _BWD (include/rgda_hip.h, checked by tests/test_abi.py)


def new_stats(*shape, device='cuda'):
    """A zeroed per-channel accumulator, rgda_stat_t[...][RGDA_STAT_REPLICAS][2][C] (64-bit fixed point)."""
    return torch.zeros(*shape, dtype=torch.int64, device=device)


def stats_value(stats, backward=False):
    """Accumulators -> float64 values (sum over nothing: the caller still adds up the replica axis)."""
    return stats.double() * 2.0 ** -(STAT_FRAC_BWD if backward else STAT_FRAC_FWD)


def _stat(t):
    if t is not None and t.dtype != torch.int64:
        raise TypeError('BatchNorm statistic accumulators are int64 fixed point (ops.new_stats), got %s' % t.dtype)
    return _p(t)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('regda_amd kernels run on the GPU only (there is no CPU fallback); '
                               'got a CPU tensor')


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def pseudo_select(soft, cutoff_top=0.8, cutoff_low=0.6, ignore_label=-1, classmax_ws=None, check=True):
    """soft (b,c,h,w) f32 -> (b,h,w) int64.  `classmax_ws`: workspace already holding the per-class
    maxima (from label_refine).  check=True reads the range flag back (one host sync, like the
    reference's assert, pseudo_generation.py:71)."""
    _need_cuda(soft)
    assert soft.dim() == 4 and soft.dtype == torch.float32
    soft = soft.contiguous()
    b, c, h, w = soft.shape
    out = torch.empty((b, h, w), dtype=torch.int64, device=soft.device)
    if out.numel() == 0:
        return out
    L = lib()
    if classmax_ws is None:
        ws = _ws(L.size('rgda_pseudo_select_workspace', b, c), soft.device)
        ready = 0
    else:
        ws, ready = classmax_ws, 1
    L.call('rgda_pseudo_select', soft.data_ptr(), out.data_ptr(), b, c, h * w, cutoff_top, cutoff_low,
           ignore_label, ready, ws.data_ptr(), ws.numel(), _stream())
    if check and h * w > 0:
        flag = ws[b * c * 4: b * c * 4 + 4].view(torch.int32)
        assert int(flag.item()) == 0, 'pseudo_selection: probabilities must lie in [0, 1]'
    return out


def lrh(labels, regions, percent, class_num, ignore_label, max_regions=4096, check=True, ws=None):
    """Homogenizer.forward.  labels/regions (b,h,w) int64 -> (b,h,w) int64 (bit-exact)."""
    _need_cuda(labels, regions)
    assert labels.dim() == 3
    assert labels.dtype == torch.int64 and regions.dtype == torch.int64 and labels.shape == regions.shape
    labels, regions = labels.contiguous(), regions.contiguous()
    b, h, w = labels.shape
    out = torch.empty_like(labels)
    if labels.numel() == 0:
        return out
    L = lib()
    need = L.size('rgda_lrh_workspace', b, max_regions, class_num)
    if ws is None or ws.numel() < need:
        ws = _ws(need, labels.device)
    L.call('rgda_lrh', labels.data_ptr(), regions.data_ptr(), out.data_ptr(), b, h * w, class_num, ignore_label,
           float(percent), max_regions, ws.data_ptr(), ws.numel(), _stream())
    if check and h * w > 0:
        off = (b * max_regions * class_num + b * max_regions) * 4
        flag = int(ws[off:off + 4].view(torch.int32).item())
        if flag & 1:
            raise ValueError(f'Homogenizer: a region id is outside [0, max_regions={max_regions})')
        if flag & 2:
            raise ValueError('Homogenizer: a label is outside [0, class_num) and is not ignore_label')
    return out


def pseudo_lrh(soft, classmax_ws, regions, cutoff_top, cutoff_low, percent, class_num, ignore_label, max_regions=4096, ws=None):
    """LRH(pseudo_selection(soft), regions) in one pass (rgda_pseudo_lrh): soft (b,c,h,w) f32, classmax_ws = the workspace
    holding the per-class maxima (label_refine(return_ws=True) / pseudo_select), regions (b,h,w) int64 -> (b,h,w) int64,
    and the workspace (flag word at byte offset (b*R*C + b*R)*4, as for lrh)."""
    _need_cuda(soft, regions)
    soft, regions = soft.contiguous(), regions.contiguous()
    b, c, h, w = soft.shape
    assert regions.shape == (b, h, w) and regions.dtype == torch.int64 and soft.dtype == torch.float32
    out = torch.empty((b, h, w), dtype=torch.int64, device=soft.device)
    L = lib()
    need = L.size('rgda_pseudo_lrh_workspace', b, h * w, max_regions, class_num)
    if ws is None or ws.numel() < need:
        ws = _ws(need, soft.device)
    L.call('rgda_pseudo_lrh', soft.data_ptr(), classmax_ws.data_ptr(), regions.data_ptr(), out.data_ptr(), b, h * w, class_num,
           cutoff_top, cutoff_low, ignore_label, float(percent), max_regions, ws.data_ptr(), ws.numel(), _stream())
    return out, ws


def masks_to_regions(masks, areas, area_threshold=1024):
    """masks (K,H,W) uint8 / bool, areas (K,) int64 -> (H,W) int32 region map (local_region_homog.py:51-56)."""
    _need_cuda(masks, areas)
    K, H, W = masks.shape
    masks = masks.contiguous().to(torch.uint8) if masks.dtype != torch.uint8 else masks.contiguous()
    areas = areas.contiguous().to(torch.int64)
    assert areas.shape == (K,)
    out = torch.empty((H, W), dtype=torch.int32, device=masks.device)
    if H * W == 0:
        return out
    lib().call('rgda_masks_to_regions', masks.data_ptr() if K else 0, areas.data_ptr() if K else 0, out.data_ptr(), K, H * W,
               int(area_threshold), _stream())
    return out


def label_refine(feat, protos, p1, p2, soft, temp=2.0, out=None, return_ws=False, views=3):
    """views: bit 0 = prototype view, bit 1 = prediction view (3 = mode 'all'); inputs of a view that is off may be None."""
    pview, lview = bool(views & 1), bool(views & 2)
    assert views in (1, 2, 3)
    soft = soft.contiguous().float()
    _need_cuda(soft)
    b, c = soft.shape[:2]
    H, W = soft.shape[-2:]
    k = 4
    if pview:
        _need_cuda(feat, protos)
        feat, protos = feat.contiguous().float(), protos.contiguous().float()
        k = feat.shape[1]
        h, w = feat.shape[-2:]
        assert feat.shape[0] == b and protos.shape == (c, k)
    if lview:
        _need_cuda(p1, p2)
        p1, p2 = p1.contiguous().float(), p2.contiguous().float()
        h, w = p1.shape[-2:]
        assert p1.shape == (b, c, h, w) and p2.shape == (b, c, h, w)
        assert not pview or feat.shape[-2:] == (h, w)
    if out is None:
        out = torch.empty_like(soft)
    L = lib()
    ws = _ws(L.size('rgda_label_refine_workspace', b, c, h, w), soft.device)
    if views == 3:
        L.call('rgda_label_refine', feat.data_ptr(), protos.data_ptr(), p1.data_ptr(), p2.data_ptr(), soft.data_ptr(),
               out.data_ptr(), b, k, c, h, w, H, W, float(temp), ws.data_ptr(), ws.numel(), _stream())
    else:
        L.call('rgda_label_refine_views', _p(feat if pview else None), _p(protos if pview else None),
               _p(p1 if lview else None), _p(p2 if lview else None), soft.data_ptr(), out.data_ptr(), b, k, c, h, w, H, W,
               float(temp), views, ws.data_ptr(), ws.numel(), _stream())
    if return_ws:
        off = L.size('rgda_label_refine_classmax_offset', b, c, h, w)
        return out, ws[off:]
    return out


def label_refine_sup(feat, protos, p1, p2, soft, label_t_sup, temp=2.0, views=3, max_regions=65536, out=None, check=True,
                     return_ws=False):
    """label_refine with the superpixel view (rgda_label_refine_sup; alignment.py:238-258).  label_t_sup: int64 ids, b*H*W
    of them in any shape; views 3 = mode 'all', 0 = mode 's' (no other inputs needed).  check=True reads the range flag back
    (one host sync)."""
    pview, lview = bool(views & 1), bool(views & 2)
    assert views in (0, 1, 2, 3)
    soft = soft.contiguous().float()
    _need_cuda(soft, label_t_sup)
    b, c = soft.shape[:2]
    H, W = soft.shape[-2:]
    assert label_t_sup.dtype == torch.int64 and label_t_sup.numel() == b * H * W
    label_t_sup = label_t_sup.contiguous()
    k, h, w = 4, 1, 1
    if pview:
        _need_cuda(feat, protos)
        feat, protos = feat.contiguous().float(), protos.contiguous().float()
        k = feat.shape[1]
        h, w = feat.shape[-2:]
        assert feat.shape[0] == b and protos.shape == (c, k)
    if lview:
        _need_cuda(p1, p2)
        p1, p2 = p1.contiguous().float(), p2.contiguous().float()
        h, w = p1.shape[-2:]
        assert p1.shape == (b, c, h, w) and p2.shape == (b, c, h, w)
        assert not pview or feat.shape[-2:] == (h, w)
    if out is None:
        out = torch.empty_like(soft)
    L = lib()
    ws = _ws(L.size('rgda_label_refine_sup_workspace', b, c, h, w, max_regions), soft.device)
    L.call('rgda_label_refine_sup', _p(feat if pview else None), _p(protos if pview else None), _p(p1 if lview else None),
           _p(p2 if lview else None), soft.data_ptr(), label_t_sup.data_ptr(), out.data_ptr(), b, k, c, h, w, H, W,
           float(temp), views, max_regions, ws.data_ptr(), ws.numel(), _stream())
    if check:
        off = L.size('rgda_label_refine_sup_flag_offset', b, c, h, w, max_regions)
        if int(ws[off + 4:off + 8].view(torch.int32).item()):
            raise ValueError(f'label_refine: a superpixel id is outside [0, max_regions={max_regions})')
    if return_ws:
        off = L.size('rgda_label_refine_classmax_offset', b, c, h, w)
        return out, ws[off:]
    return out


def proto_update(feat, label, protos, scale=16, ignore_label=-1, min_ratio=0.75, decay=0.996):
    """In-place EMA update of `protos`; returns the downscaled label (b,1,h,w) int64."""
    _need_cuda(feat, label, protos)
    feat = feat.contiguous().float()
    label = label.contiguous()
    if label.dim() == 4:
        label = label.squeeze(1)
    assert protos.is_contiguous() and protos.dtype == torch.float32 and label.dtype == torch.int64
    b, k, h, w = feat.shape
    c = protos.shape[0]
    assert label.shape == (b, h * scale, w * scale), (label.shape, feat.shape)
    ds = torch.empty((b, 1, h, w), dtype=torch.int64, device=feat.device)
    L = lib()
    ws = _ws(L.size('rgda_proto_update_workspace', c, k), feat.device)
    L.call('rgda_proto_update', feat.data_ptr(), label.data_ptr(), protos.data_ptr(), ds.data_ptr(), b, k, c, h, w,
           scale, ignore_label, float(min_ratio), float(decay), ws.data_ptr(), ws.numel(), _stream())
    return ds


def proto_stats(feat, label, scale=16, ignore_label=-1, min_ratio=0.75, class_num=6, stats=None):
    """The sufficient statistics of update_prototype for data-parallel ranks (rgda_proto_stats): returns (stats, ds) --
    `stats` a float32 buffer whose first class_num * k + class_num elements are sums[c][k] and cnt[c] (what the ranks
    all-reduce), ds the downscaled label (b,1,h,w) int64."""
    _need_cuda(feat, label)
    feat = feat.contiguous().float()
    label = label.contiguous()
    if label.dim() == 4:
        label = label.squeeze(1)
    assert label.dtype == torch.int64
    b, k, h, w = feat.shape
    assert label.shape == (b, h * scale, w * scale), (label.shape, feat.shape)
    ds = torch.empty((b, 1, h, w), dtype=torch.int64, device=feat.device)
    L = lib()
    nbytes = L.size('rgda_proto_update_workspace', class_num, k)
    if stats is None:
        stats = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=feat.device)
    assert stats.is_contiguous() and stats.dtype == torch.float32 and stats.numel() * 4 >= nbytes
    L.call('rgda_proto_stats', feat.data_ptr(), label.data_ptr(), ds.data_ptr(), b, k, class_num, h, w, scale, ignore_label,
           float(min_ratio), stats.data_ptr(), stats.numel() * 4, _stream())
    return stats, ds


def proto_apply(protos, stats, decay=0.996):
    """protos <- EMA(protos, sums / (cnt + 1e-7), kept where cnt < 1) from (all-reduced) statistics (rgda_proto_apply)."""
    _need_cuda(protos, stats)
    assert protos.is_contiguous() and protos.dtype == torch.float32 and stats.dtype == torch.float32
    c, k = protos.shape
    assert stats.numel() >= c * k + c
    lib().call('rgda_proto_apply', protos.data_ptr(), stats.data_ptr(), c, k, float(decay), _stream())


def fill_zero(t):
    """t.zero_() as a kernel of this library (contiguous tensor, 16-byte aligned storage)."""
    assert t.is_contiguous()
    lib().call('rgda_fill_zero', t.data_ptr(), t.numel() * t.element_size(), _stream())


def copy_multi(pairs):
    """[(dst, src), ...] (<= 4 pairs of contiguous tensors of equal byte size, multiples of 16) in ONE launch."""
    n = len(pairs)
    for d, s_ in pairs:
        assert d.is_contiguous() and s_.is_contiguous() and d.numel() * d.element_size() == s_.numel() * s_.element_size()
    D = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    S = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in pairs])
    B = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    lib().call('rgda_copy_multi', n, ctypes.cast(D, ctypes.c_void_p), ctypes.cast(S, ctypes.c_void_p), ctypes.cast(B, ctypes.c_void_p),
               _stream())


def set_f32(t, value):
    lib().call('rgda_set_f32', t.data_ptr(), float(value), _stream())


def dropout_mask(out, p, seed):
    """out (f32, contiguous) = Dropout keep mask scaled by 1 / (1 - p), drawn from (seed, element index)."""
    lib().call('rgda_dropout_mask', out.data_ptr(), out.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream())


def upsample_ce(p1, p2, label, ignore_label=-1, class_weight=None, want_grad=True, g1=None, g2=None):
    """-> (loss f32[1], g1, g2) ; g = d loss / d p (None if not want_grad; written into the given g1 / g2 when passed)."""
    _need_cuda(p1, p2, label)
    p1, p2 = p1.contiguous().float(), p2.contiguous().float()
    label = label.contiguous()
    assert label.dtype == torch.int64
    b, c, h, w = p1.shape
    H, W = label.shape[-2:]
    loss = torch.empty(1, dtype=torch.float32, device=p1.device)
    if want_grad:
        g1 = torch.empty_like(p1) if g1 is None else g1
        g2 = torch.empty_like(p2) if g2 is None else g2
        assert g1.is_contiguous() and g2.is_contiguous() and g1.shape == p1.shape and g2.shape == p2.shape
    else:
        g1 = g2 = None
    L = lib()
    ws = _ws(L.size('rgda_upsample_ce_workspace', b, c, h, w, H, W), p1.device)
    cw = None if class_weight is None else class_weight.contiguous().float()
    L.call('rgda_upsample_ce', p1.data_ptr(), p2.data_ptr(), label.data_ptr(), _p(cw), loss.data_ptr(), _p(g1),
           _p(g2), b, c, h, w, H, W, ignore_label, ws.data_ptr(), ws.numel(), _stream())
    return loss, g1, g2


def teacher_probs(p1, p2, size):
    _need_cuda(p1, p2)
    p1, p2 = p1.contiguous().float(), p2.contiguous().float()
    b, c, h, w = p1.shape
    H, W = size
    out = torch.empty((b, c, H, W), dtype=torch.float32, device=p1.device)
    lib().call('rgda_teacher_probs', p1.data_ptr(), p2.data_ptr(), out.data_ptr(), b, c, h, w, H, W, _stream())
    return out


def class_count(label, class_num):
    _need_cuda(label)
    label = label.contiguous()
    cnt = torch.zeros(class_num, dtype=torch.int32, device=label.device)
    lib().call('rgda_class_count', label.data_ptr(), cnt.data_ptr(), label.numel(), class_num, _stream())
    return cnt


# --------------------------------------------------------------------------- conv stack
# Activations are torch.bfloat16 2-D tensors [N*H*W, ld] ("PxC"); a view with a column offset is
# expressed by passing a slice of the buffer (data_ptr of the slice) and its row stride.

def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def conv2d(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode=0, res=None, stats=None, stat_groups=1,
           res_mask=None):
    """x [N*H*W, Cin(view)], w bf16 [Cout, kh*kw, Cin] contiguous, y [N*Ho*Wo, Cout(view)].
    stats: rgda_stat_t (int64) [stat_groups][8][2][Cout] accumulators (zeroed by the caller; ops.new_stats)."""
    Cout, taps, Cin = w.shape
    assert taps == kh * kw and x.shape[1] == Cin and y.shape[1] == Cout
    lib().call('rgda_conv2d', x.data_ptr(), _ld(x), w.data_ptr(), y.data_ptr(), _ld(y), _p(res),
               _ld(res) if res is not None else 0, _p(res_mask), _stat(stats), stat_groups, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride,
               pad, dil, mode, _stream())


class _ConvDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ('x', 'wgt', 'y', 'res', 'res_relu_mask', 'stats')] + \
               [(k, ctypes.c_int) for k in ('ldx', 'ldy', 'ldres', 'stat_groups', 'N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'kh',
                                            'kw', 'stride', 'pad', 'dil', 'mode')]


def _conv_descs(items):
    arr = (_ConvDesc * len(items))()
    for d, it in zip(arr, items):
        x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil = it[:13]
        mode = it[13] if len(it) > 13 else 0
        res = it[14] if len(it) > 14 else None
        stats = it[15] if len(it) > 15 else None
        stat_groups = it[16] if len(it) > 16 else 1
        res_mask = it[17] if len(it) > 17 else None
        Cout, taps, Cin = w.shape
        assert taps == kh * kw and x.shape[1] == Cin and y.shape[1] == Cout
        d.x, d.wgt, d.y, d.res, d.res_relu_mask, d.stats = x.data_ptr(), w.data_ptr(), y.data_ptr(), _p(res), _p(res_mask), _stat(stats)
        d.ldx, d.ldy, d.ldres, d.stat_groups = _ld(x), _ld(y), (_ld(res) if res is not None else 0), stat_groups
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, Cout
        d.kh, d.kw, d.stride, d.pad, d.dil, d.mode = kh, kw, stride, pad, dil, mode
    return arr


def conv2d_grouped(items):
    """items: list of tuples holding conv2d's arguments (x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil[, mode[, res[, stats
    [, stat_groups[, res_mask]]]]]) -- INDEPENDENT convolutions; the small-tile ones share launches (rgda_conv2d_grouped)."""
    if not items:
        return
    arr = _conv_descs(items)
    lib().call('rgda_conv2d_grouped', ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())


def conv2d_grouped_launches(items):
    """Kernel launches conv2d_grouped(items) makes; raises where it would fail."""
    arr = _conv_descs(items)
    n = lib().size('rgda_conv2d_grouped_launches', ctypes.cast(arr, ctypes.c_void_p), len(items))
    if n < 0:
        raise ValueError('rgda_conv2d_grouped: status %d' % n)
    return n


def conv2d_bneval(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, rm, rv, gamma, beta, relu, res=None, eps=1e-5):
    """conv + inference-mode BN (+ residual + ReLU) in one kernel (the EMA teacher's units)."""
    Cout, taps, Cin = w.shape
    assert taps == kh * kw and x.shape[1] == Cin and y.shape[1] == Cout
    lib().call('rgda_conv2d_bneval', x.data_ptr(), _ld(x), w.data_ptr(), y.data_ptr(), _ld(y), _p(res),
               _ld(res) if res is not None else 0, rm.data_ptr(), rv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
               int(relu), N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, _stream())


def conv2d_bnbwd(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, res, sums, groups, bn_y, bn_x, bn_mi, relu,
                 nscale=None, rows_per_image=0, relu_mask=None, res_mask=None, bn_gamma=None, bn_beta=None):
    """conv2d whose epilogue also accumulates the BN-backward sums of the consumer of `y` (see rgda_conv2d_bnbwd).
    relu: False / True, or 2 = the ReLU sign recomputed from bn_x (needs bn_gamma, bn_beta)."""
    Cout, taps, Cin = w.shape
    assert taps == kh * kw and x.shape[1] == Cin and y.shape[1] == Cout
    lib().call('rgda_conv2d_bnbwd', x.data_ptr(), _ld(x), w.data_ptr(), y.data_ptr(), _ld(y), _p(res),
               _ld(res) if res is not None else 0, _p(res_mask), _stat(sums), groups, _p(bn_y), _ld(bn_y) if bn_y is not None else 0,
               _p(relu_mask), bn_x.data_ptr(), _ld(bn_x), bn_mi.data_ptr(), _p(nscale), rows_per_image, int(relu),
               _p(bn_gamma), _p(bn_beta), N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, mode, _stream())


class _BnOperand(ctypes.Structure):
    _fields_ = [('stats', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('mi', ctypes.c_void_p),
                ('running_mean', ctypes.c_void_p), ('running_var', ctypes.c_void_p), ('num_batches_tracked', ctypes.c_void_p),
                ('eps', ctypes.c_float), ('momentum', ctypes.c_float), ('groups', ctypes.c_int), ('relu', ctypes.c_int)]


def bn_operand(stats, gamma, beta, mi=None, rm=None, rv=None, nbt=None, groups=1, relu=True, eps=1e-5, momentum=0.1):
    """rgda_bn_operand: a BatchNorm (+ ReLU) that runs on its consumer's operand path.  stats: the producing convolution's
    accumulators [groups][8][2][C] (int64); mi (out) f32 [groups][2][C].  Returns a pointer object for conv2d_bnin /
    maxpool_fwd_bnin (host memory holding device pointers; the caller keeps the tensors alive)."""
    d = _BnOperand(_stat(stats), gamma.data_ptr(), beta.data_ptr(), _p(mi), _p(rm), _p(rv), _p(nbt), eps, momentum,
                   int(groups), int(bool(relu)))
    return ctypes.cast(ctypes.pointer(d), ctypes.c_void_p)


def conv2d_bnin_supported(M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, groups):
    """0 = not served, 1 = served, 2 = served and faster than the apply pass it replaces."""
    return lib().size('rgda_conv2d_bnin_supported', M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, groups)


def conv2d_bnin(bnop, x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, res=None, stats=None, stat_groups=1):
    """conv2d (forward) over  relu(BatchNorm(x))  of the producing convolution's RAW output x, applied on the operand path
    (rgda_conv2d_bnin; bnop from bn_operand()).  Raises ValueError where no kernel carries the transform."""
    Cout, taps, Cin = w.shape
    assert taps == kh * kw and x.shape[1] == Cin and y.shape[1] == Cout
    lib().call('rgda_conv2d_bnin', bnop, x.data_ptr(), _ld(x), w.data_ptr(), y.data_ptr(), _ld(y), _p(res),
               _ld(res) if res is not None else 0, _stat(stats), stat_groups, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad,
               dil, _stream())


def conv2d_wgrad(x, dy, dw, N, H, W, Ho, Wo, kh, kw, stride, pad, dil):
    """dw f32 [Cout, kh*kw, Cin] contiguous, accumulated."""
    Cout, taps, Cin = dw.shape
    assert dw.is_contiguous() and dw.dtype == torch.float32
    conv2d_wgrad_grouped([(x, dy, dw, N, H, W, Ho, Wo, kh, kw, stride, pad, dil)])


class _WgradDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('ldx', ctypes.c_int),
                ('lddy', ctypes.c_int)] + [(k, ctypes.c_int) for k in
                                           ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'kh', 'kw', 'stride', 'pad', 'dil',
                                            'lddw', 'co_split')]


_WGRAD_WS = {}          # (device index, raw stream) -> workspace of the weight-gradient launches issued on that stream
_WGRAD_WS_OLD = []      # outgrown workspaces stay allocated: a recorded plan (regda_amd/plan.py) may hold their address


def _wgrad_ws(nbytes, device):
    """The split-K workspace of the current stream (rgda_conv2d_wgrad: counters zeroed once, calls ordered on one
    stream share it).  Grown on demand; never freed."""
    key = (device.index, _stream())
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _WGRAD_WS_OLD.append(ws)
        ws = torch.empty(max(nbytes, 32 << 20), dtype=torch.uint8, device=device)
        ws[:65536].zero_()
        _WGRAD_WS[key] = ws
    return ws


def conv2d_wgrad_grouped(items):
    """items: list of (x, dy, dw, N, H, W, Ho, Wo, kh, kw, stride, pad, dil) -- the arguments of conv2d_wgrad.
    Layers that map to the same kernel share a launch (rgda_conv2d_wgrad_grouped)."""
    if not items:
        return
    arr = (_WgradDesc * len(items))()
    for d, (x, dy, dw, N, H, W, Ho, Wo, kh, kw, stride, pad, dil) in zip(arr, items):
        # dw: [Cout, taps, Cin], dense or a channel slice of a wider [Cout, taps, C] tensor (lddw = its row stride); for a
        # 1x1 layer also [S, T, Cin] with T > 1: the layer's S * T output rows are T stacked filters of S channels
        # ([t][s] order) and land channel-major in that tensor (co_split = S)
        S, T, Cin = dw.shape
        assert dw.dtype == torch.float32 and dw.stride(2) == 1 and dw.stride(0) == T * dw.stride(1)
        stacked = kh * kw == 1 and T > 1
        assert stacked or T == kh * kw
        Cout = S * T if stacked else S
        d.x, d.dy, d.dw, d.ldx, d.lddy = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ld(x), _ld(dy)
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, Cout
        d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
        d.lddw, d.co_split = dw.stride(1), (S if stacked else 0)
    L = lib()
    ap = ctypes.cast(arr, ctypes.c_void_p)
    ws = _wgrad_ws(L.size('rgda_conv2d_wgrad_workspace', ap, len(items)), items[0][0].device)
    L.call('rgda_conv2d_wgrad_grouped', ap, len(items), ws.data_ptr(), ws.numel(), _stream())


def stem_conv(img, wb, y, stats, N, H, W, Ho, Wo, stat_groups=1):
    """The 7x7 / stride-2 stem straight from the NCHW fp32 image (rgda_stem_conv; Wo % 64 == 0): y [N*Ho*Wo, 64(view)]
    bf16, wb the padded bf16 stem weights [64, 1, 192], stats as for conv2d."""
    lib().call('rgda_stem_conv', img.data_ptr(), wb.data_ptr(), y.data_ptr(), _ld(y), _stat(stats), stat_groups, N, H, W, Ho, Wo,
               _stream())


def stem_conv_bneval(img, wb, y, rm, rv, gamma, beta, relu, N, H, W, Ho, Wo, eps=1e-5):
    lib().call('rgda_stem_conv_bneval', img.data_ptr(), wb.data_ptr(), y.data_ptr(), _ld(y), rm.data_ptr(), rv.data_ptr(),
               gamma.data_ptr(), beta.data_ptr(), eps, int(relu), N, H, W, Ho, Wo, _stream())


def stem_im2col(img, col, N, H, W, Ho, Wo):
    lib().call('rgda_stem_im2col', img.data_ptr(), col.data_ptr(), N, H, W, Ho, Wo, col.shape[1], _stream())


_STEM_WS = {}           # (device index, raw stream) -> workspace of rgda_stem_wgrad on that stream (grown on demand, never freed)
_STEM_WS_OLD = []


def stem_wgrad(img, dy, dw, N, H, W, Ho, Wo):
    """The stem's weight gradient straight from the NCHW fp32 image (rgda_stem_wgrad; Wo % 64 == 0): dw f32 [64, 147]
    (contiguous, accumulated), dy bf16 [N*Ho*Wo, 64(view)]."""
    assert dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == 64 * 147 and img.is_contiguous()
    L = lib()
    nbytes = L.size('rgda_stem_wgrad_workspace', N, H, W)
    if nbytes == 0:
        raise ValueError('rgda_stem_wgrad: geometry not served (Wo % 64 != 0): use stem_im2col + conv2d_wgrad')
    key = (img.device.index, _stream())
    ws = _STEM_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _STEM_WS_OLD.append(ws)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        _STEM_WS[key] = ws
    L.call('rgda_stem_wgrad', img.data_ptr(), dy.data_ptr(), _ld(dy), dw.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, Ho, Wo,
           _stream())


def bn_stats(x, stats, M, C):
    lib().call('rgda_bn_stats', x.data_ptr(), _ld(x), _stat(stats), M, C, _stream())


def bn_finalize(stats, mi, rm, rv, nbt, M, C, eps=1e-5, momentum=0.1, groups=1):
    lib().call('rgda_bn_finalize', _stat(stats), mi.data_ptr(), _p(rm), _p(rv), _p(nbt), M, C, groups, eps, momentum,
               _stream())


def bn_apply(x, mi, gamma, beta, y, M, C, relu, res=None, nscale=None, rows_per_image=0, groups=1):
    lib().call('rgda_bn_apply', x.data_ptr(), _ld(x), mi.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(res),
               _ld(res) if res is not None else 0, _p(nscale), rows_per_image, y.data_ptr(), _ld(y), M, C,
               int(relu), groups, _stream())


def bn_train_apply(x, stats, mi, rm, rv, nbt, gamma, beta, y, M, C, relu, res=None, nscale=None, rows_per_image=0,
                   groups=1, eps=1e-5, momentum=0.1, relu_mask=None):
    lib().call('rgda_bn_train_apply', x.data_ptr(), _ld(x), _stat(stats), mi.data_ptr(), _p(rm), _p(rv), _p(nbt),
               gamma.data_ptr(), beta.data_ptr(), _p(res), _ld(res) if res is not None else 0, _p(nscale),
               rows_per_image, y.data_ptr(), _ld(y), _p(relu_mask), M, C, int(relu), groups, eps, momentum, _stream())


def bn_bwd_reduce(g, y, x, mi, sums, M, C, relu, nscale=None, rows_per_image=0, groups=1, relu_mask=None, gamma=None,
                  beta=None):
    """relu: False / True ([y > 0] from relu_mask or y), or 2 = the sign recomputed from x (needs gamma, beta)."""
    lib().call('rgda_bn_bwd_reduce', g.data_ptr(), _ld(g), _p(y), _ld(y) if y is not None else 0, _p(relu_mask),
               x.data_ptr(), _ld(x),
               mi.data_ptr(), _p(nscale), rows_per_image, _stat(sums), M, C, int(relu), _p(gamma), _p(beta), groups, _stream())


def bn_bwd_apply(g, y, x, mi, gamma, sums, dx, M, C, relu, gmask=None, dgamma=None, dbeta=None, nscale=None,
                 rows_per_image=0, groups=1, relu_mask=None, beta=None, act_out=None):
    """relu == 2: the ReLU sign from x (needs beta); act_out (bf16 [M, C(view)]) then receives relu(BatchNorm(x))."""
    lib().call('rgda_bn_bwd_apply', g.data_ptr(), _ld(g), _p(y), _ld(y) if y is not None else 0, _p(relu_mask),
               x.data_ptr(), _ld(x),
               mi.data_ptr(), gamma.data_ptr(), _p(nscale), rows_per_image, _stat(sums), dx.data_ptr(), _ld(dx),
               _p(gmask), _ld(gmask) if gmask is not None else 0, _p(dgamma), _p(dbeta), M, C, int(relu), _p(beta),
               _p(act_out), _ld(act_out) if act_out is not None else 0, groups, _stream())


class _BnSmallFwd(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ('x', 'y', 'relu_mask', 'gamma', 'beta', 'mi', 'running_mean', 'running_var',
                                               'num_batches_tracked')] + [('M', ctypes.c_int64)] + \
               [(k, ctypes.c_int) for k in ('ldx', 'ldy', 'C', 'groups', 'relu')] + [('eps', ctypes.c_float), ('momentum', ctypes.c_float)]


class _BnSmallBwd(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ('g', 'y', 'relu_mask', 'x', 'dx', 'mi', 'gamma', 'dgamma', 'dbeta')] + \
               [('M', ctypes.c_int64)] + [(k, ctypes.c_int) for k in ('ldg', 'ldy', 'ldx', 'lddx', 'C', 'groups', 'relu')]


BN_SMALL_MAX_ROWS = 320         # rows of one statistics group the small-map BatchNorm kernels take (include/rgda_hip.h)


def bn_train_small(items, eps=1e-5, momentum=0.1):
    """Train-mode BatchNorm (+ ReLU) of several SMALL maps in one launch (rgda_bn_train_small).  items: (x, y, mi, rm, rv, nbt,
    gamma, beta, M, C, relu, groups, relu_mask) -- the tensors of bn_train_apply, without the convolution's accumulators (the
    kernel sums the stored values itself)."""
    if not items:
        return
    arr = (_BnSmallFwd * len(items))()
    for d, (x, y, mi, rm, rv, nbt, gamma, beta, M, C, relu, groups, rmask) in zip(arr, items):
        d.x, d.y, d.relu_mask, d.gamma, d.beta, d.mi = x.data_ptr(), y.data_ptr(), _p(rmask), gamma.data_ptr(), beta.data_ptr(), mi.data_ptr()
        d.running_mean, d.running_var, d.num_batches_tracked = _p(rm), _p(rv), _p(nbt)
        d.M, d.ldx, d.ldy, d.C, d.groups, d.relu, d.eps, d.momentum = M, _ld(x), _ld(y), C, groups, int(bool(relu)), eps, momentum
    lib().call('rgda_bn_train_small', ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())


def bn_bwd_small(items):
    """BatchNorm backward (reduce + apply) of several SMALL maps in one launch (rgda_bn_bwd_small).  items: (g, y, x, mi, gamma,
    dx, dgamma, dbeta, M, C, relu, groups, relu_mask)."""
    if not items:
        return
    arr = (_BnSmallBwd * len(items))()
    for d, (g, y, x, mi, gamma, dx, dgamma, dbeta, M, C, relu, groups, rmask) in zip(arr, items):
        d.g, d.y, d.relu_mask, d.x, d.dx, d.mi, d.gamma = g.data_ptr(), _p(y), _p(rmask), x.data_ptr(), dx.data_ptr(), mi.data_ptr(), gamma.data_ptr()
        d.dgamma, d.dbeta = _p(dgamma), _p(dbeta)
        d.M, d.ldg, d.ldy, d.ldx, d.lddx = M, _ld(g), (_ld(y) if y is not None else 0), _ld(x), _ld(dx)
        d.C, d.groups, d.relu = C, groups, int(bool(relu))
    lib().call('rgda_bn_bwd_small', ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())


def maxpool_fwd(x, y, idx, N, H, W, C, Ho, Wo):
    lib().call('rgda_maxpool_fwd', x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, C, Ho, Wo, _stream())


def maxpool_fwd_bnin(bnop, x, y, idx, N, H, W, C, Ho, Wo):
    """MaxPool over relu(BatchNorm(x)) of the RAW stem convolution output (rgda_maxpool_fwd_bnin)."""
    lib().call('rgda_maxpool_fwd_bnin', bnop, x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, C, Ho, Wo, _stream())


def maxpool_bwd(gy, idx, gx, N, H, W, C, Ho, Wo):
    lib().call('rgda_maxpool_bwd', gy.data_ptr(), idx.data_ptr(), gx.data_ptr(), N, H, W, C, Ho, Wo, _stream())


def instnorm_fwd(x, y0, y1, feat, mi, N, HW, C, eps=1e-5):
    ldy = _ld(y0) if y0 is not None else (_ld(y1) if y1 is not None else 0)
    lib().call('rgda_instnorm_fwd', x.data_ptr(), _ld(x), _p(y0), _p(y1), ldy, _p(feat), mi.data_ptr(), N, HW, C, eps,
               _stream())


def instnorm_bwd(ga, gb, gc, x, mi, dx, N, HW, C):
    ldg = _ld(ga) if ga is not None else (_ld(gb) if gb is not None else 0)
    lib().call('rgda_instnorm_bwd', _p(ga), _p(gb), ldg, _p(gc), _ld(gc) if gc is not None else 0, x.data_ptr(),
               _ld(x), mi.data_ptr(), dx.data_ptr(), _ld(dx), N, HW, C, _stream())


def spatial_mix_multi(ins, mats, out, N, I, C):
    """out[n][i][:] = sum_q mats[q][i,:] @ ins[q][n]  (q <= 4 sources, bf16 out)."""
    import ctypes
    n = len(ins)
    P = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ins])
    L = (ctypes.c_int * n)(*[_ld(t) for t in ins])
    Mt = (ctypes.c_void_p * n)(*[m.data_ptr() for m in mats])
    J = (ctypes.c_int * n)(*[m.shape[1] for m in mats])
    for m in mats:
        assert m.shape[0] == I and m.is_contiguous() and m.dtype == torch.float32
    lib().call('rgda_spatial_mix_multi', n, P, L, Mt, J, out.data_ptr(), _ld(out), N, I, C, _stream())


def spatial_mix(inp, Mx, out, N, I, J, C, accumulate=False):
    assert Mx.shape == (I, J) and Mx.is_contiguous() and Mx.dtype == torch.float32
    lib().call('rgda_spatial_mix', inp.data_ptr(), _ld(inp), Mx.data_ptr(), out.data_ptr(), _ld(out), N, I, J, C,
               int(accumulate), int(out.dtype == torch.float32), _stream())


def group_mix(inp, W, out, G, I, J, C):
    """out[g][i][:] = sum_j W[i][j] * inp[g][j][:] for G groups of J consecutive rows (bf16 or f32 tensors)."""
    assert W.shape == (I, J) and W.is_contiguous() and W.dtype == torch.float32
    assert inp.shape[0] == G * J and out.shape[0] == G * I
    lib().call('rgda_group_mix', inp.data_ptr(), _ld(inp), int(inp.dtype == torch.float32), W.data_ptr(),
               out.data_ptr(), _ld(out), int(out.dtype == torch.float32), G, I, J, C, _stream())


def sparse_mix(ins, csr, outs, N, C):
    """outs[q][n][i][:] = sum_k vals[k] * ins[cols[k] >> 24][n][cols[k] & 0xffffff][:] over the CSR rows, which are
    split in order over the (<= 4) image-major output tensors.  csr = (rowptr int32, cols int32, vals f32) on the
    device; ins: <= 4 tensors of one dtype; outs: a tensor or a list of tensors of one dtype."""
    import ctypes
    rowptr, cols, vals = csr
    outs = [outs] if torch.is_tensor(outs) else list(outs)
    assert rowptr.dtype == torch.int32 and cols.dtype == torch.int32 and vals.dtype == torch.float32
    rows = [o.shape[0] // N for o in outs]
    assert rowptr.numel() == sum(rows) + 1 and all(o.shape[0] == N * r for o, r in zip(outs, rows))
    n, m = len(ins), len(outs)
    P = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ins])
    L = (ctypes.c_int * n)(*[_ld(t) for t in ins])
    J = (ctypes.c_int * n)(*[t.shape[0] // N for t in ins])
    O = (ctypes.c_void_p * m)(*[t.data_ptr() for t in outs])
    OL = (ctypes.c_int * m)(*[_ld(t) for t in outs])
    OR = (ctypes.c_int * m)(*rows)
    lib().call('rgda_sparse_mix', n, P, L, J, int(ins[0].dtype == torch.float32), rowptr.data_ptr(), cols.data_ptr(),
               vals.data_ptr(), m, O, OL, OR, int(outs[0].dtype == torch.float32), N, C, _stream())


def classifier_fwd(hidden, w, bias, logits, N, HW, C, ncls):
    lib().call('rgda_classifier_fwd', hidden.data_ptr(), _ld(hidden), w.data_ptr(), bias.data_ptr(), logits.data_ptr(),
               N, HW, C, ncls, _stream())


def classifier_bwd(hidden, w, glogits, dhidden, dw, db, N, HW, C, ncls):
    L = lib()
    ws = _ws(L.size('rgda_classifier_bwd_workspace', N * HW, C, ncls), hidden.device)
    L.call('rgda_classifier_bwd', hidden.data_ptr(), _ld(hidden), w.data_ptr(), glogits.data_ptr(),
           dhidden.data_ptr(), _ld(dhidden), dw.data_ptr(), db.data_ptr(), N, HW, C, ncls, ws.data_ptr(), ws.numel(),
           _stream())


def sumsq(g, out, ws):
    lib().call('rgda_sumsq', g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), _stream())


def sgd_step(p, g, v, shadow, p_bf16, gnorm_sq, lr_dev, momentum, weight_decay, max_norm, gscale, ema_decay,
             first_step, shadow_bf16=None):
    lib().call('rgda_sgd_step', p.data_ptr(), g.data_ptr(), v.data_ptr(), _p(shadow), _p(p_bf16), _p(shadow_bf16),
               gnorm_sq.data_ptr(),
               lr_dev.data_ptr(), p.numel(), momentum, weight_decay, max_norm, gscale, ema_decay, int(first_step),
               _stream())


def weight_transpose_bf16(w, wt, Co, T, Ci):
    lib().call('rgda_weight_transpose_bf16', w.data_ptr(), wt.data_ptr(), Co, T, Ci, _stream())


def weight_transpose_batched(table, n, total_blocks):
    lib().call('rgda_weight_transpose_batched', table.data_ptr(), n, total_blocks, _stream())


def cast_bf16(src, dst):
    lib().call('rgda_cast_bf16', src.data_ptr(), dst.data_ptr(), src.numel(), _stream())


def cast_f32(src, dst):
    lib().call('rgda_cast_f32', src.data_ptr(), dst.data_ptr(), dst.numel(), _stream())


def ddp_accumulate_bf16(recv, world, out):
    lib().call('rgda_ddp_accumulate_bf16', recv.data_ptr(), world, out.data_ptr(), out.numel(), _stream())


def pad_cast_bf16(src, dst, R, K, Kp):
    lib().call('rgda_pad_cast_bf16', src.data_ptr(), dst.data_ptr(), R, K, Kp, _stream())


def unpad_acc_f32(src, dst, R, K, Kp):
    lib().call('rgda_unpad_acc_f32', src.data_ptr(), dst.data_ptr(), R, K, Kp, _stream())


def add_bf16(a, b, out, M, C):
    lib().call('rgda_add_bf16', a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), out.data_ptr(), _ld(out), M, C, _stream())


# ----------------------------------------------------------------------------- teacher / pseudo-label harness
def dihedral(src, hflip, k, flip_first, dst=None, scale=1.0, accumulate=False):
    """One TTA view (rgda_dihedral_nchw): fp32 NCHW; returns dst (allocated when None)."""
    _need_cuda(src)
    n, c, h, w = src.shape
    k = k % 4
    ho, wo = (w, h) if (k & 1) else (h, w)
    if dst is None:
        assert not accumulate
        dst = torch.empty(n, c, ho, wo, device=src.device)
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32
    assert tuple(dst.shape) == (n, c, ho, wo)
    lib().call('rgda_dihedral_nchw', src.data_ptr(), dst.data_ptr(), n, c, h, w, int(bool(hflip)), k, int(bool(flip_first)),
               float(scale), int(bool(accumulate)), _stream())
    return dst


def window_crop(full, y1, x1, h, w, Th, Tw):
    n, c, Hf, Wf = full.shape
    tile = torch.empty(n, c, Th, Tw, device=full.device)
    lib().call('rgda_window_crop', full.data_ptr(), tile.data_ptr(), n, c, Hf, Wf, y1, x1, h, w, Th, Tw, _stream())
    return tile


def window_accumulate(tile, full, count, y1, x1, h, w):
    n, c, Hf, Wf = full.shape
    lib().call('rgda_window_accumulate', tile.data_ptr(), full.data_ptr(), count.data_ptr(), n, c, Hf, Wf, y1, x1, h, w,
               tile.shape[2], tile.shape[3], _stream())


def window_normalise(full, count):
    n, c, Hf, Wf = full.shape
    lib().call('rgda_window_normalise', full.data_ptr(), count.data_ptr(), n, c, Hf, Wf, _stream())


def resize_bilinear_ac(src, size):
    n, c, h, w = src.shape
    dst = torch.empty(n, c, size[0], size[1], device=src.device)
    lib().call('rgda_resize_bilinear_ac', src.contiguous().data_ptr(), dst.data_ptr(), n, c, h, w, size[0], size[1], _stream())
    return dst


def pad_rows(src, top, bottom):
    n, c, h, w = src.shape
    dst = torch.empty(n, c, h + top + bottom, w, device=src.device)
    lib().call('rgda_pad_rows_nchw', src.contiguous().data_ptr(), dst.data_ptr(), n, c, h, w, top, bottom, _stream())
    return dst


# ----------------------------------------------------------------------------- evaluation path
def argmax_nchw(probs):
    n, c, h, w = probs.shape
    out = torch.empty(n, h, w, dtype=torch.int64, device=probs.device)
    lib().call('rgda_argmax_nchw', probs.contiguous().float().data_ptr(), out.data_ptr(), n, c, h * w, _stream())
    return out


def confusion_accumulate(y_true, y_pred, cm, flag):
    """cm int64 [C, C] += counts over the pixels with y_true >= 0 (row = true class)."""
    assert cm.dtype == torch.int64 and cm.is_contiguous() and cm.shape[0] == cm.shape[1]
    yt, yp = y_true.contiguous().view(-1), y_pred.contiguous().view(-1)
    assert yt.dtype == yp.dtype == torch.int64 and yt.numel() == yp.numel()
    lib().call('rgda_confusion_accumulate', yt.data_ptr(), yp.data_ptr(), cm.data_ptr(), flag.data_ptr(), yt.numel(),
               cm.shape[0], _stream())


# ----------------------------------------------------------------------------- stage 2 ("align")
def pcl_loss(feat, labels, protos, temperature=8.0, ignore_label=-1, weight=1.0, loss=None, dfeat=None, accumulate=False):
    """PrototypeContrastiveLoss forward (+ gradient w.r.t. feat into `dfeat` bf16 [b*h*w, K] when given).
    Returns the (accumulating) fp32 loss tensor."""
    _need_cuda(feat, labels, protos)
    feat = feat.contiguous().float()
    b, K, h, w = feat.shape
    labels = labels.contiguous().view(b, h, w)
    assert labels.dtype == torch.int64 and protos.is_contiguous() and protos.dtype == torch.float32
    C = protos.shape[0]
    if loss is None:
        loss = torch.zeros(1, device=feat.device)
    L = lib()
    ws = _ws(L.size('rgda_pcl_loss_workspace', C, K), feat.device)
    L.call('rgda_pcl_loss', feat.data_ptr(), labels.data_ptr(), protos.data_ptr(), loss.data_ptr(), _p(dfeat),
           _ld(dfeat) if dfeat is not None else 0, int(bool(accumulate)), b, K, C, h, w, ignore_label, float(temperature),
           float(weight), ws.data_ptr(), ws.numel(), _stream())
    return loss


# ---------------------------------------------------------------- ASPP head (Classifier_Module)
def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def aspp_gather(z, biases, out1, out2, N, h, w, C, dils):
    """z bf16 [N*h*w, >= 72*C]; biases: eight f32 [C] tensors ([head][dilation]); out1/out2 f32 (N,C,h,w)."""
    import ctypes
    assert len(biases) == 8 and len(dils) == 4
    P = _ptr_array(biases)
    D = (ctypes.c_int * 4)(*[int(d) for d in dils])
    lib().call('rgda_aspp_gather', z.data_ptr(), _ld(z), ctypes.cast(P, ctypes.c_void_p), out1.data_ptr(),
               out2.data_ptr(), N, h, w, C, ctypes.cast(D, ctypes.c_void_p), _stream())


def aspp_scatter(g1, g2, dz, dbiases, N, h, w, C, dils):
    """g1/g2 f32 (N,C,h,w) contiguous -> dz bf16 [N*h*w, zc]; dbiases: eight f32 [C] tensors, accumulated."""
    import ctypes
    assert len(dbiases) == 8 and len(dils) == 4 and g1.is_contiguous() and g2.is_contiguous()
    P = _ptr_array(dbiases)
    D = (ctypes.c_int * 4)(*[int(d) for d in dils])
    lib().call('rgda_aspp_scatter', g1.data_ptr(), g2.data_ptr(), dz.data_ptr(), _ld(dz), dz.shape[1],
               ctypes.cast(P, ctypes.c_void_p), N, h, w, C, ctypes.cast(D, ctypes.c_void_p), _stream())
