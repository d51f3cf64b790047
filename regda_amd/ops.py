"""Thin tensor-level wrappers over the C ABI (include/rgda_hip.h).

Everything here takes/returns CUDA(ROCm) torch tensors, enqueues on
`torch.cuda.current_stream()` and never synchronises.  PyTorch is plumbing
(device memory + streams); the arithmetic is in librgda_hip.so.
"""
import torch

from ._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


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


def label_refine(feat, protos, p1, p2, soft, temp=2.0, out=None, return_ws=False):
    _need_cuda(feat, protos, p1, p2, soft)
    feat, protos, p1, p2, soft = [t.contiguous().float() for t in (feat, protos, p1, p2, soft)]
    b, k, h, w = feat.shape
    c = protos.shape[0]
    H, W = soft.shape[-2:]
    assert p1.shape == (b, c, h, w) and p2.shape == (b, c, h, w) and soft.shape == (b, c, H, W)
    if out is None:
        out = torch.empty_like(soft)
    L = lib()
    ws = _ws(L.size('rgda_label_refine_workspace', b, c, h, w), feat.device)
    L.call('rgda_label_refine', feat.data_ptr(), protos.data_ptr(), p1.data_ptr(), p2.data_ptr(), soft.data_ptr(),
           out.data_ptr(), b, k, c, h, w, H, W, float(temp), ws.data_ptr(), ws.numel(), _stream())
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


def upsample_ce(p1, p2, label, ignore_label=-1, class_weight=None, want_grad=True):
    """-> (loss f32[1], g1, g2) ; g = d loss / d p (None if not want_grad)."""
    _need_cuda(p1, p2, label)
    p1, p2 = p1.contiguous().float(), p2.contiguous().float()
    label = label.contiguous()
    assert label.dtype == torch.int64
    b, c, h, w = p1.shape
    H, W = label.shape[-2:]
    loss = torch.empty(1, dtype=torch.float32, device=p1.device)
    g1 = torch.empty_like(p1) if want_grad else None
    g2 = torch.empty_like(p2) if want_grad else None
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
