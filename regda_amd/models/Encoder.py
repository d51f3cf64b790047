"""Deeplabv2 (ResNet encoder, output stride 16, InstanceNorm, two PPMBilinear heads) -- the host-side
mirror of regda/models/Encoder.py:87-186 + regda/resnet.py:43-207 + regda/_resnets.py:72-112 for the
configuration every st.regda.* entry point builds (tools/train_ssl_reg.py:94-111):
    multi_layer=True, cascade=False, use_ppm=True, is_ins_norm=True,
and for its use_ppm=False sibling with two ASPP heads (Classifier_Module, Encoder.py:68-84,111-114).

Same constructor (a config dict), same train()/eval() outputs ((x1, x2, feat) / class probabilities),
same state_dict layout (688 keys for ResNet-101) as the reference, so checkpoints interchange.

Execution is NOT torch.nn: the network is a static plan of HIP kernels from librgda_hip.so over
pixel-major bf16 activations ("PxC": [N*H*W][C], channels contiguous -- the layout whose implicit-GEMM
operands are K-contiguous for bf16 MFMA; NCHW only at the API boundary), with an explicit backward
plan.  Parameters live in ONE flat fp32 buffer (conv weights physically [Cout][kh][kw][Cin]; the
nn.Parameter objects are zero-copy views with the reference's logical [Cout,Cin,kh,kw] shape), with a
bf16 mirror for the forward pass, a transposed bf16 copy for the data-gradient pass and ONE flat fp32
gradient buffer (what the RCCL all-reduce and the fused SGD kernel work on).
"""
import contextlib
import math

import os

import torch
import torch.nn as nn

from .. import ops
from .. import plan

BF = torch.bfloat16
# blocks per stage of the topologies the reference builds (regda/_resnets.py:232-262); tests register a shallow one of
# their own (tests/conftest.py) -- the table is read when a model is constructed
LAYERS = {'resnet101': (3, 4, 23, 3), 'resnet50': (3, 4, 6, 3)}
POOL_SCALES = (1, 2, 3, 6)
NREP = 8                # RGDA_STAT_REPLICAS (include/rgda_hip.h)
LAYOUT_TILE = 64        # RGDA_LAYOUT_TILE


def _layout_blocks(co, ci, taps):
    """Blocks one table row of rgda_weight_transpose_batched owns."""
    return -(-ci // LAYOUT_TILE) * -(-co // LAYOUT_TILE) * taps

ASPP_DILATIONS = (6, 12, 18, 24)     # dilation_series = padding_series of every Classifier_Module (Encoder.py:101-114)
STEM_KP = 192           # 7*7*3 = 147 im2col columns, zero padded to a multiple of 64


# ----------------------------------------------------------------------------- spatial matrices
def pool_matrix(H, W, s):
    """AdaptiveAvgPool2d(s) on an HxW map as a dense [s*s, H*W] matrix (bins floor/ceil, Encoder.py:16-18)."""
    P = torch.zeros(s * s, H * W)
    for i in range(s):
        h0, h1 = (i * H) // s, -((-(i + 1) * H) // s)
        for j in range(s):
            w0, w1 = (j * W) // s, -((-(j + 1) * W) // s)
            v = 1.0 / ((h1 - h0) * (w1 - w0))
            for y in range(h0, h1):
                P[i * s + j, y * W + w0: y * W + w1] = v
    return P


def _src_index(dst, scale):
    src = scale * (dst + 0.5) - 0.5          # align_corners=False
    return max(src, 0.0)


def upsample_matrix(h, w, H, W):
    """F.interpolate(mode='bilinear', align_corners=False) from hxw to HxW as a dense [H*W, h*w] matrix
    (Encoder.py:48-51)."""
    import numpy as np
    U = torch.zeros(H * W, h * w)
    sh, sw = np.float32(h) / np.float32(H), np.float32(w) / np.float32(W)

    def taps(dst, scale, n):
        src = np.float32(scale) * (np.float32(dst) + np.float32(0.5)) - np.float32(0.5)
        src = max(src, np.float32(0))
        i0 = int(src)
        i1 = i0 + (1 if i0 < n - 1 else 0)
        l1 = np.float32(src) - np.float32(i0)
        return i0, i1, float(np.float32(1) - l1), float(l1)
    for Y in range(H):
        y0, y1, ly0, ly1 = taps(Y, sh, h)
        for X in range(W):
            x0, x1, lx0, lx1 = taps(X, sw, w)
            r = Y * W + X
            U[r, y0 * w + x0] += ly0 * lx0
            U[r, y0 * w + x1] += ly0 * lx1
            U[r, y1 * w + x0] += ly1 * lx0
            U[r, y1 * w + x1] += ly1 * lx1
    return U


def ppm_tap_matrix(h, w, s):
    """The 3x3 / pad-1 head conv applied to a bilinearly upsampled s x s map q is linear in q, and the channel
    mixing commutes with the spatial mixing:  conv(U q)[p] = sum_tap sum_j U[p + d_tap][j] * (W_tap q[j]).
    Returns V [h*w, 9*s*s] with V[p][j*9 + tap] = U[p + d_tap][j] (0 where p + d_tap is padding), so that
    conv(U q) = V @ Z with Z[j*9 + tap] = W_tap q[j] computed at LOW resolution (Encoder.py:30-51 restated)."""
    U = upsample_matrix(s, s, h, w).view(h, w, s * s)
    V = torch.zeros(h, w, s * s, 9)
    for kh in range(3):
        for kw in range(3):
            dy, dx = kh - 1, kw - 1
            y0, y1 = max(0, -dy), min(h, h - dy)
            x0, x1 = max(0, -dx), min(w, w - dx)
            V[y0:y1, x0:x1, :, kh * 3 + kw] = U[y0 + dy:y1 + dy, x0 + dx:x1 + dx, :]
    return V.reshape(h * w, s * s * 9).contiguous()


def upsample_taps_1d(n_in, n_out):
    """One axis of upsample_matrix (bilinear, align_corners=False): [n_out, n_in], same float32 arithmetic."""
    import numpy as np
    U = torch.zeros(n_out, n_in)
    sc = np.float32(n_in) / np.float32(n_out)
    for d in range(n_out):
        src = max(np.float32(sc) * (np.float32(d) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = int(src)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        l1 = np.float32(src) - np.float32(i0)
        U[d, i0] += float(np.float32(1) - l1)
        U[d, i1] += float(l1)
    return U


def pool_taps_1d(n_in, s):
    """One axis of pool_matrix (AdaptiveAvgPool bins floor/ceil): [s, n_in]."""
    P = torch.zeros(s, n_in)
    for i in range(s):
        a, b = (i * n_in) // s, -((-(i + 1) * n_in) // s)
        P[i, a:b] = 1.0 / (b - a)
    return P


def pool_factored_maps(h, w):
    """pool_matrix is separable too: P_s[(jy,jx)][(y,x)] = Py_s[jy][y] * Px_s[jx][x].  With r = (s, jx) indexing
    R = sum(s) intermediate rows per image row:  Px [R, w] (the x direction, rgda_group_mix) and the y direction as
    dense Py_s-expanded matrices Qy_s [s*s, h*R], Qy_s[(jy,jx)][y*R + off_s + jx] = Py_s[jy][y] (CSR operands)."""
    R = sum(POOL_SCALES)
    Px = torch.zeros(R, w)
    Qy = []
    off = 0
    for s in POOL_SCALES:
        py, px = pool_taps_1d(h, s), pool_taps_1d(w, s)
        Px[off:off + s] = px
        Q = torch.zeros(s * s, h * R)
        for jy in range(s):
            for jx in range(s):
                Q[jy * s + jx, torch.arange(h) * R + off + jx] = py[jy]
        Qy.append(Q)
        off += s
    return Px, Qy


def _csr(mats):
    """Dense [I, J_q] matrices of up to four sources -> (rowptr int32 [I+1], cols int32 = (q << 24) | j, vals f32),
    the operand format of rgda_sparse_mix."""
    I = mats[0].shape[0]
    rows, cols, vals = [], [], []
    for q, m in enumerate(mats):
        assert m.shape[0] == I and m.shape[1] < (1 << 24)
        nz = m.nonzero()
        rows.append(nz[:, 0])
        cols.append(nz[:, 1] + (q << 24))
        vals.append(m[nz[:, 0], nz[:, 1]])
    rows, cols, vals = torch.cat(rows), torch.cat(cols), torch.cat(vals)
    order = torch.argsort(rows * (1 << 32) + cols)
    rowptr = torch.zeros(I + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=I), 0)
    return rowptr.to(torch.int32), cols[order].to(torch.int32), vals[order].float().contiguous()


def ppm_factored_maps(h, w):
    """ppm_tap_matrix is separable: V_s[(y,x)][(jy,jx),(ky,kx)] = Uy_s[y+ky-1][jy] * Ux_s[x+kx-1][jx] (0 where the
    shifted row / column is padding).  With r = (s, jx, kx) indexing R = 3 * sum(s) intermediate rows per image row:
        Wx  [R, w]              Wx[r][x] = Ux_s[x+kx-1][jx]                     (the x direction, rgda_group_mix)
        Ay_s [h*R, 9*s*s]       Ay_s[y*R + r][(jy*s+jx)*9 + ky*3+kx] = Uy_s[y+ky-1][jy]   (the y direction, CSR)
    so that  V_s = blockdiag_y(Wx^T) @ Ay_s.  Returns dense Wx and the dense Ay_s list (callers build the CSR forms)."""
    R = 3 * sum(POOL_SCALES)
    Wx = torch.zeros(R, w)
    Ay = []
    off = 0
    for s in POOL_SCALES:
        Uy, Ux = upsample_taps_1d(s, h), upsample_taps_1d(s, w)
        A = torch.zeros(h * R, 9 * s * s)
        for jx in range(s):
            for kx in range(3):
                r = off + jx * 3 + kx
                x0, x1 = max(0, 1 - kx), min(w, w + 1 - kx)
                Wx[r, x0:x1] = Ux[x0 + kx - 1:x1 + kx - 1, jx]
                for jy in range(s):
                    for ky in range(3):
                        y0, y1 = max(0, 1 - ky), min(h, h + 1 - ky)
                        ys = torch.arange(y0, y1)
                        A[ys * R + r, (jy * s + jx) * 9 + ky * 3 + kx] = Uy[ys + ky - 1, jy]
        Ay.append(A)
        off += 3 * s
    return Wx, Ay


# ----------------------------------------------------------------------------- parameter specs
def _block_specs(resnet_type):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] at output stride 16
    (resnet.py:62-63,192-207: layer4 strides -> 1, its later blocks dilation 2)."""
    specs, inpl = [], 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), LAYERS[resnet_type]), start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li in (2, 3)) else 1
            dil = 2 if (li == 4 and bi > 0) else 1
            specs.append((f'encoder.resnet.layer{li}.{bi}', inpl, planes, stride, dil, bi == 0))
            inpl = planes * 4
    return specs


class _Conv:
    __slots__ = ('name', 'co', 'ci', 'k', 'stride', 'pad', 'dil', 'w', 'g', 'wb', 'wtb', 'bias', 'gbias')

    def out_hw(self, H, W):
        e = self.dil * (self.k - 1) + 1
        return (H + 2 * self.pad - e) // self.stride + 1, (W + 2 * self.pad - e) // self.stride + 1


class _BN:
    __slots__ = ('name', 'c', 'gamma', 'beta', 'rm', 'rv', 'nbt', 'dgamma', 'dbeta')


class _Container(nn.Module):
    pass


class _Lazy:
    """The output of a conv + BatchNorm (+ ReLU) unit whose BatchNorm has NOT been applied: the raw convolution output
    `c` and the statistic accumulators.  The consuming convolution applies relu(BatchNorm(c)) on its operand path
    (ops.conv2d_bnin) where a kernel carries the transform; otherwise `_materialise` runs the ordinary apply pass."""
    __slots__ = ('key', 'c', 'stats', 'bn', 'G', 'relu', 'M', 'shape')

    def __init__(self, key, c, stats, bn, G, relu, M):
        self.key, self.c, self.stats, self.bn, self.G, self.relu, self.M = key, c, stats, bn, G, relu, M
        self.shape = c.shape


FROM_X = 'sign-from-x'      # tape marker (in the ReLU-mask slot): the unit's activation was never written; the backward
                            # kernels recompute its ReLU sign from the raw convolution output (relu == 2)


def _attach(root, dotted, kind, tensor):
    """Register `tensor` as parameter/buffer under the reference's dotted name, creating containers."""
    parts = dotted.split('.')
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Container())
        m = m._modules[p]
    if kind == 'param':
        m.register_parameter(parts[-1], tensor)
    else:
        m.register_buffer(parts[-1], tensor)


def _pad64(n):
    return (n + 63) // 64 * 64


def _param_entries(resnet_type, head_kind, num_classes):
    """(name, kind, shape) of every state_dict entry in the reference's order (688 for ResNet-101 + two PPM heads), the
    block specifications and the name of a head's first convolution.  Pure: no device, no tensors."""
    entries = []            # (name, kind, shape) in the reference's state_dict order

    def conv(name, co, ci, k, bias=False):
        entries.append((name + '.weight', 'convw', (co, ci, k, k)))
        if bias:
            entries.append((name + '.bias', 'vec', (co,)))

    def bn(name, c):
        entries.append((name + '.weight', 'vec', (c,)))
        entries.append((name + '.bias', 'vec', (c,)))
        entries.append((name + '.running_mean', 'buf', (c,)))
        entries.append((name + '.running_var', 'buf', (c,)))
        entries.append((name + '.num_batches_tracked', 'nbt', ()))

    conv('encoder.resnet.conv1', 64, 3, 7); bn('encoder.resnet.bn1', 64)
    blocks = _block_specs(resnet_type)
    for p, inpl, planes, stride, dil, ds in blocks:
        conv(p + '.conv1', planes, inpl, 1); bn(p + '.bn1', planes)
        conv(p + '.conv2', planes, planes, 3); bn(p + '.bn2', planes)
        conv(p + '.conv3', planes * 4, planes, 1); bn(p + '.bn3', planes * 4)
        if ds:
            conv(p + '.downsample.0', planes * 4, inpl, 1); bn(p + '.downsample.1', planes * 4)
    for head in ('layer5', 'layer6'):
        if head_kind == 'aspp':
            for i in range(len(ASPP_DILATIONS)):
                conv(f'{head}.conv2d_list.{i}', num_classes, 2048, 3, bias=True)
            continue
        for i in range(4):
            conv(f'{head}.ppm.{i}.1', 512, 2048, 1); bn(f'{head}.ppm.{i}.2', 512)
        conv(f'{head}.conv_last.0', 512, 2048 + 4 * 512, 3); bn(f'{head}.conv_last.1', 512)
        conv(f'{head}.conv_last.4', num_classes, 512, 1, bias=True)
    head_first = 'ppm.0.1' if head_kind == 'ppm' else 'conv2d_list.0'

    return entries, blocks, head_first


def flat_layout(resnet_type='resnet101', head_kind='ppm', num_classes=6):
    """Element offset and size of every parameter in the flat fp32 buffers (flat_p / flat_g / momentum / EMA shadow):
    {name: (offset, numel)}, tensors padded to multiples of 64 elements, in `named_parameters()` order.  Pure host
    arithmetic (the CPU tests of the gradient exchange use it; `Deeplabv2._build_params` lays the buffers out the same way)."""
    entries, _, _ = _param_entries(resnet_type, head_kind, num_classes)
    out, op = {}, 0
    for name, kind, shape in entries:
        if kind in ('convw', 'vec'):
            n = math.prod(shape)
            out[name] = (op, n)
            op += _pad64(n)
    out['__total__'] = (op, 0)
    return out


def backward_progress_offsets(resnet_type='resnet101', head_kind='ppm', num_classes=6):
    """The offsets `Deeplabv2._backward_plan` reports through `on_progress`, in order: "every gradient at a flat offset >=
    this one is final" -- after both heads (the first head's first convolution), then after each residual block from the
    last to the first (its conv1).  The stem's gradients are final when backward returns (`FlatGradReducer.finish`)."""
    lay = flat_layout(resnet_type, head_kind, num_classes)
    _, blocks, head_first = _param_entries(resnet_type, head_kind, num_classes)
    offs = [lay[f'layer5.{head_first}.weight'][0]]
    offs += [lay[p + '.conv1.weight'][0] for p, *_ in reversed(blocks)]
    return offs


def bucket_boundaries(resnet_type='resnet101', head_kind='ppm', num_classes=6):
    """`Deeplabv2.param_boundaries()` without a model: the legal bucket cuts (block and head starts)."""
    lay = flat_layout(resnet_type, head_kind, num_classes)
    _, blocks, head_first = _param_entries(resnet_type, head_kind, num_classes)
    return sorted([lay[p + '.conv1.weight'][0] for p, *_ in blocks] +
                  [lay[f'{h}.{head_first}.weight'][0] for h in ('layer5', 'layer6')])


class Deeplabv2(nn.Module):
    def __init__(self, config):
        super().__init__()
        cfg = dict(backbone=dict(resnet_type='resnet50', output_stride=16, pretrained=True), multi_layer=False,
                   cascade=False, use_ppm=False, ppm=dict(num_classes=7, use_aux=False), inchannels=2048,
                   num_classes=7, is_ins_norm=False)          # Encoder.py:167-186 defaults
        for k, v in dict(config).items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k] = dict(cfg[k], **v)
            else:
                cfg[k] = v
        self.config = cfg
        rt = cfg['backbone']['resnet_type']
        if not (cfg['multi_layer'] and not cfg['cascade'] and cfg['is_ins_norm']):
            raise NotImplementedError('regda_amd builds the st.regda.* model: multi_layer=True, cascade=False, '
                                      'is_ins_norm=True, with the PPM heads (use_ppm=True) or the ASPP heads '
                                      '(use_ppm=False); the cascade / single-head variants are not built')
        # 'ppm': PPMBilinear heads (Encoder.py:10-64, what every st.regda entry point uses);
        # 'aspp': Classifier_Module heads, dilations = paddings = 6/12/18/24 (Encoder.py:68-84,111-114)
        self.head_kind = 'ppm' if cfg['use_ppm'] else 'aspp'
        if rt not in LAYERS or cfg['backbone'].get('output_stride', 16) != 16:
            raise NotImplementedError('resnet50/resnet101 at output_stride 16 only')
        if not torch.cuda.is_available():
            raise RuntimeError('regda_amd.Deeplabv2 needs an MI355X: there is no CPU fallback')
        self.resnet_type = rt
        self.num_classes = int(cfg['num_classes'])
        self.device = torch.device('cuda', torch.cuda.current_device())
        self._build_params()
        self._init_weights()
        self._anchor = torch.zeros(1, device=self.device, requires_grad=True)   # routes autograd into backward()
        self._drop_override = None
        self.fuse_bn_bwd = True      # fold BN-backward reductions into the producing data-gradient conv
        # weight gradients are collected and launched in groups (rgda_conv2d_wgrad_grouped) once this much work
        # is pending; 0 = one launch per layer
        self.wgrad_group_gflop = 500.0
        # ... and always where backward leaves these stages (so that what is still queued when the main chain ends is
        # one stage's worth, whatever the threshold): '' = never
        self.wgrad_flush_after = ('layer2',)
        # keep ReLU sign bits (1/16 of y) for the backward pass instead of re-reading y, for units with at least this
        # many channels (0 / False = never, 1 / True = always).  Round 1 kept them for the 1024-channel units only; with
        # the leaner BatchNorm kernels of round 2 every unit pays (A/B on one box: threshold 1024 -> 21.83, 512 -> 21.77,
        # 256 -> 21.79, all units -> 21.73 ms/step)
        self.relu_sign_mask = 1
        # PPM heads: apply the tap-shifted bilinear maps in their separable form (csrc/mix_kernels.hip); False = the
        # one-pass sparse maps (rgda_spatial_mix / rgda_spatial_mix_multi), kept as the cross-check
        self.factored_ppm = True
        self.parallel_heads = True       # training forward: the second head on its own stream
        self.group_small_convs = True    # the PPM branches' small convolutions: the four scales in one launch (rgda_conv2d_grouped)
        self.small_bn = True             # ... and their BatchNorms: statistics + apply / reduce + apply of the four scales in one launch
        self.parallel_tails = True       # backward: the PPM half of a head on the head stream, under the other head's convolution
        self.early_last_flush = True     # layer1's weight gradients start before the stem's backward
        self.fused_stem = True           # conv1 straight from the image where the map width allows it (rgda_stem_conv)
        self.fused_stem_wgrad = True     # ... and its weight gradient too (rgda_stem_wgrad): no patch matrix at all
        self.parallel_ds = True          # downsample branches on the head stream
        # bn1 / bn2 (+ ReLU) of every bottleneck and the stem's bn1 run on the CONSUMER's operand path (the next
        # convolution / the max-pool) wherever a kernel carries the transform: the activation is never written in the
        # forward pass (ops.conv2d_bnin).  False = one rgda_bn_train_apply pass per unit (the cross-check)
        self.bn_on_operand = True
        self.bn_operand_units = {'stem', 'bn1', 'bn2'}   # which units may be deferred (A/B experiments: bench.py --bn-operand)
        self.bn_operand_level = 2      # 2: only where rgda_conv2d_bnin_supported() says the transform pays; 1: wherever it is served
        self._head_stream = None
        self._mat_cache = {}
        self._synced_version = -1
        self.sync_weights()

    # ------------------------------------------------------------------ parameters
    def _build_params(self):
        dev = self.device
        entries, self.blocks, self._head_first = _param_entries(self.resnet_type, self.head_kind, self.num_classes)

        n_param = sum(_pad64(math.prod(s)) for _, k, s in entries if k in ('convw', 'vec'))
        n_buf = sum(_pad64(math.prod(s)) for _, k, s in entries if k == 'buf')
        n_nbt = sum(1 for _, k, _s in entries if k == 'nbt')
        self.flat_p = torch.zeros(n_param, device=dev)             # fp32 master weights
        self.flat_g = torch.zeros(n_param, device=dev)             # fp32 gradients
        self.flat_pb = torch.zeros(n_param, dtype=BF, device=dev)  # bf16 mirror (same offsets)
        self.flat_buf = torch.zeros(n_buf, device=dev)             # BN running statistics
        self.flat_nbt = torch.zeros((n_nbt + 1) // 2 * 2, dtype=torch.int64, device=dev)    # (whole 16-byte vectors: rgda_copy_multi)
        self.n_param_elems = sum(math.prod(s) for _, k, s in entries if k in ('convw', 'vec'))
        self._views, self._gviews = {}, {}
        self.convs, self.bns = {}, {}
        op = ob = on = 0
        wt_total = 0
        for name, kind, shape in entries:
            n = math.prod(shape)
            if kind == 'convw':
                co, ci, k, _ = shape
                phys = self.flat_p[op:op + n].view(co, k, k, ci)
                gphys = self.flat_g[op:op + n].view(co, k, k, ci)
                par = nn.Parameter(phys.permute(0, 3, 1, 2))       # logical [Cout,Cin,kh,kw], zero copy
                _attach(self, name, 'param', par)
                self._views[name], self._gviews[name] = par, gphys.permute(0, 3, 1, 2)
                c = _Conv()
                c.name, c.co, c.ci, c.k = name[:-7], co, ci, k
                c.stride, c.pad, c.dil = 1, 0, 1
                c.w, c.g = self.flat_p[op:op + n].view(co, k * k, ci), self.flat_g[op:op + n].view(co, k * k, ci)
                c.wb = self.flat_pb[op:op + n].view(co, k * k, ci)
                c.wtb, c.bias, c.gbias = None, None, None
                self.convs[c.name] = c
                wt_total += _pad64(n)
                op += _pad64(n)
            elif kind == 'vec':
                par = nn.Parameter(self.flat_p[op:op + n].view(shape))
                _attach(self, name, 'param', par)
                self._views[name], self._gviews[name] = par, self.flat_g[op:op + n].view(shape)
                base = name.rsplit('.', 1)[0]
                if base in self.convs:                               # classifier bias
                    self.convs[base].bias, self.convs[base].gbias = self.flat_p[op:op + n], self.flat_g[op:op + n]
                else:
                    b = self.bns.setdefault(base, _BN())
                    b.name, b.c = base, n
                    if name.endswith('.weight'):
                        b.gamma, b.dgamma = self.flat_p[op:op + n], self.flat_g[op:op + n]
                    else:
                        b.beta, b.dbeta = self.flat_p[op:op + n], self.flat_g[op:op + n]
                op += _pad64(n)
            elif kind == 'buf':
                t = self.flat_buf[ob:ob + n]
                _attach(self, name, 'buf', t)
                b = self.bns[name.rsplit('.', 1)[0]]
                if name.endswith('running_mean'):
                    b.rm = t
                else:
                    b.rv = t
                ob += _pad64(n)
            else:
                t = self.flat_nbt[on:on + 1].view(())
                _attach(self, name, 'buf', t)
                self.bns[name.rsplit('.', 1)[0]].nbt = t
                on += 1
        # geometry of the 3x3 / strided convs
        for p, inpl, planes, stride, dil, ds in self.blocks:
            c2 = self.convs[p + '.conv2']
            c2.stride, c2.pad, c2.dil = stride, dil, dil
            if ds:
                self.convs[p + '.downsample.0'].stride = stride
        for head in ('layer5', 'layer6'):
            if self.head_kind == 'aspp':
                for i, d in enumerate(ASPP_DILATIONS):
                    self.convs[f'{head}.conv2d_list.{i}'].pad = self.convs[f'{head}.conv2d_list.{i}'].dil = d
            else:
                self.convs[f'{head}.conv_last.0'].pad = 1
        # transposed bf16 weights for the data-gradient pass (every conv but the stem and the classifiers)
        self.flat_wt = torch.zeros(wt_total, dtype=BF, device=dev)
        o = 0
        for c in self.convs.values():
            if c.name == 'encoder.resnet.conv1' or c.bias is not None:
                continue
            n = c.co * c.k * c.k * c.ci
            c.wtb = self.flat_wt[o:o + n].view(c.ci, c.k * c.k, c.co)
            o += _pad64(n)
        rows, blk = [], 0
        for c in self.convs.values():
            if c.wtb is not None:
                T = c.k * c.k
                rows.append([self._mirror_ptr(c.w.data_ptr()), c.wtb.data_ptr(), c.co, T, c.ci, blk, c.ci, 0 | 16])
                blk += _layout_blocks(c.co, c.ci, T)
        self._wt_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._wt_blocks = blk
        self.stem_wb = torch.zeros(64, 1, STEM_KP, dtype=BF, device=dev)
        # fp32 landing buffer of the one weight gradient that is not written straight into the flat gradient (the stem's,
        # on maps the fused kernel does not serve), zeroed at the start of every backward pass
        self.grad_arena = torch.zeros(64 * STEM_KP, device=dev)
        self.stem_gtmp = self.grad_arena.view(64, 1, STEM_KP)
        # head 3x3 conv (4096 -> 512) split into its feature half and the four PPM branches (see ppm_tap_matrix):
        #   wfeat [512][9][2048]            the feature-map half, contiguous
        #   wz[i] [9*512][1][512]           Z = q_i @ W_tap^T for all nine taps at once (a 1x1 conv, Cout = 4608)
        #   wzt[i] [512][1][9*512]          its transpose, for the gradient w.r.t. q_i
        # (their weight gradients are written straight into the channel slices of the master gradient [512][9][4096]:
        #  rgda_wgrad_desc.lddw / co_split)
        self.head_w = {}
        self._hw_ready = None
        if self.head_kind == 'aspp':
            self._build_aspp_weights()
            return
        for head in ('layer5', 'layer6'):
            self.head_w[head] = {
                'wfeat': torch.zeros(512, 9, 2048, dtype=BF, device=dev),
                'wz': [torch.zeros(9 * 512, 1, 512, dtype=BF, device=dev) for _ in POOL_SCALES],
                'wzt': [torch.zeros(512, 1, 9 * 512, dtype=BF, device=dev) for _ in POOL_SCALES],
            }
        self._hw_ready = None
        # the head slices are two more tables for the same kernel: the forward operands (feature half + stacked tap
        # filters; also all the EMA teacher needs) and the transposes for the gradient w.r.t. the PPM branches

        def table(kinds):
            rows, blk = [], 0
            for head, hw in self.head_w.items():
                # source: the bf16 mirror of the master [512][9][4096] (mode bit 4; same rounding, half the bytes)
                wsrc = self._mirror_ptr(self.convs[f'{head}.conv_last.0'].w.data_ptr())
                if 'fwd' in kinds:
                    rows.append([wsrc, hw['wfeat'].data_ptr(), 512, 9, 2048, blk, 4096, 2 | 16])
                    blk += _layout_blocks(512, 2048, 9)
                for i in range(len(POOL_SCALES)):
                    src = wsrc + 2 * (2048 + 512 * i)
                    if 'fwd' in kinds:
                        rows.append([src, hw['wz'][i].data_ptr(), 512, 9, 512, blk, 4096, 1 | 16])
                        blk += _layout_blocks(512, 512, 9)
                    if 'bwd' in kinds:
                        rows.append([src, hw['wzt'][i].data_ptr(), 512, 9, 512, blk, 4096, 0 | 16])
                        blk += _layout_blocks(512, 512, 9)
            return torch.tensor(rows, dtype=torch.int64, device=dev), blk
        self._hw_fwd_table, self._hw_fwd_blocks = table(('fwd',))
        self._hw_bwd_table, self._hw_bwd_blocks = table(('bwd',))

    def _build_aspp_weights(self):
        """ASPP heads as one 1x1 convolution (csrc/aspp_kernels.hip): `aspp_wz` [ZC][1][2048] bf16 is the eight
        master weights [C][3][3][2048] back to back (one row per (head, dilation, class, tap), zero rows up to a
        multiple of 64), `aspp_wzt` its transpose for the gradient w.r.t. the features, `aspp_gz` the fp32
        landing buffer of the weight gradient."""
        dev, C = self.device, self.num_classes
        self.aspp_rows = 2 * len(ASPP_DILATIONS) * C * 9
        self.aspp_zc = (self.aspp_rows + 63) // 64 * 64
        self.aspp_wz = torch.zeros(self.aspp_zc, 1, 2048, dtype=BF, device=dev)
        self.aspp_wzt = torch.zeros(2048, 1, self.aspp_zc, dtype=BF, device=dev)
        self.aspp_gz = torch.zeros(self.aspp_zc, 1, 2048, device=dev)
        self.aspp_convs = [self.convs[f'{head}.conv2d_list.{i}'] for head in ('layer5', 'layer6')
                           for i in range(len(ASPP_DILATIONS))]
        rows, blk = [], 0
        for j, c in enumerate(self.aspp_convs):          # mode 2: fp32 master slice -> bf16, same layout
            rows.append([self._mirror_ptr(c.w.data_ptr()), self.aspp_wz.data_ptr() + 2 * j * C * 9 * 2048, C, 9, 2048, blk, 2048, 2 | 16])
            blk += _layout_blocks(C, 2048, 9)
        self._hw_fwd_table, self._hw_fwd_blocks = torch.tensor(rows, dtype=torch.int64, device=dev), blk

    def _init_weights(self):
        """kaiming_normal_(fan_out, relu) convs, BN weight 1 / bias 0 (_resnets.py:164-169); heads keep the
        torch defaults of nn.Conv2d (kaiming_uniform(a=sqrt(5)))."""
        with torch.no_grad():
            for name, par in self._views.items():
                if par.dim() == 4:
                    co, ci, k, _ = par.shape
                    if name.startswith('encoder.'):
                        par.copy_(torch.randn(co, ci, k, k, device=self.device) * math.sqrt(2.0 / (co * k * k)))
                    elif '.conv2d_list.' in name:                       # Encoder.py:77-78: normal_(0, 0.01)
                        par.copy_(torch.randn(co, ci, k, k, device=self.device) * 0.01)
                    else:
                        bound = 1.0 / math.sqrt(ci * k * k)
                        par.copy_((torch.rand(co, ci, k, k, device=self.device) * 2 - 1) * bound)
                elif name.endswith('.weight'):
                    par.fill_(1.0)
                else:
                    par.zero_()
            for c in self.convs.values():
                if c.bias is not None:
                    bound = 1.0 / math.sqrt(c.ci * c.k * c.k)
                    c.bias.copy_((torch.rand(c.co, device=self.device) * 2 - 1) * bound)
            for b in self.bns.values():
                b.rv.fill_(1.0)

    def _mirror_ptr(self, master_ptr):
        """Address in the bf16 mirror (flat_pb) of the element that lives at `master_ptr` in the fp32 master (flat_p)."""
        return self.flat_pb.data_ptr() + (master_ptr - self.flat_p.data_ptr()) // 2

    def sync_weights(self):
        """Refresh the bf16 mirror and, from it, the transposed copies."""
        ops.cast_bf16(self.flat_p, self.flat_pb)
        self.sync_derived_weights()
        self._synced_version = self.flat_p._version

    def sync_derived_weights(self, side_stream=None):
        """Transposed (data-gradient) copies + padded stem weights; the bf16 mirror is already fresh.
        The transposed copies are read by the NEXT backward pass only: with `side_stream` they are rebuilt there,
        off the critical path, and `_backward_plan` waits for the recorded event."""
        s = self.convs['encoder.resnet.conv1']
        ops.pad_cast_bf16(s.w, self.stem_wb, 64, 147, STEM_KP)
        if side_stream is None:
            ops.weight_transpose_batched(self._wt_table, self._wt_table.shape[0], self._wt_blocks)
            self._sync_head_weights(True)
            self._wt_ready = self._hw_ready = None
        else:
            plan.wait_event(side_stream, plan.record_event(torch.cuda.current_stream()))
            with ops.use_stream(side_stream):
                self._sync_head_weights(True)
                plan.host(lambda: setattr(self, '_hw_ready', side_stream.record_event()))    # needed by the next forward's heads
                ops.weight_transpose_batched(self._wt_table, self._wt_table.shape[0], self._wt_blocks)
            plan.host(lambda: setattr(self, '_wt_ready', side_stream.record_event()))        # needed by the next backward

    def _sync_head_weights(self, with_transposes):
        """Re-slice the head convs' weights from the fp32 master (same rounding as the bf16 mirror)."""
        ops.weight_transpose_batched(self._hw_fwd_table, self._hw_fwd_table.shape[0], self._hw_fwd_blocks)
        if with_transposes:
            if self.head_kind == 'aspp':
                plan.host(lambda: self.aspp_wzt.view(2048, self.aspp_zc).copy_(self.aspp_wz.view(self.aspp_zc, 2048).t()))
            else:
                ops.weight_transpose_batched(self._hw_bwd_table, self._hw_bwd_table.shape[0], self._hw_bwd_blocks)

    def _maybe_sync(self):
        if self.flat_p._version != self._synced_version:
            self.sync_weights()

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.sync_weights()
        return r

    def new_tape(self, groups=1):
        # ONE zeroed arena per step for every BatchNorm's forward statistics and backward sums (one clear instead of two)
        n = sum(groups * NREP * 2 * _pad64(b.c) for b in self.bns.values()) + 64
        buf = torch.empty(2 * n, dtype=torch.int64, device=self.device)
        ops.fill_zero(buf)
        return {'groups': groups, 'keep': [], 'stats_pool': _StatsPool(buf[:n]), 'sums_pool_next': _StatsPool(buf[n:])}

    def param_boundaries(self):
        """Element offsets (into flat_p / flat_g) where a residual block / head starts: legal bucket cuts."""
        base = self.flat_p.data_ptr()
        offs = [(self.convs[p + '.conv1'].w.data_ptr() - base) // 4 for p, *_ in self.blocks]
        offs += [(self.convs[f'{h}.{self._head_first}'].w.data_ptr() - base) // 4 for h in ('layer5', 'layer6')]
        return sorted(offs)

    def _offset_of(self, conv_name):
        return (self.convs[conv_name].w.data_ptr() - self.flat_p.data_ptr()) // 4

    def make_teacher(self):
        """EMA teacher (regda/utils/ema.py:41-58): its own (shadow) parameters.  ema.py averages parameters only, so
        the BatchNorm buffers are the student's -- taken as a SNAPSHOT (`adopt_buffers`) at a defined point of the
        step, never aliased: the teacher's eval forward may run on another stream next to the student's training
        forward, which rewrites those buffers."""
        t = Deeplabv2(self.config)
        with torch.no_grad():
            t.flat_p.copy_(self.flat_p)                       # ema.register(): shadow = param.clone()
        t.adopt_buffers(self)
        t.refresh_from_master()
        t.eval()
        return t

    def adopt_buffers(self, other):
        """Copy `other`'s BatchNorm running statistics (one flat buffer each) on the current stream."""
        ops.copy_multi([(self.flat_buf, other.flat_buf), (self.flat_nbt, other.flat_nbt)])

    def refresh_from_master(self, mirror_is_fresh=False):
        """bf16 mirror + padded stem weights only (a forward-only model needs no transposed copies).
        mirror_is_fresh: the caller's optimizer already wrote flat_pb (rgda_sgd_step's shadow_bf16)."""
        if not mirror_is_fresh:
            ops.cast_bf16(self.flat_p, self.flat_pb)
        s = self.convs['encoder.resnet.conv1']
        ops.pad_cast_bf16(s.w, self.stem_wb, 64, 147, STEM_KP)
        self._sync_head_weights(False)
        self._synced_version = self.flat_p._version

    def set_drop_masks(self, m5, m6):
        """Test hook: fix the Dropout2d(0.1) keep-masks (b,512) of the two heads (None -> random)."""
        self._drop_override = None if m5 is None else (m5, m6)

    # ------------------------------------------------------------------ building blocks
    def _mats(self, h, w):
        key = (h, w)
        if key not in self._mat_cache:
            d = {}
            for s in POOL_SCALES:
                P = pool_matrix(h, w, s)
                U = upsample_matrix(s, s, h, w)
                V = ppm_tap_matrix(h, w, s)
                d[s] = tuple(t.to(self.device).contiguous() for t in (P, P.t(), U, U.t(), V, V.t()))
            self._mat_cache[key] = d
        return self._mat_cache[key]

    def _ppm_maps(self, h, w):
        """Device operands of the factored tap-shifted bilinear maps (ppm_factored_maps)."""
        key = ('ppm', h, w)
        if key not in self._mat_cache:
            dev = self.device
            Wx, Ay = ppm_factored_maps(h, w)
            self._mat_cache[key] = {
                'R': Wx.shape[0], 'Wx': Wx.to(dev).contiguous(), 'Wxt': Wx.t().contiguous().to(dev),
                'fwd': tuple(t.to(dev) for t in _csr(Ay)),                       # B rows <- the four Z_s
                'bwd': tuple(t.to(dev) for t in _csr([torch.cat([A.t() for A in Ay], 0).contiguous()])),  # dZ_s <- A
            }
        return self._mat_cache[key]

    def _pool_maps(self, h, w):
        """Device operands of the factored adaptive-average-pool maps (pool_factored_maps)."""
        key = ('pool', h, w)
        if key not in self._mat_cache:
            dev = self.device
            Px, Qy = pool_factored_maps(h, w)
            self._mat_cache[key] = {
                'R': Px.shape[0], 'Px': Px.to(dev).contiguous(), 'Pxt': Px.t().contiguous().to(dev),
                'fwd': tuple(t.to(dev) for t in _csr([torch.cat(Qy, 0).contiguous()])),      # pooled_s rows <- x-pooled rows
                'bwd': tuple(t.to(dev) for t in _csr([Q.t().contiguous() for Q in Qy])),     # rows <- the four dpool_s
            }
        return self._mat_cache[key]

    def _conv_stats(self, x, w, c, stats, G, N, H, W, Ho, Wo, k, stride, pad, dil, res=None, queue=None):
        """Forward conv (+ `res` added in the epilogue, before the statistics) with BatchNorm statistics per row
        group; falls back to one launch per group when the groups are not a multiple of the pixel tile (tiny PPM
        maps).  queue (a list): the launch(es) are appended as ops.conv2d_grouped items instead of being made."""
        if G == 1 or stats is None:
            if queue is not None:
                queue.append((x, w, c, N, H, W, Ho, Wo, k, k, stride, pad, dil, 0, res, stats, 1))
            else:
                ops.conv2d(x, w, c, N, H, W, Ho, Wo, k, k, stride, pad, dil, 0, res, stats, 1)
            return
        whole = (x, w, c, N, H, W, Ho, Wo, k, k, stride, pad, dil, 0, res, stats, G)
        try:
            if queue is not None:
                ops.conv2d_grouped_launches([whole])        # (validation only: raises where the groups do not tile)
                queue.append(whole)
            else:
                ops.conv2d(*whole)
        except ValueError:
            Ng, C = N // G, c.shape[1]
            st = stats.view(G, -1)
            for g in range(G):
                ro = slice(g * Ng * Ho * Wo, (g + 1) * Ng * Ho * Wo)
                part = (x[g * Ng * H * W:(g + 1) * Ng * H * W], w, c[ro], Ng, H, W, Ho, Wo, k, k, stride, pad, dil, 0,
                        None if res is None else res[ro], st[g], 1)
                if queue is not None:
                    queue.append(part)
                else:
                    ops.conv2d(*part)

    def _materialise(self, T, lz):
        """The ordinary apply pass for a deferred unit whose consumer has no operand-transform kernel."""
        mi = torch.empty(lz.G, 2, lz.bn.c, device=self.device)
        y = torch.empty(lz.M, lz.bn.c, dtype=BF, device=self.device)
        rmask = (torch.empty(lz.M, lz.bn.c // 8, dtype=torch.uint8, device=self.device)
                 if (lz.relu and self.relu_sign_mask and lz.bn.c >= self.relu_sign_mask) else None)
        geom = T[lz.key][4]
        ops.bn_train_apply(lz.c, lz.stats, mi, lz.bn.rm, lz.bn.rv, lz.bn.nbt, lz.bn.gamma, lz.bn.beta, y, lz.M, lz.bn.c,
                           lz.relu, None, None, geom[3] * geom[4], groups=lz.G, relu_mask=rmask)
        e = T[lz.key]
        T[lz.key] = (e[0], e[1], y, mi, e[4], e[5], rmask)
        return y

    def _bn_operand(self, T, lz):
        """rgda_bn_operand of a deferred unit + its tape entry: (mean, invstd) are written by the consumer's launch."""
        mi = torch.empty(lz.G, 2, lz.bn.c, device=self.device)
        e = T[lz.key]
        T[lz.key] = (e[0], e[1], None, mi, e[4], e[5], FROM_X)
        bnop = ops.bn_operand(lz.stats, lz.bn.gamma, lz.bn.beta, mi, lz.bn.rm, lz.bn.rv, lz.bn.nbt, lz.G, lz.relu)
        T['keep'].append(bnop)
        return bnop

    def _cbr_fwd(self, T, key, conv, bn, x, N, H, W, relu, res=None, nscale=None, wb=None, geom=None, defer=False,
                 preconv=None):
        """conv + BatchNorm (+ residual + ReLU) unit.  `x` may be a deferred unit (_Lazy): its BatchNorm + ReLU then runs on
        this convolution's operand path.  defer=True (training only): return this unit deferred in turn."""
        Ho, Wo = conv.out_hw(H, W) if geom is None else geom
        M = N * Ho * Wo
        train = T is not None
        G = T['groups'] if train else 1
        if not train:       # inference: conv + BN + residual + ReLU in one kernel
            y = torch.empty(M, conv.co, dtype=BF, device=self.device)
            if geom is None:
                ops.conv2d_bneval(x, conv.wb if wb is None else wb, y, N, H, W, Ho, Wo, conv.k, conv.k, conv.stride,
                                  conv.pad, conv.dil, bn.rm, bn.rv, bn.gamma, bn.beta, relu, res)
            else:
                ops.conv2d_bneval(x, wb, y, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1, bn.rm, bn.rv, bn.gamma, bn.beta, relu, res)
            return y, Ho, Wo
        if M // G < 2:
            raise ValueError('Expected more than 1 value per channel when training')
        if preconv is not None:        # the convolution (with its statistics) was launched by the caller, grouped with others
            c, stats = preconv
        else:
            c = torch.empty(M, conv.co, dtype=BF, device=self.device)
            stats = T['stats_pool'].take(G * NREP * 2 * conv.co) if train else None
        if isinstance(x, _Lazy):
            if geom is None and ops.conv2d_bnin_supported(M, conv.co, conv.ci, conv.k, conv.k, conv.stride, conv.pad,
                                                          conv.dil, H, W, Ho, Wo, G) >= self.bn_operand_level:
                ops.conv2d_bnin(self._bn_operand(T, x), x.c, conv.wb if wb is None else wb, c, N, H, W, Ho, Wo, conv.k,
                                conv.k, conv.stride, conv.pad, conv.dil, None, stats, G)
            else:
                x = self._materialise(T, x)
        if not isinstance(x, _Lazy) and preconv is None:
            if geom is None:
                self._conv_stats(x, conv.wb if wb is None else wb, c, stats, G, N, H, W, Ho, Wo, conv.k, conv.stride,
                                 conv.pad, conv.dil)
            else:       # stem: GEMM over the im2col matrix
                self._conv_stats(x, wb, c, stats, G, N, Ho, Wo, Ho, Wo, 1, 1, 0, 1)
        if defer and self.bn_on_operand and res is None and nscale is None:
            T[key] = (x, c, None, None, (N, H, W, Ho, Wo), None, None)      # completed by the consumer (_bn_operand / _materialise)
            return _Lazy(key, c, stats, bn, G, relu, M), Ho, Wo
        mi = torch.empty(G, 2, conv.co, device=self.device)
        y = torch.empty(M, conv.co, dtype=BF, device=self.device)
        # the backward pass needs only the SIGN of y (ReLU): one bit per element written next to y.  The byte
        # stores cost the forward ~2 us per launch, so only wide units do it (see relu_sign_mask).
        rmask = (torch.empty(M, conv.co // 8, dtype=torch.uint8, device=self.device)
                 if (relu and self.relu_sign_mask and conv.co >= self.relu_sign_mask) else None)
        ops.bn_train_apply(c, stats, mi, bn.rm, bn.rv, bn.nbt, bn.gamma, bn.beta, y, M, conv.co, relu, res, nscale,
                           Ho * Wo, groups=G, relu_mask=rmask)
        T[key] = (x, c, y, mi, (N, H, W, Ho, Wo), nscale, rmask)
        return y, Ho, Wo

    def _head_last_fwd(self, T, head, xn, qs, N, h, w, nscale):
        """conv_last.0 (3x3, 4096 -> 512) + BN + ReLU (+ Dropout2d scale) of a PPM head WITHOUT materialising the
        4096-channel concat (Encoder.py:30-51): the feature half is a 3x3 conv over the 2048 instance-normalised
        channels; the four upsampled PPM branches contribute  V_i @ (q_i W_i^T)  (ppm_tap_matrix), i.e. four tiny
        1x1 convs at s x s resolution and one spatial mix, added in the feature conv's epilogue before the BN
        statistics.  Exactly the reference arithmetic re-associated; half the FLOPs of the head conv."""
        C, B = self.convs, self.bns
        conv, bn = C[f'{head}.conv_last.0'], B[f'{head}.conv_last.1']
        hw = self.head_w[head]
        dev = self.device
        HW, M = h * w, N * h * w
        mats = self._mats(h, w)
        zs, queue = [], []
        for i, s in enumerate(POOL_SCALES):
            z = torch.empty(N * s * s, 9 * 512, dtype=BF, device=dev)
            queue.append((qs[i], hw['wz'][i], z, N, s, s, s, s, 1, 1, 1, 0, 1))
            zs.append(z.view(N * s * s * 9, 512))
        if self.group_small_convs:
            ops.conv2d_grouped(queue)           # the four scales' Z = q W^T in one launch
        else:
            for it in queue:
                ops.conv2d(*it)
        ppm = torch.empty(M, 512, dtype=BF, device=dev)
        if self.factored_ppm:       # V @ Z as the y map (gather from the four Z_s) then the x map (LDS-staged rows)
            fm = self._ppm_maps(h, w)
            rows = torch.empty(N * h * fm['R'], 512, device=dev)
            ops.sparse_mix(zs, fm['fwd'], rows, N, 512)
            ops.group_mix(rows, fm['Wxt'], ppm, N * h, w, fm['R'], 512)
        else:
            ops.spatial_mix_multi(zs, [mats[s][4] for s in POOL_SCALES], ppm, N, HW, 512)
        train = T is not None
        G = T['groups'] if train else 1
        c = torch.empty(M, 512, dtype=BF, device=dev)
        stats = T['stats_pool'].take(G * NREP * 2 * 512) if train else None
        self._conv_stats(xn, hw['wfeat'], c, stats, G, N, h, w, h, w, 3, 1, 1, 1, res=ppm)
        mi = torch.empty(G, 2, 512, device=dev)
        y = torch.empty(M, 512, dtype=BF, device=dev)
        if train:
            ops.bn_train_apply(c, stats, mi, bn.rm, bn.rv, bn.nbt, bn.gamma, bn.beta, y, M, 512, True, None, nscale,
                               HW, groups=G)
            T[f'{head}.last'] = (xn, c, y, mi, (N, h, w, h, w), nscale, None)
            T[f'{head}.last.q'] = qs
        else:
            ops.bn_finalize(None, mi, bn.rm, bn.rv, None, M, 512)
            ops.bn_apply(c, mi, bn.gamma, bn.beta, y, M, 512, True, None, nscale, HW, groups=G)
        return y

    def _head_last_bwd(self, T, head, g, dfeat_prev, tail=True):
        """Backward of _head_last_fwd: BN backward, then dW / dX of the feature half as an ordinary 3x3 conv over 2048
        channels and, per PPM branch, dZ_i = V_i^T @ dc (a pooling with the tap-shifted bilinear weights),
        dq_i = dZ_i W_i and dW_i = dZ_i^T q_i at s x s resolution.  Returns (dfeat [M,2048], [dq_i]); tail=False: the
        feature half only -> (dfeat, dc), the caller runs _head_last_bwd_tail."""
        C, B = self.convs, self.bns
        conv, bn = C[f'{head}.conv_last.0'], B[f'{head}.conv_last.1']
        hw = self.head_w[head]
        dev = self.device
        key = f'{head}.last'
        xn, c, y, mi, (N, h, w, _, _), nscale, _ = T[key]
        qs = T[key + '.q']
        HW, M, G = h * w, N * h * w, T['groups']
        mats = self._mats(h, w)
        sums = T['sums_pool'].take(G * NREP * 2 * 512)
        ops.bn_bwd_reduce(g, y, c, mi, sums, M, 512, True, nscale, HW, groups=G)
        dc = torch.empty(M, 512, dtype=BF, device=dev)
        ops.bn_bwd_apply(g, y, c, mi, bn.gamma, sums, dc, M, 512, True, None, bn.dgamma, bn.dbeta, nscale, HW, groups=G)
        gview = conv.g.view(512, 9, 4096)
        # the weight gradients go straight into the channel slices of the master gradient (row stride 4096)
        T['wgrad_pending'].append((xn, dc, gview[:, :, :2048], N, h, w, h, w, 3, 3, 1, 1, 1))
        T['wgrad_pending_flop'] += 2.0 * M * 512 * 2048 * 9
        dfeat = torch.empty(M, 2048, dtype=BF, device=dev)
        ops.conv2d(dc, conv.wtb[:2048], dfeat, N, h, w, h, w, 3, 3, 1, 1, 1, 1, dfeat_prev, None)
        T['keep'].append((dc, dfeat))
        if tail:
            return dfeat, self._head_last_bwd_tail(T, head, dc)
        return dfeat, dc

    def _head_last_bwd_tail(self, T, head, dc):
        """The PPM half of _head_last_bwd: dZ_i = V_i^T dc, dq_i = dZ_i W_i, dW_i queued -> [dq_i].  A dozen launches of at
        most 113 workgroups: with `parallel_tails` they run on the head stream under the other head's convolution."""
        C = self.convs
        conv = C[f'{head}.conv_last.0']
        hw = self.head_w[head]
        dev = self.device
        xn, c, y, mi, (N, h, w, _, _), nscale, _ = T[f'{head}.last']
        qs = T[f'{head}.last.q']
        HW = h * w
        mats = self._mats(h, w)
        gview = conv.g.view(512, 9, 4096)
        dqs = []
        if self.factored_ppm:       # V^T @ dc: the x map once for all scales, then one y map per scale
            fm = self._ppm_maps(h, w)
            rows = torch.empty(N * h * fm['R'], 512, device=dev)
            ops.group_mix(dc, fm['Wx'], rows, N * h, fm['R'], w, 512)
            dzs = [torch.empty(N * s * s * 9, 512, dtype=BF, device=dev) for s in POOL_SCALES]
            ops.sparse_mix([rows], fm['bwd'], dzs, N, 512)
        queue = []
        for i, s in enumerate(POOL_SCALES):
            if self.factored_ppm:
                dz = dzs[i]
            else:
                dz = torch.empty(N * s * s * 9, 512, dtype=BF, device=dev)
                ops.spatial_mix(dc, mats[s][5], dz, N, 9 * s * s, HW, 512)
            dzr = dz.view(N * s * s, 9 * 512)
            dq = torch.empty(N * s * s, 512, dtype=BF, device=dev)
            # dq_i = dZ_i W_i: K = 4608 on 1 - 9 pixel tiles (72 K tiles on 4 - 36 workgroups, 35 us each alone): the four
            # scales share one launch
            if self.group_small_convs:
                queue.append((dzr, hw['wzt'][i], dq, N, s, s, s, s, 1, 1, 1, 0, 1))
            else:
                ops.conv2d(dzr, hw['wzt'][i], dq, N, s, s, s, s, 1, 1, 1, 0, 1)
            # nine stacked 1x1 filters (rows tap * 512 + co) written channel-major into [co][tap][2048 + 512 i ...]
            T['wgrad_pending'].append((qs[i], dzr, gview[:, :, 2048 + 512 * i: 2048 + 512 * (i + 1)], N, s, s, s, s, 1, 1, 1, 0, 1))
            T['keep'].append((dz, dq))
            dqs.append(dq)
        if queue:
            ops.conv2d_grouped(queue)
        return dqs

    def _aspp_fwd(self, T, xn, N, h, w):
        """Both Classifier_Module heads (Encoder.py:80-84): one 1x1 convolution for all 2 x 4 x 9 taps, then the
        dilated gather (csrc/aspp_kernels.hip)."""
        dev, C = self.device, self.num_classes
        z = torch.empty(N * h * w, self.aspp_zc, dtype=BF, device=dev)
        ops.conv2d(xn, self.aspp_wz, z, N, h, w, h, w, 1, 1, 1, 0, 1, 0)
        x1 = torch.empty(N, C, h, w, device=dev)
        x2 = torch.empty(N, C, h, w, device=dev)
        ops.aspp_gather(z, [c.bias for c in self.aspp_convs], x1, x2, N, h, w, C, ASPP_DILATIONS)
        if T is not None:
            T['aspp'] = (xn, (N, h, w))
        return x1, x2

    def _aspp_bwd(self, T, g1, g2):
        """-> d(loss)/d(xn) bf16 [M, 2048]; weight and bias gradients are accumulated into the master gradients."""
        dev, C = self.device, self.num_classes
        xn, (N, h, w) = T['aspp']
        dz = torch.empty(N * h * w, self.aspp_zc, dtype=BF, device=dev)
        ops.aspp_scatter(g1, g2, dz, [c.gbias for c in self.aspp_convs],
                         N, h, w, C, ASPP_DILATIONS)
        dxn = torch.empty(N * h * w, 2048, dtype=BF, device=dev)
        ops.conv2d(dz, self.aspp_wzt, dxn, N, h, w, h, w, 1, 1, 1, 0, 1, 0)
        plan.host(self.aspp_gz.zero_)
        ops.conv2d_wgrad(xn, dz, self.aspp_gz, N, h, w, h, w, 1, 1, 1, 0, 1)
        n = C * 9 * 2048
        for j, c in enumerate(self.aspp_convs):      # the rows of one conv are its master layout [C][3][3][2048]
            plan.host(lambda j=j, c=c: c.g.view(-1).add_(self.aspp_gz.view(-1)[j * n:(j + 1) * n]))
        return dxn

    def _flush_wgrads(self, T):
        """Launch the queued weight gradients (grouped by kernel) -- on the second HIP stream when there is one,
        next to the BN-backward / data-gradient chain of the layers below, which is the critical path -- and
        release the all-reduce progress mark that was waiting for them."""
        pend = T['wgrad_pending']
        if pend:
            side = T.get('wgrad_stream')
            if side is None:
                ops.conv2d_wgrad_grouped(pend)
            else:
                plan.wait_event(side, plan.record_event(T['main_stream']))
                with ops.use_stream(side):
                    ops.conv2d_wgrad_grouped(pend)
                # keep the operands alive until the streams join (no record_stream: the step must stay
                # capturable into a hipGraph)
                T['keep'].extend((it[0], it[1]) for it in pend)
            post = T.get('wgrad_post')
            if post:        # strided adds of densely written gradients (head convs), behind the launch that made them
                if side is None:
                    for fn in post:
                        plan.host(fn)
                else:
                    with ops.use_stream(side):
                        for fn in post:
                            plan.host(fn)
                T['wgrad_post'] = []
            T['wgrad_pending'] = []
            T['wgrad_pending_flop'] = 0.0
        off = T.pop('progress_deferred', None)
        if off is not None:
            T['on_progress'](off)

    def _progress(self, T, offset):
        """All gradients of the parameters at flat offsets >= `offset` have been produced or queued."""
        if T.get('on_progress') is None:
            return
        if T['wgrad_pending']:
            T['progress_deferred'] = offset       # released by the flush that launches the queued layers
        else:
            T['on_progress'](offset)

    def _cbr_bwd(self, T, key, conv, bn, g, relu, need_dx=True, want_gmask=False, dx_res=None, stem=False,
                 consumer=None, dx_res_mask=None, conv_queue=None, bn_queue=None):
        """Backward of one conv+BN(+ReLU) unit.  `consumer` = (tape key, relu) of the unit that will consume this
        unit's data gradient: its BN-backward reduction is then folded into our data-gradient conv's epilogue.
        `dx_res` (+ optional ReLU sign mask gating it) is added to the data gradient in the same epilogue."""
        x, c, y, mi, (N, H, W, Ho, Wo), nscale, rmask = T[key]
        M, C, G = N * Ho * Wo, conv.co, T['groups']
        # a unit whose BatchNorm + ReLU ran on its consumer's operand path: no y, no sign mask -- the ReLU sign is recomputed
        # from c (relu == 2), and the activation the consumer's WEIGHT gradient needs is written by our backward apply
        from_x = rmask is FROM_X
        rl = 2 if (from_x and relu) else relu
        if from_x:
            rmask = None
        sums = T.pop('sums:' + key, None)
        deferred = T.pop('wgrad_deferred:' + key, None)
        dc = torch.empty(M, C, dtype=BF, device=self.device)
        gm = torch.empty(M, C, dtype=BF, device=self.device) if want_gmask else None
        act = torch.empty(M, C, dtype=BF, device=self.device) if deferred is not None else None
        if (bn_queue is not None and conv_queue is not None and sums is None and rl != 2 and nscale is None and not want_gmask
                and deferred is None and M // G <= ops.BN_SMALL_MAX_ROWS and ('presums:' + key) not in T):
            # a small map: reduce + apply in one workgroup per 128 channels, launched by the caller together with its
            # siblings (ops.bn_bwd_small) -- and before the data gradients it queues in conv_queue
            bn_queue.append((g, y if (rl == 1 and rmask is None) else None, c, mi, bn.gamma, dc, bn.dgamma, bn.dbeta, M, C,
                             rl, G, rmask))
        else:
            if sums is None:
                sums = T.pop('presums:' + key, None)        # arena slice reserved by a producer that could not fuse
                if sums is None:
                    sums = T['sums_pool'].take(G * NREP * 2 * C)
                ops.bn_bwd_reduce(g, y if (rl == 1 and rmask is None) else None, c, mi, sums, M, C, rl, nscale, Ho * Wo,
                                  groups=G, relu_mask=rmask, gamma=bn.gamma, beta=bn.beta)
            ops.bn_bwd_apply(g, y if (rl == 1 and rmask is None) else None, c, mi, bn.gamma, sums, dc, M, C, rl, gm,
                             bn.dgamma, bn.dbeta, nscale, Ho * Wo, groups=G, relu_mask=rmask, beta=bn.beta, act_out=act)
        if deferred is not None:        # the consumer's weight gradient: its operand exists from here on
            ddc, dg, geom, flop = deferred
            T['wgrad_pending'].append((act, ddc, dg) + geom)
            T['wgrad_pending_flop'] += flop
        if stem:
            if isinstance(x, tuple):        # the fp32 image batches themselves (one per BatchNorm group): no patch matrix
                Ng = N // len(x)
                for gi, xg in enumerate(x):
                    ops.stem_wgrad(xg, dc[gi * Ng * Ho * Wo:(gi + 1) * Ng * Ho * Wo], conv.g.view(64, 147), Ng, H, W, Ho, Wo)
            else:
                ops.fill_zero(self.grad_arena)          # fp32 landing buffer of the padded [64][192] gradient
                ops.conv2d_wgrad(x, dc, self.stem_gtmp, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1)
                ops.unpad_acc_f32(self.stem_gtmp, conv.g, 64, 147, STEM_KP)
            return None, gm
        # weight gradients are only needed by the optimizer: they are queued, and launched in groups
        wgeom = (N, H, W, Ho, Wo, conv.k, conv.k, conv.stride, conv.pad, conv.dil)
        wflop = 2.0 * M * conv.co * conv.ci * conv.k * conv.k
        if isinstance(x, _Lazy):        # the operand was never written: queued by the producing unit's backward (above)
            T['wgrad_deferred:' + x.key] = (dc, conv.g, wgeom, wflop)
        else:
            T['wgrad_pending'].append((x, dc, conv.g) + wgeom)
            T['wgrad_pending_flop'] += wflop
        if T['wgrad_pending_flop'] >= self.wgrad_group_gflop * 1e9:
            self._flush_wgrads(T)
        dx = None
        if need_dx:
            dx = torch.empty(N * H * W, conv.ci, dtype=BF, device=self.device)
            fused = False
            if consumer is not None and self.fuse_bn_bwd:
                ckey, crelu = consumer
                cx, cc, cy, cmi, (cN, cH, cW, cHo, cWo), cns, cmask = T[ckey]
                assert cN * cHo * cWo == N * H * W and cc.shape[1] == conv.ci
                csums = T['sums_pool'].take(G * NREP * 2 * conv.ci)
                cbn = None
                if cmask is FROM_X:     # the consumer's activation was never written: its ReLU sign comes from cc
                    cbn, cmask, crelu = x.bn, None, (2 if crelu else 0)
                try:
                    ops.conv2d_bnbwd(dc, conv.wtb, dx, N, Ho, Wo, H, W, conv.k, conv.k, conv.stride, conv.pad, conv.dil,
                                     1, dx_res, csums, G, cy if (crelu == 1 and cmask is None) else None, cc, cmi, crelu,
                                     cns, cHo * cWo, relu_mask=cmask, res_mask=dx_res_mask,
                                     bn_gamma=None if cbn is None else cbn.gamma, bn_beta=None if cbn is None else cbn.beta)
                    T['sums:' + ckey] = csums
                    fused = True
                except ValueError:          # row groups do not tile (tiny maps): plain conv, standalone reduction
                    T['presums:' + ckey] = csums
            if not fused:
                if conv_queue is not None:      # launched by the caller, grouped with its siblings (ops.conv2d_grouped)
                    conv_queue.append((dc, conv.wtb, dx, N, Ho, Wo, H, W, conv.k, conv.k, conv.stride, conv.pad, conv.dil, 1,
                                       dx_res, None, 1, dx_res_mask))
                else:
                    ops.conv2d(dc, conv.wtb, dx, N, Ho, Wo, H, W, conv.k, conv.k, conv.stride, conv.pad, conv.dil, 1,
                               dx_res, None, res_mask=dx_res_mask)
        return dx, gm

    def _stem_fwd(self, T, xs, Ng, H, W, H1, W1, main_stream):
        """conv1 + bn1 + ReLU from the fp32 images in one convolution kernel per BatchNorm group (rgda_stem_conv: no patch
        matrix).  The weight gradient reads the images again at the end of backward (rgda_stem_wgrad): training keeps
        private copies of them, made on the head stream next to the forward (without `fused_stem_wgrad`: the patch matrix
        of rgda_stem_im2col instead); -> (activation, event after the copies / the im2col)."""
        dev = self.device
        conv, bn = self.convs['encoder.resnet.conv1'], self.bns['encoder.resnet.bn1']
        G = len(xs)
        N, M = Ng * G, Ng * G * H1 * W1
        y = torch.empty(M, 64, dtype=BF, device=dev)
        if T is None:
            for gi, xg in enumerate(xs):
                ops.stem_conv_bneval(xg, self.stem_wb, y[gi * Ng * H1 * W1:(gi + 1) * Ng * H1 * W1], bn.rm, bn.rv, bn.gamma,
                                     bn.beta, True, Ng, H, W, H1, W1)
            return y, None
        side = None
        if self.parallel_heads:
            if self._head_stream is None:
                self._head_stream = torch.cuda.Stream(device=dev)
            side = self._head_stream
            plan.wait_event(side, plan.record_event(main_stream))       # the images are ready on the main stream
        with (ops.use_stream(side) if side is not None else contextlib.nullcontext()):
            if self.fused_stem_wgrad:
                # the weight gradient reads the images again at the END of backward (rgda_stem_wgrad): private copies, so
                # the caller's buffers are free for the next batch once the forward has read them (25 MB, beside the forward)
                col = tuple(torch.empty_like(xg) for xg in xs)
                for i in range(0, len(xs), 4):
                    ops.copy_multi(list(zip(col[i:i + 4], xs[i:i + 4])))
            else:
                col = torch.empty(M, STEM_KP, dtype=BF, device=dev)
                for gi, xg in enumerate(xs):
                    ops.stem_im2col(xg, col[gi * Ng * H1 * W1:(gi + 1) * Ng * H1 * W1], Ng, H, W, H1, W1)
            col_ready = plan.record_event(side) if side is not None else None
        c = torch.empty(M, 64, dtype=BF, device=dev)
        stats = T['stats_pool'].take(G * NREP * 2 * 64)
        st = stats.view(G, -1)
        for gi, xg in enumerate(xs):
            ops.stem_conv(xg, self.stem_wb, c[gi * Ng * H1 * W1:(gi + 1) * Ng * H1 * W1], st[gi], Ng, H, W, H1, W1)
        if self.bn_on_operand and 'stem' in self.bn_operand_units and getattr(self, '_debug_taps', None) is None:
            # bn1 + ReLU run on the max-pool's operand path (ops.maxpool_fwd_bnin): the 64-channel full-resolution
            # activation (134 MB for 16 images of 512 x 512) is never written
            T['stem'] = (col, c, None, None, (N, H, W, H1, W1), None, None)
            return _Lazy('stem', c, stats, bn, G, True, M), col_ready
        mi = torch.empty(G, 2, 64, device=dev)
        rmask = (torch.empty(M, 8, dtype=torch.uint8, device=dev) if (self.relu_sign_mask and 64 >= self.relu_sign_mask) else None)
        ops.bn_train_apply(c, stats, mi, bn.rm, bn.rv, bn.nbt, bn.gamma, bn.beta, y, M, 64, True, None, None, H1 * W1, groups=G,
                           relu_mask=rmask)
        T['stem'] = (col, c, y, mi, (N, H, W, H1, W1), None, rmask)
        return y, col_ready

    def _block_fwd(self, T, blk, y, N, h, w, main_stream, bs=None):
        """One bottleneck block (regda/_resnets.py:92-112) on the pixel-major map y [N*h*w][inplanes]: returns (out, h', w').
        bs: the stream a downsample branch runs on next to conv1 .. conv3 (None: the current one)."""
        C, B = self.convs, self.bns
        p, inpl, planes, stride, dil, ds = blk
        idt = y
        joined = None
        if ds:
            if bs is not None:
                plan.wait_event(bs, plan.record_event(main_stream))
            with (ops.use_stream(bs) if bs is not None else contextlib.nullcontext()):
                idt, _, _ = self._cbr_fwd(T, p + '.d', C[p + '.downsample.0'], B[p + '.downsample.1'], y, N, h, w,
                                          False)
                if bs is not None:
                    joined = plan.record_event(bs)
        # bn1 / bn2 (+ ReLU) are deferred to the next convolution's operand path (training; see bn_on_operand)
        a1, _, _ = self._cbr_fwd(T, p + '.1', C[p + '.conv1'], B[p + '.bn1'], y, N, h, w, True,
                                 defer='bn1' in self.bn_operand_units)
        a2, h2, w2 = self._cbr_fwd(T, p + '.2', C[p + '.conv2'], B[p + '.bn2'], a1, N, h, w, True,
                                   defer='bn2' in self.bn_operand_units)
        if joined is not None:
            plan.wait_event(main_stream, joined)
        y, _, _ = self._cbr_fwd(T, p + '.3', C[p + '.conv3'], B[p + '.bn3'], a2, N, h2, w2, True, res=idt)
        return y, h2, w2

    def _block_bwd(self, T, blk, g, below):
        """Backward of one bottleneck block: g = d loss / d (block output) [N*h'*w'][4 planes] -> d loss / d (block input).
        below: (tape key, relu) of the unit that consumes the returned gradient (bn3 of the block before; None: the stem
        side), whose BatchNorm-backward reduction is folded into conv1's data-gradient epilogue."""
        C, B = self.convs, self.bns
        p, inpl, planes, stride, dil, ds = blk
        # identity blocks: the skip-path gradient g * [y3 > 0] is never written -- conv1's data-gradient epilogue
        # adds g gated by bn3's ReLU sign mask (when the unit kept one)
        g3, m3 = g, T[p + '.3'][6]
        gate_in_epilogue = (not ds) and (m3 is not None)
        da2, gm = self._cbr_bwd(T, p + '.3', C[p + '.conv3'], B[p + '.bn3'], g, True,
                                want_gmask=not gate_in_epilogue, consumer=(p + '.2', True))
        da1, _ = self._cbr_bwd(T, p + '.2', C[p + '.conv2'], B[p + '.bn2'], da2, True, consumer=(p + '.1', True))
        if ds:
            dxd, _ = self._cbr_bwd(T, p + '.d', C[p + '.downsample.0'], B[p + '.downsample.1'], gm, False)
            g, _ = self._cbr_bwd(T, p + '.1', C[p + '.conv1'], B[p + '.bn1'], da1, True, dx_res=dxd, consumer=below)
        elif gate_in_epilogue:
            g, _ = self._cbr_bwd(T, p + '.1', C[p + '.conv1'], B[p + '.bn1'], da1, True, dx_res=g3,
                                 dx_res_mask=m3, consumer=below)
        else:
            g, _ = self._cbr_bwd(T, p + '.1', C[p + '.conv1'], B[p + '.bn1'], da1, True, dx_res=gm, consumer=below)
        return g

    # ------------------------------------------------------------------ forward plan
    def _forward_plan(self, x, T):
        """x: one NCHW image batch, or a list of equally shaped batches that run through the network together
        as BatchNorm groups (the source and the target batch of an SSL step)."""
        dev = self.device
        main_stream = torch.cuda.current_stream()
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        Ng, _, H, W = xs[0].shape
        N = Ng * len(xs)
        assert T is None or T['groups'] == len(xs)
        C = self.convs
        B = self.bns
        H1, W1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        col_ready = None
        if W1 % 64 == 0 and self.fused_stem and 3 * H * W < (1 << 28):      # (what rgda_stem_conv / rgda_stem_wgrad serve)
            a0, col_ready = self._stem_fwd(T, xs, Ng, H, W, H1, W1, main_stream)
        else:
            col = torch.empty(N * H1 * W1, STEM_KP, dtype=BF, device=dev)
            for gi, xg in enumerate(xs):
                ops.stem_im2col(xg, col[gi * Ng * H1 * W1:(gi + 1) * Ng * H1 * W1], Ng, H, W, H1, W1)
            a0, _, _ = self._cbr_fwd(T, 'stem', C['encoder.resnet.conv1'], B['encoder.resnet.bn1'], col, N, H, W, True,
                                     wb=self.stem_wb, geom=(H1, W1))
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        y = torch.empty(N * H2 * W2, 64, dtype=BF, device=dev)
        idx = torch.empty(N * H2 * W2, 64, dtype=torch.uint8, device=dev)
        if isinstance(a0, _Lazy):
            ops.maxpool_fwd_bnin(self._bn_operand(T, a0), a0.c, y, idx, N, H1, W1, 64, H2, W2)
        else:
            ops.maxpool_fwd(a0, y, idx, N, H1, W1, 64, H2, W2)
        if T is not None:
            T['pool'] = (idx, (N, H1, W1, H2, W2))
        dbg = getattr(self, '_debug_taps', None)
        if dbg is not None:
            dbg['stem'] = a0.float().reshape(N, H1, W1, -1).permute(0, 3, 1, 2)       # (debug taps keep the stem materialised)
            dbg['pool'] = y.float().reshape(N, H2, W2, -1).permute(0, 3, 1, 2)
        h, w = H2, W2
        # a downsample branch (first block of a layer) only meets the main branch in bn3's residual add: in training it
        # runs on the head stream next to conv1 .. conv3
        bs = None
        if T is not None and self.parallel_heads and self.parallel_ds:
            if self._head_stream is None:
                self._head_stream = torch.cuda.Stream(device=dev)
            bs = self._head_stream
        for blk in self.blocks:
            y, h, w = self._block_fwd(T, blk, y, N, h, w, main_stream, bs)
            if dbg is not None:
                dbg[blk[0]] = y.float().reshape(N, h, w, -1).permute(0, 3, 1, 2)
        return self._heads_fwd(T, y, N, h, w, main_stream, col_ready)

    def _heads_fwd(self, T, y, N, h, w, main_stream, col_ready=None):
        """Instance norm + the two heads (regda/models/Encoder.py:123,146-155) on the pixel-major layer-4 output
        y [N*h*w][2048]: returns (logits of layer5, logits of layer6, feat); T = None: eval (no tape, no dropout)."""
        dev = self.device
        C, B = self.convs, self.bns
        dbg = getattr(self, '_debug_taps', None)
        HW, M = h * w, N * h * w
        xn = torch.empty(M, 2048, dtype=BF, device=dev)          # instance-normalised features, shared by both heads
        feat = torch.empty(N, 2048, h, w, device=dev) if T is not None else None   # eval returns probabilities only
        imi = torch.empty(N, 2, 2048, device=dev)
        ops.instnorm_fwd(y, xn, None, feat, imi, N, HW, 2048)
        if col_ready is not None:           # the stem's patch matrix (written beside the forward) is backward's operand
            plan.wait_event(main_stream, col_ready)
        def wait_head_weights():                                   # head weight slices rebuilt on another stream
            if self._hw_ready is not None:
                main_stream.wait_event(self._hw_ready)
                self._hw_ready = None
        plan.host(wait_head_weights)
        if T is not None:
            T['inorm'] = (y, imi, (N, h, w))
        if self.head_kind == 'aspp':
            x1, x2 = self._aspp_fwd(T, xn, N, h, w)
            if T is not None:
                T['out_shape'] = tuple(x1.shape)
            return x1, x2, feat
        mats = self._mats(h, w)
        if T is not None:
            # Dropout2d(0.1) keep-masks, scaled: drawn from torch's generator at every step (a host action of a plan)
            both = torch.empty(2, N, 512, device=dev)
            masks = [both[0], both[1]]

            def draw():
                if self._drop_override is not None:
                    for mk, m in zip(masks, self._drop_override):
                        mk.copy_(m.to(dev).float().repeat(N // m.shape[0], 1) / 0.9)
                elif torch.cuda.is_current_stream_capturing():
                    # inside a hipGraph capture a kernel argument is frozen: torch's graph-safe generator (its Philox offset
                    # advances at every replay) draws the masks there
                    both.uniform_().ge_(0.1).mul_(1.0 / 0.9)
                else:       # one kernel for both heads; the seed comes from torch's (host) generator, so torch.manual_seed governs it
                    ops.dropout_mask(both, 0.1, int(torch.randint(0, 2 ** 62, (1,)).item()))
            plan.host(draw)
        else:
            masks = [None, None]
        logits = []
        # both heads pool the same instance-normalised map: do it once
        pooled_all = [torch.empty(N * s * s, 2048, dtype=BF, device=dev) for s in POOL_SCALES]
        if self.factored_ppm:       # all four AdaptiveAvgPool2d in one read of xn: x direction, then y direction
            pm = self._pool_maps(h, w)
            prow = torch.empty(N * h * pm['R'], 2048, device=dev)
            ops.group_mix(xn, pm['Px'], prow, N * h, pm['R'], w, 2048)
            ops.sparse_mix([prow], pm['fwd'], pooled_all, N, 2048)
        else:
            for s, pooled in zip(POOL_SCALES, pooled_all):
                ops.spatial_mix(xn, mats[s][0], pooled, N, s * s, HW, 2048)
        # The two heads are independent: in training the second one runs on its own stream, so that its dozen small
        # launches (PPM branches, Z convs, mixes) overlap the first head's 280 us convolution instead of queueing
        # behind it.
        hs = None
        if T is not None and self.parallel_heads:
            if self._head_stream is None:
                self._head_stream = torch.cuda.Stream(device=dev)
            hs = self._head_stream
            caller = main_stream
            plan.wait_event(hs, plan.record_event(caller))
        for hi, head in enumerate(('layer5', 'layer6')):
            with (ops.use_stream(hs) if (hs is not None and hi == 1) else contextlib.nullcontext()):
                qs = []
                pre = [None] * len(POOL_SCALES)
                G = T['groups'] if T is not None else 1
                small = (T is not None and self.group_small_convs and self.small_bn and
                         all(2 <= N * s * s // G <= ops.BN_SMALL_MAX_ROWS for s in POOL_SCALES))
                if small:
                    # conv -> BatchNorm -> ReLU of the four branches (s x s maps): ONE launch for the four 2048 -> 512
                    # convolutions and one for the four BatchNorms, which sum the stored values themselves (no per-group
                    # convolution problems, no accumulators)
                    queue, bnq = [], []
                    for i, s in enumerate(POOL_SCALES):
                        cv, bn = C[f'{head}.ppm.{i}.1'], B[f'{head}.ppm.{i}.2']
                        Ms = N * s * s
                        cc = torch.empty(Ms, cv.co, dtype=BF, device=dev)
                        y = torch.empty(Ms, cv.co, dtype=BF, device=dev)
                        mi = torch.empty(G, 2, cv.co, device=dev)
                        rmask = (torch.empty(Ms, cv.co // 8, dtype=torch.uint8, device=dev)
                                 if (self.relu_sign_mask and cv.co >= self.relu_sign_mask) else None)
                        queue.append((pooled_all[i], cv.wb, cc, N, s, s, s, s, cv.k, cv.k, cv.stride, cv.pad, cv.dil))
                        bnq.append((cc, y, mi, bn.rm, bn.rv, bn.nbt, bn.gamma, bn.beta, Ms, cv.co, True, G, rmask))
                        T[f'{head}.ppm{i}'] = (pooled_all[i], cc, y, mi, (N, s, s, s, s), None, rmask)
                        qs.append(y)
                    ops.conv2d_grouped(queue)
                    ops.bn_train_small(bnq)
                elif T is not None and self.group_small_convs:
                    # the four branch convolutions (2048 -> 512 on s x s maps, one problem per scale and statistics group:
                    # 4 - 20 workgroups each, 32 K tiles) in ONE launch instead of eight in a row
                    queue = []
                    for i, s in enumerate(POOL_SCALES):
                        cv = C[f'{head}.ppm.{i}.1']
                        cc = torch.empty(N * s * s, cv.co, dtype=BF, device=dev)
                        st = T['stats_pool'].take(T['groups'] * NREP * 2 * cv.co)
                        self._conv_stats(pooled_all[i], cv.wb, cc, st, T['groups'], N, s, s, s, s, cv.k, cv.stride, cv.pad,
                                         cv.dil, queue=queue)
                        pre[i] = (cc, st)
                    ops.conv2d_grouped(queue)
                for i, s in enumerate(POOL_SCALES):
                    if small:
                        q = qs[i]
                    else:
                        q, _, _ = self._cbr_fwd(T, f'{head}.ppm{i}', C[f'{head}.ppm.{i}.1'], B[f'{head}.ppm.{i}.2'],
                                                pooled_all[i], N, s, s, True, preconv=pre[i])
                        qs.append(q)
                    if dbg is not None:
                        dbg[f'{head}.q{i}'] = q.float().reshape(N, s, s, -1).permute(0, 3, 1, 2)
                hid = self._head_last_fwd(T, head, xn, qs, N, h, w, masks[hi])
                if dbg is not None:
                    dbg[head + '.hidden'] = hid.float().reshape(N, h, w, -1).permute(0, 3, 1, 2)
                cl = C[f'{head}.conv_last.4']
                lg = torch.empty(N, self.num_classes, h, w, device=dev)
                ops.classifier_fwd(hid, cl.w.view(cl.co, cl.ci), cl.bias, lg, N, HW, 512, self.num_classes)
                if T is not None:
                    T[f'{head}.cls'] = (hid, (N, h, w))
                logits.append(lg)
        if hs is not None:
            plan.wait_event(caller, plan.record_event(hs))
        if T is not None:
            T['out_shape'] = tuple(logits[0].shape)
        return logits[0], logits[1], feat

    def _begin_backward(self, T, on_progress=None):
        """Per-step state of a backward pass on tape T (the queue of pending weight gradients, the statistic arena)."""
        T['on_progress'] = on_progress
        T['wgrad_pending'], T['wgrad_pending_flop'], T['wgrad_post'] = [], 0.0, []
        T['sums_pool'] = T['sums_pool_next']

    def _heads_bwd(self, T, g1, g2, gfeat, bwd_stream):
        """Backward of the two heads and the instance norm: d loss / d logits (N, classes, h, w) f32 of the two heads
        (+ optional d loss / d feat, bf16 pixel-major) -> d loss / d (layer-4 output), pixel-major bf16 [N*h*w][2048]."""
        dev = self.device
        C, B = self.convs, self.bns
        y4, imi, (N, h, w) = T['inorm']
        HW, M = h * w, N * h * w
        mats = self._mats(h, w) if self.head_kind == 'ppm' else None
        dfeat = None
        dpools = [None] * len(POOL_SCALES)
        dbg = getattr(self, '_debug_grads', None)
        tail_stream, tail_wgrads = None, []
        if self.head_kind == 'ppm' and self.parallel_heads and self.parallel_tails and T.get('wgrad_stream') is not None:
            if self._head_stream is None:
                self._head_stream = torch.cuda.Stream(device=dev)
            tail_stream = self._head_stream

        def nchw(t, hh, ww):
            return t.float().reshape(N, hh, ww, -1).permute(0, 3, 1, 2)
        if self.head_kind == 'aspp':
            g = torch.empty(M, 2048, dtype=BF, device=dev)
            ops.instnorm_bwd(self._aspp_bwd(T, g1, g2), gfeat, None, y4, imi, g, N, HW, 2048)
        for hi, (head, gl) in enumerate((('layer5', g1), ('layer6', g2)) if self.head_kind == 'ppm' else ()):
            hid, _ = T[f'{head}.cls']
            cl = C[f'{head}.conv_last.4']
            dh = torch.empty(M, 512, dtype=BF, device=dev)
            ops.classifier_bwd(hid, cl.w.view(cl.co, cl.ci), gl, dh, cl.g.view(cl.co, cl.ci),
                               cl.gbias, N, HW, 512, self.num_classes)
            # the second head's feature gradient is added onto the first head's in the conv epilogue
            dfeat, dc_head = self._head_last_bwd(T, head, dh, dfeat, tail=False)
            if dbg is not None:
                dbg[head + '.hidden'] = nchw(dh, h, w)

            def ppm_tail(head=head, dc_head=dc_head):
                dqs = self._head_last_bwd_tail(T, head, dc_head)
                queue = [] if self.group_small_convs else None
                bnq = [] if (self.group_small_convs and self.small_bn) else None
                for i, s in enumerate(POOL_SCALES):
                    # ... and so are the gradients of the shared pooled maps
                    dpools[i], _ = self._cbr_bwd(T, f'{head}.ppm{i}', C[f'{head}.ppm.{i}.1'], B[f'{head}.ppm.{i}.2'],
                                                 dqs[i], True, dx_res=dpools[i], conv_queue=queue, bn_queue=bnq)
                if bnq:
                    ops.bn_bwd_small(bnq)           # the four scales' BatchNorm backward (reduce + apply) in one launch
                if queue:
                    ops.conv2d_grouped(queue)       # the four scales' 512 -> 2048 data gradients in one launch
            if tail_stream is None and not (self.group_small_convs and self.small_bn):
                ppm_tail()                  # every unit's BatchNorm backward is launched inside its _cbr_bwd: flushes are safe there
                if T['wgrad_pending_flop'] >= self.wgrad_group_gflop * 1e9:
                    self._flush_wgrads(T)
            elif tail_stream is None:
                # the branches' weight-gradient operands (dc) are written by ops.bn_bwd_small at the END of the tail: they
                # are queued apart (no flush can fire inside the tail) and join the pending list once it has been launched
                pend, flop = T['wgrad_pending'], T['wgrad_pending_flop']
                mine = []
                T['wgrad_pending'], T['wgrad_pending_flop'] = mine, float('-inf')
                try:
                    ppm_tail()
                finally:
                    T['wgrad_pending'], T['wgrad_pending_flop'] = pend, flop
                T['wgrad_pending'].extend(mine)
                T['wgrad_pending_flop'] += sum(2.0 * it[3] * it[6] * it[7] * it[2].numel() for it in mine)
                if T['wgrad_pending_flop'] >= self.wgrad_group_gflop * 1e9:
                    self._flush_wgrads(T)
            else:
                # the PPM half of the head's backward (mixes, dq, the branches' BatchNorm and data gradients: ~90 us of launches
                # of <= 113 workgroups) on the head stream, under the OTHER head's 230 us convolution.  Its weight-gradient
                # operands are queued apart and join the pending list behind the stream join: a flush orders the weight-gradient
                # stream behind the MAIN stream only
                plan.wait_event(tail_stream, plan.record_event(bwd_stream))
                pend, flop = T['wgrad_pending'], T['wgrad_pending_flop']
                T['wgrad_pending'], T['wgrad_pending_flop'] = tail_wgrads, float('-inf')
                try:
                    with ops.use_stream(tail_stream):
                        ppm_tail()
                finally:
                    T['wgrad_pending'], T['wgrad_pending_flop'] = pend, flop
                if T['wgrad_pending_flop'] >= self.wgrad_group_gflop * 1e9:
                    self._flush_wgrads(T)
        if tail_stream is not None:
            plan.wait_event(bwd_stream, plan.record_event(tail_stream))
            T['wgrad_pending'].extend(tail_wgrads)
            T['wgrad_pending_flop'] += sum(2.0 * it[3] * it[6] * it[7] * it[2].numel() for it in tail_wgrads)
            if T['wgrad_pending_flop'] >= self.wgrad_group_gflop * 1e9:
                self._flush_wgrads(T)
        if self.head_kind == 'ppm':
            gpool = torch.empty(M, 2048, dtype=BF, device=dev)
            if self.factored_ppm:
                pm = self._pool_maps(h, w)
                prow = torch.empty(N * h * pm['R'], 2048, device=dev)
                ops.sparse_mix(dpools, pm['bwd'], prow, N, 2048)
                ops.group_mix(prow, pm['Pxt'], gpool, N * h, w, pm['R'], 2048)
            else:
                ops.spatial_mix_multi(dpools, [mats[s][1] for s in POOL_SCALES], gpool, N, HW, 2048)
            g = torch.empty(M, 2048, dtype=BF, device=dev)
            ops.instnorm_bwd(dfeat, gfeat, gpool, y4, imi, g, N, HW, 2048)
            del dfeat, gpool
        return g

    # ------------------------------------------------------------------ backward plan
    def _backward_plan(self, T, g1, g2, on_progress=None, gfeat=None):
        """g1, g2: d(loss)/d(logits) of the two heads (N, classes, h, w) f32.  gfeat (optional): d(loss)/d(feat) of the
        third forward output, bf16 pixel-major [N*h*w, 2048] -- the stage-2 prototype loss acts on the features."""
        dev = self.device
        bwd_stream = torch.cuda.current_stream()

        def wait_transposed_weights():                         # transposed weights rebuilt on another stream
            if getattr(self, '_wt_ready', None) is not None:
                bwd_stream.wait_event(self._wt_ready)
                self._wt_ready = None
        plan.host(wait_transposed_weights)
        self._begin_backward(T, on_progress)
        g = self._heads_bwd(T, g1, g2, gfeat, bwd_stream)
        C, B = self.convs, self.bns
        _, _, (N, h, w) = T['inorm']
        dbg = getattr(self, '_debug_grads', None)

        def nchw(t, hh, ww):
            return t.float().reshape(N, hh, ww, -1).permute(0, 3, 1, 2)
        self._progress(T, self._offset_of('layer5.' + self._head_first))
        hh, ww = h, w
        order = [b[0] for b in self.blocks]
        for bi in range(len(self.blocks) - 1, -1, -1):
            p, inpl, planes, stride, dil, ds = self.blocks[bi]
            if dbg is not None:
                dbg[p] = nchw(g, hh, ww)
                hh, ww = hh * stride, ww * stride
            # the gradient that leaves this block is consumed by bn3 of the block above it in the net
            below = (order[bi - 1] + '.3', True) if bi > 0 else None
            g = self._block_bwd(T, self.blocks[bi], g, below)
            self._progress(T, self._offset_of(p + '.conv1'))
            if bi > 0 and T['wgrad_pending']:
                stage = p.split('.')[-2]                      # 'encoder.resnet.layerK.i' -> 'layerK'
                if stage in self.wgrad_flush_after and self.blocks[bi - 1][0].split('.')[-2] != stage:
                    self._flush_wgrads(T)
        # layer1's queued weight gradients go to the second stream NOW, next to the stem's backward (max-pool, BN, its
        # own weight gradient: ~0.4 ms on this stream) -- flushed after it they were a tail the optimizer waited for
        if self.early_last_flush:
            self._flush_wgrads(T)
        idx, (N, H1, W1, H2, W2) = T['pool']
        ga0 = torch.empty(N * H1 * W1, 64, dtype=BF, device=dev)
        ops.maxpool_bwd(g, idx, ga0, N, H1, W1, 64, H2, W2)
        self._cbr_bwd(T, 'stem', C['encoder.resnet.conv1'], B['encoder.resnet.bn1'], ga0, True, stem=True)
        self._flush_wgrads(T)
        if T.get('mark') is not None:
            T['mark']('backward main chain done')
        if T.get('wgrad_stream') is not None:
            plan.wait_stream(T['main_stream'], T['wgrad_stream'])

    # ------------------------------------------------------------------ grads <-> torch
    def attach_grads(self):
        for name, par in self._views.items():
            par.grad = self._gviews[name]

    def _prepare_grads(self):
        """torch semantics: .grad accumulates; zero_grad(set_to_none=True) drops it.  If any .grad was
        dropped since our last backward the flat buffer is re-zeroed and every view re-attached."""
        if any(par.grad is None for par in self._views.values()):
            self.flat_g.zero_()
            self.attach_grads()

    # ------------------------------------------------------------------ public forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('regda_amd.Deeplabv2 runs on the GPU only')
        x = x.contiguous().float()
        self._maybe_sync()
        if self.training:
            if x.shape[0] < 2:
                raise ValueError('Expected more than 1 value per channel when training (PPM scale-1 BatchNorm)')
            return _ModelFn.apply(x, self._anchor, self)
        with torch.no_grad():
            x1, x2, _ = self._forward_plan(x, None)
            return ops.teacher_probs(x1, x2, tuple(x.shape[-2:]))     # Encoder.py:152-155


class _StatsPool:
    """One zero-initialised arena per forward for every BatchNorm's (sum, sumsq) accumulator (rgda_stat_t: 64-bit
    fixed point, order-independent totals -- include/rgda_hip.h)."""

    def __init__(self, buf):
        self.buf = buf              # zeroed by the caller (Deeplabv2.new_tape)
        self.off = 0

    def take(self, n):
        t = self.buf[self.off:self.off + n]
        self.off += (n + 63) // 64 * 64
        return t


class _ModelFn(torch.autograd.Function):
    """The drop-in nn.Module surface: (x1, x2, feat) = model(x) with all three outputs differentiable, like the
    reference's (Encoder.py:146-151) -- the stage-2 losses act on `feat`."""

    @staticmethod
    def forward(ctx, x, anchor, model):
        T = model.new_tape()
        x1, x2, feat = model._forward_plan(x, T)
        ctx.model, ctx.tape = model, T
        ctx.set_materialize_grads(False)
        return x1, x2, feat

    @staticmethod
    def backward(ctx, g1, g2, gfeat):
        model, T = ctx.model, ctx.tape
        model._prepare_grads()
        x1shape = T['out_shape']
        if g1 is None:
            g1 = torch.zeros(x1shape, device=model.device)
        if g2 is None:
            g2 = torch.zeros(x1shape, device=model.device)
        gf = None
        if gfeat is not None:       # NCHW fp32 -> pixel-major bf16 rows, the layout the InstanceNorm backward consumes
            gf = gfeat.permute(0, 2, 3, 1).reshape(-1, gfeat.shape[1]).to(BF).contiguous()
        model._backward_plan(T, g1.contiguous().float(), g2.contiguous().float(), gfeat=gf)
        T.clear()
        return None, None, None
