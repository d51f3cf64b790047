"""Functional fp32 CPU restatement of the reference network.  TEST INFRASTRUCTURE ONLY.

  Deeplabv2.forward (PPM branch)  <- regda/models/Encoder.py:129,145-155
  Classifier_Module (ASPP heads)  <- regda/models/Encoder.py:68-84,111-114   (head='aspp': use_ppm=False)
  PPMBilinear.forward             <- regda/models/Encoder.py:43-55
  ResNetEncoder.forward (OS16)    <- regda/resnet.py:140-166,192-207
  Bottleneck.forward              <- regda/_resnets.py:92-112

It works on a plain dict in the reference's state_dict layout (688 keys for
ResNet-101) so the same weights can be fed to the
reference model, to this oracle and to the HIP path.
"""
import torch
import torch.nn.functional as F

LAYERS = {'resnet101': (3, 4, 23, 3), 'resnet50': (3, 4, 6, 3),
          'resnet17t': (2, 1, 1, 2)}   # resnet17t: test-only shallow topology (same code paths, 6 blocks)
POOL_SCALES = (1, 2, 3, 6)
ASPP_DILATIONS = (6, 12, 18, 24)       # dilation_series = padding_series, Encoder.py:111-114


def layer_specs(resnet_type='resnet101'):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] for every
    Bottleneck at output_stride 16 (layer4: stride 2 -> 1; first block conv2
    dilation 1, later blocks dilation 2; resnet.py:62-63,192-207)."""
    specs = []
    inpl = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), LAYERS[resnet_type]), start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li in (2, 3)) else 1
            dil = 2 if (li == 4 and bi > 0) else 1
            ds = bi == 0
            specs.append((f'encoder.resnet.layer{li}.{bi}', inpl, planes, stride, dil, ds))
            inpl = planes * 4
    return specs


def init_state_dict(resnet_type='resnet101', num_classes=6, seed=0, dtype=torch.float32, res_gamma=0.1,
                    ppm0_gamma=0.0, head='ppm'):
    """Seeded random weights in the reference layout: conv kaiming_normal(fan_out)
    (_resnets.py:164-169), BN gamma ~ U(0.5,1.5), beta ~ N(0,0.1) (non-trivial on
    purpose so affine terms are exercised), running stats (0,1).  The last BN of every
    residual branch (bn3) is scaled by `res_gamma`: with O(1) residual gains a 101-layer
    random net is chaotic (rounding the weights to bf16 alone moves the fp32 logits by
    ~70 %), with 0.1 it is as well conditioned as a trained network (~2 %), which is
    what a bf16-vs-fp32 parity tolerance can be stated against.
    ppm0_gamma scales the BN weight of the scale-1 PPM branch: that branch pools an
    instance-normalised map globally, i.e. its input is exactly 0 up to rounding noise in
    the reference itself, and its BatchNorm (2..8 samples) turns that noise into +-gamma;
    no two implementations (or two cuDNN algorithms) agree on it, so parity fixtures set
    gamma to 0 there (the branch then contributes relu(beta), a constant)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k, bias=False):
        std = (2.0 / (co * k * k)) ** 0.5
        sd[name + '.weight'] = torch.randn(co, ci, k, k, generator=g, dtype=dtype) * std
        if bias:
            sd[name + '.bias'] = torch.randn(co, generator=g, dtype=dtype) * 0.1

    def bn(name, c, scale=1.0):
        sd[name + '.weight'] = (torch.rand(c, generator=g, dtype=dtype) + 0.5) * scale
        sd[name + '.bias'] = torch.randn(c, generator=g, dtype=dtype) * 0.1
        sd[name + '.running_mean'] = torch.zeros(c, dtype=dtype)
        sd[name + '.running_var'] = torch.ones(c, dtype=dtype)
        sd[name + '.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)

    conv('encoder.resnet.conv1', 64, 3, 7)
    bn('encoder.resnet.bn1', 64)
    for p, inpl, planes, stride, dil, ds in layer_specs(resnet_type):
        conv(p + '.conv1', planes, inpl, 1); bn(p + '.bn1', planes)
        conv(p + '.conv2', planes, planes, 3); bn(p + '.bn2', planes)
        conv(p + '.conv3', planes * 4, planes, 1); bn(p + '.bn3', planes * 4, res_gamma)
        if ds:
            conv(p + '.downsample.0', planes * 4, inpl, 1); bn(p + '.downsample.1', planes * 4)
    for hd in ('layer5', 'layer6'):
        if head == 'aspp':
            for i in range(len(ASPP_DILATIONS)):
                sd[f'{hd}.conv2d_list.{i}.weight'] = torch.randn(num_classes, 2048, 3, 3, generator=g, dtype=dtype) * 0.01
                sd[f'{hd}.conv2d_list.{i}.bias'] = torch.randn(num_classes, generator=g, dtype=dtype) * 0.1
            continue
        for i in range(4):
            conv(f'{hd}.ppm.{i}.1', 512, 2048, 1); bn(f'{hd}.ppm.{i}.2', 512, ppm0_gamma if i == 0 else 1.0)
        conv(f'{hd}.conv_last.0', 512, 2048 + 4 * 512, 3); bn(f'{hd}.conv_last.1', 512)
        conv(f'{hd}.conv_last.4', num_classes, 512, 1, bias=True)
    return sd


def _bn(x, sd, name, training, new_stats):
    w, b = sd[name + '.weight'], sd[name + '.bias']
    rm, rv = sd[name + '.running_mean'], sd[name + '.running_var']
    if training:
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, 0.1, 1e-5)
        if new_stats is not None:
            new_stats[name + '.running_mean'] = rm2
            new_stats[name + '.running_var'] = rv2
            new_stats[name + '.num_batches_tracked'] = sd[name + '.num_batches_tracked'] + 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5)


def _rb(x):
    """Round to bf16 with a straight-through gradient (used by emulate_bf16)."""
    return x + (x.to(torch.bfloat16).float() - x).detach()


class _RoundBoth(torch.autograd.Function):
    """Round to bf16 in the forward pass AND round the gradient that flows back through this point to bf16: the HIP path
    keeps the activation gradients (data-gradient outputs, BatchNorm-backward outputs) in bf16 as well."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def aspp_head(x, weights, biases, dilations=ASPP_DILATIONS):
    """Classifier_Module.forward (Encoder.py:80-84): sum of Conv2d(3x3, padding = dilation = d, bias)(x)."""
    out = None
    for w, b, d in zip(weights, biases, dilations):
        t = F.conv2d(x, w, b, 1, d, d)
        out = t if out is None else out + t
    return out


def bottleneck(y, sd, spec, training=True, new_stats=None, rb=lambda t: t):
    """One bottleneck block, regda/_resnets.py:92-112 (conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3,
    + identity / downsample, relu).  spec: a row of layer_specs(); rb: the rounding applied where the HIP path stores bf16
    (identity for the fp32 reference semantics).  `forward` is made of these; the per-unit GPU test calls them directly."""
    p, inpl, planes, stride, dil, ds = spec
    idt = y
    o = rb(F.conv2d(y, sd[p + '.conv1.weight']))
    o = rb(F.relu(_bn(o, sd, p + '.bn1', training, new_stats)))
    o = rb(F.conv2d(o, sd[p + '.conv2.weight'], None, stride, dil, dil))
    o = rb(F.relu(_bn(o, sd, p + '.bn2', training, new_stats)))
    o = rb(F.conv2d(o, sd[p + '.conv3.weight']))
    o = _bn(o, sd, p + '.bn3', training, new_stats)
    if ds:
        idt = rb(F.conv2d(y, sd[p + '.downsample.0.weight'], None, stride))
        idt = rb(_bn(idt, sd, p + '.downsample.1', training, new_stats))
    return rb(F.relu(o + idt))


def heads(y, sd, training=True, drop_masks=None, new_stats=None, rb=lambda t: t, tap=lambda name, t: t, rb_tail=None):
    """Instance norm + the two heads on the layer-4 output y (regda/models/Encoder.py:8-65,68-84,123,146-151):
    returns ([logits of layer5, logits of layer6], feat)."""
    aspp = 'layer5.conv2d_list.0.weight' in sd
    if rb_tail is None:         # rounding of the last two stored tensors of a head (the 512-channel conv and its activation)
        rb_tail = rb
    feat = F.instance_norm(y, eps=1e-5)                       # Encoder.py:123,146-147
    tap('feat', feat)
    featq = rb(feat)
    outs = []
    for head in (('layer5', 'layer6') if aspp else ()):
        # Classifier_Module.forward (Encoder.py:80-84): the four dilated 3x3 convs, summed
        outs.append(aspp_head(featq, [sd[f'{head}.conv2d_list.{i}.weight'] for i in range(len(ASPP_DILATIONS))],
                              [sd[f'{head}.conv2d_list.{i}.bias'] for i in range(len(ASPP_DILATIONS))]))
    for hi, head in enumerate(() if aspp else ('layer5', 'layer6')):
        size = feat.shape[-2:]
        parts = [featq]
        for i, s in enumerate(POOL_SCALES):
            q = rb(F.adaptive_avg_pool2d(featq, s))
            q = rb(F.conv2d(q, sd[f'{head}.ppm.{i}.1.weight']))
            q = rb(F.relu(_bn(q, sd, f'{head}.ppm.{i}.2', training, new_stats)))
            tap(f'{head}.q{i}', q)
            parts.append(rb(F.interpolate(q, size, mode='bilinear', align_corners=False)))
        cat = torch.cat(parts, 1)
        tap(head + '.cat', cat)
        o = rb_tail(F.conv2d(cat, sd[f'{head}.conv_last.0.weight'], None, 1, 1))
        o = F.relu(_bn(o, sd, f'{head}.conv_last.1', training, new_stats))
        if training and drop_masks is not None:
            o = o * (drop_masks[hi].to(o.dtype) / 0.9)[:, :, None, None]
        o = rb_tail(o)
        tap(head + '.hidden', o)
        o = F.conv2d(o, sd[f'{head}.conv_last.4.weight'], sd[f'{head}.conv_last.4.bias'])
        outs.append(o)
    return outs, feat


def forward(sd, x, training=True, drop_masks=None, resnet_type='resnet101', new_stats=None,
            taps=None, emulate_bf16=False, emulate_where=None):
    """Train: (x1, x2, feat).  Eval: per-pixel class probabilities at input size.

    emulate_bf16: numerics model of the HIP path -- conv weights/inputs and every stored
    activation are rounded to bf16 at the points where regda_amd stores bf16 (conv output,
    BN/ReLU output, pooled / upsampled / instance-normalised tensors); accumulation stays
    fp32 and the rounding has a straight-through gradient.  A 101-layer random net turns a
    2^-9 relative perturbation into ~25 % gradient changes (measured, DESIGN.md "Parity"),
    so gradients of a bf16 pipeline can only be compared tightly against this model; the
    plain fp32 path (emulate_bf16=False) is the reference semantics.

    emulate_where: None = every storage point (the model of the HIP path); or a set out of {'weights', 'stem', 'layer1',
    'layer2', 'layer3', 'layer4', 'head', 'tail'} -- round ONLY there (attribution of the rounding noise to a storage
    point, tests/golden/attribute_teacher_noise.py; 'tail' = the heads' 512-channel conv output and its activation).

    drop_masks: optional (m5, m6), each (b,512) of {0,1}: the Dropout2d(0.1)
    channel keep-masks of the two heads (Encoder.py:39); kept channels are
    scaled by 1/0.9.  None -> no dropout (identity), which is what eval does.
    new_stats: dict that receives updated BN buffers (train mode).
    taps: optional dict that receives named intermediate activations.
    """
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    # emulate_bf16 = 'grad': the activation gradients are rounded at the same points too (tests/golden/derive_tolerances.py)
    rb = (_RoundBoth.apply if emulate_bf16 == 'grad' else _rb) if emulate_bf16 else (lambda t: t)
    aspp = 'layer5.conv2d_list.0.weight' in sd
    ident = lambda t: t
    at = (lambda where: rb) if emulate_where is None else (lambda where: rb if where in emulate_where else ident)
    if emulate_bf16 and at('weights') is not ident:
        sd = {k: (_rb(v) if (v.dim() == 4 and 'conv_last.4' not in k) else v) for k, v in sd.items()}
        x = _rb(x)
    y = at('stem')(F.conv2d(x, sd['encoder.resnet.conv1.weight'], None, 2, 3))
    y = at('stem')(F.relu(_bn(y, sd, 'encoder.resnet.bn1', training, new_stats)))
    tap('stem', y)
    y = F.max_pool2d(y, 3, 2, 1)
    tap('pool', y)
    for spec in layer_specs(resnet_type):
        y = bottleneck(y, sd, spec, training, new_stats, at(spec[0].split('.')[2]))      # 'encoder.resnet.layerN.i'
        tap(spec[0], y)
    outs, feat = heads(y, sd, training, drop_masks, new_stats, at('head'), tap, rb_tail=at('tail'))
    if training:
        return outs[0], outs[1], feat
    x1 = F.interpolate(outs[0], x.shape[-2:], mode='bilinear', align_corners=True)
    x2 = F.interpolate(outs[1], x.shape[-2:], mode='bilinear', align_corners=True)
    return (x1.softmax(dim=1) + x2.softmax(dim=1)) / 2


def param_names(sd):
    """Keys that are nn.Parameters in the reference (everything but BN buffers)."""
    return [k for k in sd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
