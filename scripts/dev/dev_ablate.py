"""dev: marginal cost of kernel classes in the OVERLAPPED step: time the step with one class of launches skipped against
a baseline measured right before it, each on a freshly built model (skipped launches leave garbage behind; only the
timing is meaningful, and NaN-poisoned buffers must not leak into the next measurement)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
cfg = dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
b = make_batch(b=8, size=512, seed=21, with_soft=False)
orig = {k: getattr(ops, k) for k in dir(ops)}
def fresh():
    torch.manual_seed(0)
    m = Deeplabv2(cfg)
    with torch.no_grad():
        for head in ('layer5', 'layer6'):
            m.convs[f'{head}.conv_last.4'].w.mul_(40.0)
    m.sync_weights()
    return SSLStep(m, torch.randn(6, 2048), ema_decay=0.999)
def run(st, n=8):
    for _ in range(4):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-4)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def restore():
    for k, v in orig.items():
        setattr(ops, k, v)
def small_filter(fn, thresh):
    def f(x, w, y, N, H, W, Ho, Wo, *a, **k):
        if N * Ho * Wo <= thresh:
            return None
        return fn(x, w, y, N, H, W, Ho, Wo, *a, **k)
    return f
cases = [('tiny fwd', None), ('tiny bwd', None), ('tiny teacher', None), ('tiny convs', None), ('tiny fwd', None), ('tiny bwd', None)]
cases_all = [('wgrad', ['conv2d_wgrad_grouped', 'conv2d_wgrad']), ('bn_train_apply', ['bn_train_apply']), ('bn_bwd_apply', ['bn_bwd_apply']),
         ('bn_bwd_reduce', ['bn_bwd_reduce']), ('sgd+sumsq', ['sgd_step', 'sumsq']), ('weight layouts', ['weight_transpose_batched']),
         ('mixes', ['group_mix', 'sparse_mix']), ('instnorm bwd', ['instnorm_bwd']), ('maxpool bwd', ['maxpool_bwd']),
         ('classifier bwd', ['classifier_bwd']), ('tiny convs', None), ('teacher forward', None)]
for label, names in cases:
    st = fresh()
    base = run(st)
    if names is not None:
        for n in names:
            setattr(ops, n, lambda *a, **k: None)
    elif label == 'tiny convs':
        for name in ('conv2d', 'conv2d_bneval', 'conv2d_bnbwd'):
            setattr(ops, name, small_filter(orig[name], 16 * 36))
    elif label == 'tiny teacher':
        setattr(ops, 'conv2d_bneval', small_filter(orig['conv2d_bneval'], 16 * 36))
    elif label in ('tiny fwd', 'tiny bwd'):
        want = 0 if label == 'tiny fwd' else 1
        def f(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode=0, *a, **k):
            if N * Ho * Wo <= 16 * 36 and mode == want:
                return None
            return orig['conv2d'](x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, *a, **k)
        ops.conv2d = f
        if want == 1:
            ops.conv2d_bnbwd = small_filter(orig['conv2d_bnbwd'], 16 * 36)
    else:
        soft = st.teacher_probs(b['images_t']).clone()
        st.teacher_probs = lambda images, snapshot=True: soft
    try:
        t = run(st)
    finally:
        restore()
    print('%-16s baseline %.3f  without %.3f  marginal %.3f ms' % (label, base, t, base - t))
    del st
    torch.cuda.empty_cache()
