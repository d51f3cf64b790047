"""dev: marginal cost of kernel classes in the OVERLAPPED step: time the step with one class of launches skipped
(results are wrong then; only the timing is meaningful)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
cfg = dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
m = Deeplabv2(cfg)
with torch.no_grad():
    for head in ('layer5', 'layer6'):
        m.convs[f'{head}.conv_last.4'].w.mul_(40.0)
m.sync_weights()
b = make_batch(b=8, size=512, seed=21, with_soft=False)
st = SSLStep(m, torch.randn(6, 2048), ema_decay=0.999)
def run(n=8):
    for _ in range(2):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
base = run()
print('baseline %.3f ms' % base)
orig = {k: getattr(ops, k) for k in dir(ops)}
def skip(names):
    for n in names:
        setattr(ops, n, lambda *a, **k: None)
def restore():
    for k, v in orig.items():
        setattr(ops, k, v)
for label, names in [('wgrad', ['conv2d_wgrad_grouped', 'conv2d_wgrad']), ('bn_train_apply', ['bn_train_apply']), ('bn_bwd_apply', ['bn_bwd_apply']),
                     ('bn_bwd_reduce', ['bn_bwd_reduce']), ('sgd+sumsq', ['sgd_step', 'sumsq']), ('weight layouts', ['weight_transpose_batched']),
                     ('label path', ['label_refine', 'pseudo_select', 'lrh', 'proto_update']),
                     ('mixes', ['group_mix', 'sparse_mix']), ('instnorm', ['instnorm_fwd', 'instnorm_bwd']), ('maxpool+im2col', ['maxpool_fwd', 'maxpool_bwd', 'stem_im2col']),
                     ('classifier', ['classifier_fwd', 'classifier_bwd']), ('upsample_ce', [])]:
    if not names:
        continue
    skip(names)
    try:
        t = run()
    finally:
        restore()
    print('without %-16s %.3f ms  (marginal %.3f)' % (label, t, base - t))
# the teacher forward
tp = st.teacher_probs
soft = tp(b['images_t']).clone()
st.teacher_probs = lambda images, snapshot=True: soft
print('without teacher forward  %.3f ms  (marginal %.3f)' % (run(), base - run()))
st.teacher_probs = tp
# convs by kernel size class: skip conv launches with few workgroups (the PPM branches)
def small_filter(fn, thresh):
    def f(x, w, y, N, H, W, Ho, Wo, *a, **k):
        if N * Ho * Wo <= thresh:
            return None
        return fn(x, w, y, N, H, W, Ho, Wo, *a, **k)
    return f
for name in ('conv2d', 'conv2d_bneval', 'conv2d_bnbwd'):
    setattr(ops, name, small_filter(orig[name], 16 * 36))
t = run(); restore()
print('without tiny convs (M <= 576) %.3f ms (marginal %.3f)' % (t, base - t))
print('baseline again %.3f ms' % run())
