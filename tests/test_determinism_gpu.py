"""GPU: run-to-run reproducibility of the fused SSL step.

Every reduction of the step has an order that does not depend on which workgroup retires first: BatchNorm statistics
and BatchNorm-backward sums are accumulated as 64-bit fixed point with integer atomics (include/rgda_hip.h:
rgda_stat_t), a weight gradient receives at most two fp32 partial sums on a zeroed buffer (commutative), the prototype
sums walk the images in order, the classifier gradient and the gradient norm are two-stage reductions.  So two runs of
the same steps -- eager or replayed from a recorded plan, with or without stream overlap of the weight gradients --
are BIT-identical: weights, BatchNorm buffers, EMA shadow, prototypes, losses and the pseudo-label masks
(BASELINE.json north_star: "pseudo-label masks bit-identical"; tools/train_ssl_reg.py:210-223)."""
import pytest
import torch

from oracle import model as omodel

pytestmark = pytest.mark.gpu


def build(rt):
    from regda_amd.models.Encoder import Deeplabv2
    return Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                          cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                          inchannels=2048, num_classes=6, is_ins_norm=True))


def run_steps(rt, sd, batches, lrs, plan_from=None, ema=0.9, overlap=True, soft=False, size_seed=5, balancers=False):
    """-> everything a step leaves behind, cloned per step: losses, pseudo labels, refined soft input; and at the end
    the weights, buffers, momentum, shadow, prototypes."""
    from regda_amd.gast.balance import ClassBalance
    from regda_amd.ssl import SSLStep
    m = build(rt)
    m.load_state_dict(sd, strict=True)
    torch.manual_seed(77)               # Dropout2d masks come from the global generator
    kw = {}
    if balancers:
        kw = dict(class_balancer_s=ClassBalance(class_num=6, ignore_label=-1, decay=0.99, temperature=0.5),
                  class_balancer_t=ClassBalance(class_num=6, ignore_label=-1, decay=0.99, temperature=0.5))
    st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(size_seed)), ema_decay=ema,
                 overlap_wgrad=overlap, **kw)
    per_step = []
    for i, (b, lr) in enumerate(zip(batches, lrs)):
        s = b.get('soft_t') if soft else None
        if plan_from is not None and i == plan_from:
            st.record_plan(b['images_s'], b['label_s'], b['images_t'], s, b['regs_t'], lr=lr)
            out = st._out
        else:
            out = st.step(b['images_s'], b['label_s'], b['images_t'], s, b['regs_t'], lr)
        per_step.append(dict(loss_s=out[0].clone(), loss_t=out[1].clone(), gn=out[2].clone(), hard=st.last_hard.clone(),
                             soft=st.last_soft_t.clone()))
    torch.cuda.synchronize()
    final = dict(p=m.flat_p.clone(), buf=m.flat_buf.clone(), mom=st.mom.clone(), protos=st.prototypes.clone(),
                 shadow=None if st.teacher is None else st.teacher.flat_p.clone(), g=m.flat_g.clone())
    return per_step, final, st


def assert_identical(a, b, what):
    pa, fa = a[:2]
    pb, fb = b[:2]
    for i, (x, y) in enumerate(zip(pa, pb)):
        for k in x:
            assert torch.equal(x[k], y[k]), f'{what}: step {i}: {k} differs ' \
                f'(max abs {(x[k].double() - y[k].double()).abs().max().item():.3e})'
    for k in fa:
        if fa[k] is not None:
            assert torch.equal(fa[k], fb[k]), f'{what}: final {k} differs ' \
                f'({(fa[k] != fb[k]).sum().item()} of {fa[k].numel()} elements)'


def test_two_eager_runs_of_three_steps_are_bit_identical():
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=12)
    b1, b2 = make_batch(b=2, size=128, seed=21, with_soft=False), make_batch(b=2, size=128, seed=22, with_soft=False)
    seq, lrs = [b1, b2, b1], [1e-3, 2e-3, 1e-3]
    r1 = run_steps(rt, sd, seq, lrs)
    r2 = run_steps(rt, sd, seq, lrs)
    assert_identical(r1, r2, 'eager vs eager')
    # the steps did train (nothing here is trivially equal)
    assert not torch.equal(r1[1]['p'], build(rt).flat_p) and len({float(s['loss_s']) for s in r1[0]}) == 3
    # one stream (weight gradients behind the data gradients) and two streams give the same bits as well: the result
    # does not depend on what runs next to what
    r3 = run_steps(rt, sd, seq, lrs, overlap=False)
    assert_identical(r1, r3, 'two streams vs one stream')


def test_plan_replay_is_bit_identical_to_the_eager_step():
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=13)
    b1, b2 = make_batch(b=2, size=128, seed=31, with_soft=False), make_batch(b=2, size=128, seed=32, with_soft=False)
    seq, lrs = [b1, b1, b2, b1, b2], [1e-3, 2e-3, 1e-3, 3e-3, 1e-3]
    eager = run_steps(rt, sd, seq, lrs)
    planned = run_steps(rt, sd, seq, lrs, plan_from=1)       # step 1 is the recording, steps 2-4 are replays
    assert planned[2]._plan is not None and eager[2]._plan is None
    assert_identical(eager, planned, 'eager vs recorded plan')


def test_pseudo_label_masks_are_reproducible_on_resnet101_with_offline_soft_labels_and_class_balancing():
    """The deep topology (grouped weight-gradient launches, 33 bottlenecks of BatchNorm statistics), the reference's
    offline soft labels, --bcs / --bct class balancing: masks, losses and weights repeat bit for bit."""
    from regda_amd.synthetic import make_batch
    rt = 'resnet101'
    sd = omodel.init_state_dict(rt, 6, seed=14)
    b1, b2 = make_batch(b=2, size=128, seed=41), make_batch(b=2, size=128, seed=42)
    seq, lrs = [b1, b2, b1], [5e-4, 1e-3, 5e-4]
    r1 = run_steps(rt, sd, seq, lrs, ema=None, soft=True, balancers=True)
    r2 = run_steps(rt, sd, seq, lrs, ema=None, soft=True, balancers=True)
    assert_identical(r1, r2, 'resnet101 eager vs eager')
    hard = r1[0][-1]['hard']
    assert hard.shape == (2, 128, 128) and hard.min().item() >= -1 and hard.max().item() < 6


def test_full_size_config1_steps_are_bit_identical():
    """BASELINE config[1] at full size (ResNet-101, 8 + 8 images of 512 x 512, online EMA teacher, three streams): two
    runs of two steps, eager and recorded, give the same bits.  At this size the weight gradients of the stem and of
    layer 1 split K hundreds of ways (two-level combine), every convolution tile shape is in play and the step runs as it
    does in bench.py."""
    from regda_amd.synthetic import make_batch
    rt = 'resnet101'
    sd = omodel.init_state_dict(rt, 6, seed=15)
    for k in ('layer5.conv_last.4.weight', 'layer6.conv_last.4.weight'):
        sd[k] = sd[k] * 40.0                       # confident classifiers: pseudo labels pass the thresholds (as in bench.py)
    b = make_batch(b=8, size=512, seed=51, with_soft=False)
    seq, lrs = [b, b], [1e-3, 1e-3]
    r1 = run_steps(rt, sd, seq, lrs, ema=0.999)
    r2 = run_steps(rt, sd, seq, lrs, ema=0.999)
    assert_identical(r1, r2, 'full size, eager vs eager')
    r3 = run_steps(rt, sd, seq, lrs, ema=0.999, plan_from=1)
    assert_identical(r1, r3, 'full size, eager vs recorded plan')
    labelled = (r1[0][-1]['hard'] >= 0).float().mean().item()
    assert 0.05 < labelled < 1.0, labelled
