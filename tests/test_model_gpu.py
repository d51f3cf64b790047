"""GPU: the HIP Deeplabv2 (regda_amd.models.Encoder) against the oracle and the reference goldens.

Tolerance statement (DESIGN.md "Parity"): activations are stored in bf16, accumulation is fp32.  A
randomly initialised 101-layer BN network is chaotic -- rounding only the conv weights to bf16 moves the
fp32 oracle's logits by ~2 % and its gradients by ~25 % (measured) -- so
  * forward (logits, feat), loss, global gradient norm and direction are bounded in units of the fixture's bf16
    ROUNDING NOISE N = |bf16-emulating oracle - fp32 oracle|, computed on the CPU (tests/golden/derive_tolerances.py ->
    bf16_tolerances.json; shallow topology: 2.3-2.9 % of the logits, ResNet-101: 5.2-7.0 %): 3 N against the plain
    fp32 oracle / the reference golden (another realisation of the same noise), 1.5 N against the bf16-EMULATING
    oracle (same rounding points: correlated realisations);
  * per-layer backward wiring is checked tightly by re-running every layer's backward in torch from
    the tensors the HIP path itself saved (test_layerwise_backward_consistency).
Integer outputs of the label path given identical inputs stay bit-exact (tests/test_label_gpu.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import labelpath as opath
from oracle import model as omodel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
# rounding-noise units of each fixture (tests/golden/derive_tolerances.py, CPU): N = |bf16-emulating oracle - fp32 oracle|
_TOL = json.load(open(os.path.join(GOLD, 'bf16_tolerances.json')))


def tolN(fixture, key, units, floor=1e-3):
    return max(units * _TOL[fixture][key], floor)


def tol_gn(fixture, units):
    """Gradient norm: `units` N, or the length uncertainty (1 - cos theta) / 2 of a vector whose direction carries
    rounding noise of angle theta (tests/golden/derive_tolerances.py)."""
    return max(units * _TOL[fixture]['grad_norm'], 0.5 * (1.0 - _TOL[fixture]['grad_cos']))


def build(rt):
    from regda_amd.models.Encoder import Deeplabv2
    return Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                          cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                          inchannels=2048, num_classes=6, is_ins_norm=True))


def l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.fixture(scope='module')
def r101():
    m = build('resnet101')
    sd = omodel.init_state_dict('resnet101', 6, seed=1)
    m.load_state_dict(sd, strict=True)
    return m, sd


def test_state_dict_layout_matches_reference(r101):
    m, sd = r101
    with open(os.path.join(GOLD, 'state_dict_manifest.json')) as f:
        man = json.load(f)
    got = m.state_dict()
    assert [k for k, _, _ in man] == list(got.keys())
    for (k, shape, dt) in man:
        assert list(got[k].shape) == shape and str(got[k].dtype).replace('torch.', '') == dt, k
    assert sum(p.numel() for p in m.parameters()) == 88653900
    # round trip: what we save is what the reference layout expects, bit for bit
    for k, v in sd.items():
        assert torch.equal(got[k].cpu(), v), k
    names = [n for n, _ in m.named_parameters()]
    assert names == omodel.param_names(sd)


def _run_case(m, sd, rt, xs, lab, masks, emulate):
    m.train()
    m.load_state_dict(sd, strict=True)
    m.set_drop_masks(*masks)
    names = omodel.param_names(sd)
    sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    r1, r2, rf = omodel.forward(sdr, xs, True, masks, rt, {}, None, emulate_bf16=emulate)
    lref = opath.loss_calc([r1, r2], lab, -1)
    gref = torch.autograd.grad(lref, [sdr[k] for k in names])
    from regda_amd.gast.balance import CrossEntropy
    from regda_amd.utils.tools import loss_calc
    m.zero_grad(set_to_none=True)
    x1, x2, feat = m(xs.cuda())
    loss = loss_calc([x1, x2], lab.cuda(), CrossEntropy(-1), multi=True)
    loss.backward()
    named = dict(m.named_parameters())
    keep = [k for k in names if 'ppm.0.' not in k]          # degenerate branch, see oracle.model.init_state_dict
    a = torch.cat([named[k].grad.float().cpu().reshape(-1) for k in keep])
    b = torch.cat([g.reshape(-1) for k, g in zip(names, gref) if k in keep])
    return dict(x1=l2(x1, r1), x2=l2(x2, r2), feat=l2(feat, rf), loss=(loss.item(), lref.item()),
                cos=(a @ b / (a.norm() * b.norm())).item(), gn=(a.norm().item(), b.norm().item()))


def test_shallow_topology_forward_backward():
    rt = 'resnet17t'
    m = build(rt)
    sd = omodel.init_state_dict(rt, 6, seed=1)
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(4, 3, 128, 128, generator=gen)
    masks = ((torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8), (torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8))
    lab = torch.from_numpy(np.kron(np.random.default_rng(0).integers(-1, 6, size=(4, 8, 8)), np.ones((16, 16), np.int64)))
    F_ = 'shallow_model'
    for emulate, u in ((True, 1.5), (False, 3.0)):
        r = _run_case(m, sd, rt, xs, lab, masks, emulate=emulate)
        print('\n[shallow model vs %s oracle]' % ('emulating' if emulate else 'fp32'), r)
        assert r['x1'] < tolN(F_, 'x1', u) and r['x2'] < tolN(F_, 'x2', u) and r['feat'] < tolN(F_, 'feat', u), r
        assert r['loss'][0] == pytest.approx(r['loss'][1], rel=tolN(F_, 'loss', u, floor=2e-3)), r
        assert r['cos'] > 1 - u * (1 - _TOL[F_]['grad_cos']) and r['gn'][0] == pytest.approx(r['gn'][1], rel=tol_gn(F_, u)), r


def test_resnet101_forward_backward_vs_oracle(r101, gold):
    m, sd = r101
    g = gold('model_small.npz')
    xs = torch.from_numpy(g['xs'])
    masks = (torch.from_numpy(g['m5'][0]), torch.from_numpy(g['m6'][0]))
    lab = torch.from_numpy(g['lab_s'].astype(np.int64))
    r = _run_case(m, sd, 'resnet101', xs, lab, masks, emulate=True)
    print('\n[resnet101 model vs emulating oracle]', r)
    F_, u = 'resnet101_model', 1.5
    assert r['x1'] < tolN(F_, 'x1', u) and r['x2'] < tolN(F_, 'x2', u) and r['feat'] < tolN(F_, 'feat', u), r
    assert r['loss'][0] == pytest.approx(r['loss'][1], rel=tolN(F_, 'loss', u)), r
    assert r['cos'] > 1 - u * (1 - _TOL[F_]['grad_cos']) and r['gn'][0] == pytest.approx(r['gn'][1], rel=tol_gn(F_, u)), r


def test_resnet101_vs_reference_golden(r101, gold):
    """Logits / feat / BN buffers / teacher probabilities against the REFERENCE's own outputs."""
    m, sd = r101
    g = gold('model_small.npz')
    m.load_state_dict(sd, strict=True)
    m.train()
    m.set_drop_masks(torch.from_numpy(g['m5'][0]), torch.from_numpy(g['m6'][0]))
    with torch.no_grad():
        s1, s2, fs = m(torch.from_numpy(g['xs']).cuda())
        m.set_drop_masks(torch.from_numpy(g['m5'][1]), torch.from_numpy(g['m6'][1]))
        t1, t2, ft = m(torch.from_numpy(g['xt']).cuda())
    assert l2(s1, torch.from_numpy(g['s1'])) < 0.10 and l2(s2, torch.from_numpy(g['s2'])) < 0.10
    assert l2(t1, torch.from_numpy(g['t1'])) < 0.10 and l2(ft[:, :32], torch.from_numpy(g['feat_t'])) < 0.10
    sdn = m.state_dict()
    # two train-mode forwards -> running stats updated twice, like the reference
    assert int(sdn['encoder.resnet.bn1.num_batches_tracked']) == int(g['nbt']) == 2
    np.testing.assert_allclose(sdn['encoder.resnet.bn1.running_mean'].cpu().numpy(), g['bn1_rm'], rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(sdn['encoder.resnet.bn1.running_var'].cpu().numpy(), g['bn1_rv'], rtol=2e-2, atol=2e-3)
    assert l2(sdn['layer5.conv_last.1.running_var'], torch.from_numpy(g['l5bn_rv'])) < 0.10
    m.eval()
    probs = m(torch.from_numpy(g['xt']).cuda())
    assert probs.shape == (2, 6, 64, 64)
    # eval mode with running statistics two updates away from their (0, 1) initialisation: large, saturating
    # logits; compare the class decision and the mean probability error rather than the worst pixel
    ref = torch.from_numpy(g['probs'])
    assert (probs.cpu() - ref).abs().mean().item() < 0.02
    assert (probs.cpu().argmax(1) == ref.argmax(1)).float().mean().item() > 0.9
    torch.testing.assert_close(probs.sum(1).cpu(), torch.ones(2, 64, 64), rtol=1e-5, atol=1e-5)


def test_eval_branch_on_warm_statistics_vs_reference_golden(capsys):
    """The eval branch -- the teacher's output, Encoder.py:152-155 -- against the REFERENCE's own probabilities on WARM
    BatchNorm statistics (tests/golden/eval_warm.npz: the reference ResNet-101 after 20 train-mode forwards, every
    BatchNorm buffer of that state loaded here, so nothing but the eval path is compared).  `model_small.npz`'s eval output
    (test above) sits two updates from the (0, 1) initialisation where the logits saturate and only a loose bound holds.
    Bounds: three rounding-noise units of this fixture (bf16_tolerances.json "resnet101_eval_warm": the bf16-emulating
    oracle against the fp32 oracle on the CPU -- mean |dp| N = 2.7e-3; over ALL pixels bf16 storage itself leaves only ~97 %
    argmax agreement on a random-init net whatever the classifier gain, so the class decision is asserted where the
    reference is decisive: top-1 minus top-2 probability > 0.05)."""
    import sys
    sys.path.insert(0, GOLD)
    import derive_tolerances as D
    F_ = 'resnet101_eval_warm'
    sd, xe, ref = D.eval_warm_inputs()
    m = build('resnet101')
    m.load_state_dict(sd, strict=True)
    m.eval()
    with torch.no_grad():
        probs = m(xe.cuda()).cpu()
    torch.testing.assert_close(probs.sum(1), torch.ones(1, 128, 128), rtol=1e-5, atol=1e-5)
    r = D.eval_agreement(probs, ref)
    with capsys.disabled():
        print('\n[eval branch on warm statistics vs the reference]', {k: '%.4g' % v for k, v in r.items()},
              ' rounding model N:', {k: '%.4g' % v for k, v in _TOL[F_].items()})
    assert r['mean_abs'] < tolN(F_, 'mean_abs', 3.0)
    assert 1.0 - r['argmax_agree'] < tolN(F_, 'argmax_disagree', 3.0, floor=0.01)
    assert r['decisive_disagree'] < tolN(F_, 'decisive_disagree', 3.0, floor=2e-3)
    assert r['decisive_share'] > 0.2


def test_layerwise_backward_consistency():
    """Every conv+BN(+ReLU) unit of the shallow net: recompute its backward with torch from the tensors the
    HIP path saved on its tape and compare dX, dW, dgamma, dbeta tightly (one layer of bf16 rounding)."""
    from regda_amd.models import Encoder as E
    rt = 'resnet17t'
    m = build(rt)
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=2), strict=True)
    m.train()
    m.wgrad_group_gflop = 0.0       # one weight-gradient launch per layer, so the spy sees each layer's dW at once
    m.group_small_convs = False     # ... and every data gradient when its unit returns (not queued for a grouped launch)
    rec = []
    orig = E.Deeplabv2._cbr_bwd

    def spy(self, T, key, conv, bn, g, relu, need_dx=True, want_gmask=False, dx_res=None, stem=False, consumer=None,
            dx_res_mask=None, conv_queue=None, bn_queue=None):
        assert conv_queue is None and bn_queue is None
        x, c, y, mi, dims, nscale, rmask = T[key]
        g0 = conv.g.clone()
        dg0, db0 = bn.dgamma.clone(), bn.dbeta.clone()
        out = orig(self, T, key, conv, bn, g, relu, need_dx, want_gmask, dx_res, stem, consumer, dx_res_mask)
        if not stem and nscale is None:
            res_eff = dx_res
            if dx_res is not None and dx_res_mask is not None:      # residual gated by a ReLU sign mask in the epilogue
                bits = ((dx_res_mask.unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8, device='cuda')) & 1).bool()
                res_eff = torch.where(bits.reshape(dx_res.shape), dx_res, torch.zeros_like(dx_res))
            rec.append((key, conv, bn, x.clone(), c.clone(), y.clone(), g.clone(), dims, relu,
                        None if res_eff is None else res_eff.clone(), None if out[0] is None else out[0].clone(),
                        (conv.g - g0).clone(), (bn.dgamma - dg0).clone(), (bn.dbeta - db0).clone()))
        return out
    E.Deeplabv2._cbr_bwd = spy
    try:
        gen = torch.Generator().manual_seed(5)
        x1, x2, _ = m(torch.randn(2, 3, 64, 64, generator=gen).cuda())
        (x1.square().mean() + x2.mean()).backward()
    finally:
        E.Deeplabv2._cbr_bwd = orig
    assert len(rec) >= 20
    for (key, conv, bn, x, c, y, g, (N, H, W, Ho, Wo), relu, dx_res, dx, dW, dgam, dbet) in rec:
        xt = x[:, :conv.ci].float().reshape(N, H, W, conv.ci).permute(0, 3, 1, 2).cpu().requires_grad_(True)
        wt = conv.wb.float().reshape(conv.co, conv.k, conv.k, conv.ci).permute(0, 3, 1, 2).cpu().requires_grad_(True)
        gam, bet = bn.gamma.detach().cpu().clone().requires_grad_(True), bn.beta.detach().cpu().clone().requires_grad_(True)
        co = F.conv2d(xt, wt, None, conv.stride, conv.pad, conv.dil)
        craw = c.float().reshape(N, Ho, Wo, conv.co).permute(0, 3, 1, 2).cpu()
        co = co + (craw - co).detach()                       # use the stored (bf16) conv output for BN
        o = F.batch_norm(co, None, None, gam, bet, True, 0.1, 1e-5)
        yt = y.float().reshape(N, Ho, Wo, conv.co).permute(0, 3, 1, 2).cpu()
        gt = g.float().reshape(N, Ho, Wo, conv.co).permute(0, 3, 1, 2).cpu()
        if relu:
            gt = gt * (yt > 0)
        o.backward(gt)
        assert l2(dgam, gam.grad) < 2e-2 or (dgam.cpu() - gam.grad).abs().max() < 1e-3, key
        assert l2(dbet, bet.grad) < 2e-2 or (dbet.cpu() - bet.grad).abs().max() < 1e-3, key
        dWt = dW.reshape(conv.co, conv.k, conv.k, conv.ci).permute(0, 3, 1, 2)
        assert l2(dWt, wt.grad) < 3e-2, key
        if dx is not None:
            ref = xt.grad
            if dx_res is not None:
                ref = ref + dx_res.float().reshape(N, H, W, conv.ci).permute(0, 3, 1, 2).cpu()
            assert l2(dx.float().reshape(N, H, W, conv.ci).permute(0, 3, 1, 2), ref) < 3e-2, key


def test_grouped_forward_backward_equals_two_separate_passes():
    """The fused SSL step runs the source and the target batch through the network together as two
    BatchNorm groups; results (logits, BN buffers, gradients) must equal two separate passes."""
    rt = 'resnet17t'
    m = build(rt)
    sd = omodel.init_state_dict(rt, 6, seed=4)
    gen = torch.Generator().manual_seed(9)
    xs, xt = torch.randn(2, 3, 64, 64, generator=gen).cuda(), (torch.randn(2, 3, 64, 64, generator=gen) * 2 + 1).cuda()
    g1 = torch.randn(4, 6, 4, 4, generator=gen).cuda()
    g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda()
    ones = torch.ones(2, 512)
    # (a) two separate passes
    m.load_state_dict(sd, strict=True)
    m.train()
    m.set_drop_masks(ones, ones)
    m.flat_g.zero_()
    Ta, Tb = m.new_tape(), m.new_tape()
    with torch.no_grad():
        a1, a2, fa = m._forward_plan(xs, Ta)
        b1, b2, fb = m._forward_plan(xt, Tb)
        m._backward_plan(Ta, g1[:2], g2[:2])
        m._backward_plan(Tb, g1[2:], g2[2:])
    ga = m.flat_g.clone()
    bufa = m.flat_buf.clone()
    # (b) one grouped pass
    m.load_state_dict(sd, strict=True)
    m.flat_g.zero_()
    T = m.new_tape(groups=2)
    with torch.no_grad():
        c1, c2, fc = m._forward_plan([xs, xt], T)
        m._backward_plan(T, g1, g2)
    # not bit-identical: the fp32 statistics of the second group are summed in a different order
    # (different tile -> XCD replica mapping), a 1e-10 perturbation that the deep net amplifies to < 2 %
    assert l2(c1[:2], a1) < 2e-2 and l2(c1[2:], b1) < 2e-2 and l2(c2[2:], b2) < 2e-2
    assert l2(fc[:2], fa) < 2e-2 and l2(fc[2:], fb) < 2e-2
    assert l2(m.flat_buf, bufa) < 1e-3                       # running statistics: src update then tgt update
    assert int(m.state_dict()['encoder.resnet.bn1.num_batches_tracked']) == 2
    cos = (m.flat_g @ ga / (m.flat_g.norm() * ga.norm())).item()
    assert cos > 0.99 and abs(m.flat_g.norm().item() / ga.norm().item() - 1) < 0.03


def test_relu_sign_mask_mode_gives_identical_gradients():
    """relu_sign_mask=True keeps one bit per activation for the backward pass instead of re-reading y: same logits
    and gradients as the y-reading mode, up to the run-to-run noise of atomic summation order."""
    rt = 'resnet17t'
    m = build(rt)
    sd = omodel.init_state_dict(rt, 6, seed=6)
    gen = torch.Generator().manual_seed(13)
    x = [torch.randn(2, 3, 64, 64, generator=gen).cuda(), torch.randn(2, 3, 64, 64, generator=gen).cuda()]
    g1, g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda(), torch.randn(4, 6, 4, 4, generator=gen).cuda()
    ones = torch.ones(2, 512)
    out = []
    default = m.relu_sign_mask
    for flag in (False, True):
        m.load_state_dict(sd, strict=True)
        m.train()
        m.relu_sign_mask = flag
        m.set_drop_masks(ones, ones)
        m.flat_g.zero_()
        T = m.new_tape(groups=2)
        with torch.no_grad():
            c1, c2, f = m._forward_plan(x, T)
            m._backward_plan(T, g1, g2)
        out.append((c1.clone(), c2.clone(), m.flat_g.clone()))
    m.relu_sign_mask = default
    # the mask carries exactly the bits [y > 0] (tests/test_conv_gpu.py pins that per kernel); two passes over the net
    # are still not bit-identical because the fp32 BN statistics are accumulated with atomics in a varying order and
    # the deep net amplifies that last-bit noise (same bound as the grouped-vs-separate test above)
    assert l2(out[1][0], out[0][0]) < 2e-2 and l2(out[1][1], out[0][1]) < 2e-2
    ga, gb = out[0][2], out[1][2]
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos > 0.99 and abs(gb.norm().item() / ga.norm().item() - 1) < 0.03


def test_grouped_small_convolutions_give_identical_bits():
    """group_small_convs: the PPM branches' convolutions in shared launches (rgda_conv2d_grouped) run the same kernel over the
    same tiles as one launch each -- logits, features and every gradient are BIT-identical (the fixed-point BatchNorm
    statistics land in other replicas, their totals are the same integers), at one and at two statistics groups."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=4)
    gen = torch.Generator().manual_seed(21)
    x = [torch.randn(2, 3, 64, 64, generator=gen).cuda(), torch.randn(2, 3, 64, 64, generator=gen).cuda()]
    g1, g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda(), torch.randn(4, 6, 4, 4, generator=gen).cuda()
    ones = torch.ones(2, 512)
    for groups in (2, 1):
        out = []
        for flag in (True, False):
            m = build(rt)
            m.group_small_convs = flag
            m.small_bn = False              # (the small-map BatchNorm kernels sum in another order: next test)
            m.load_state_dict(sd, strict=True)
            m.train()
            m.set_drop_masks(ones, ones)
            m.flat_g.zero_()
            T = m.new_tape(groups=groups)
            with torch.no_grad():
                c1, c2, f = m._forward_plan(x if groups == 2 else torch.cat(x), T)
                m._backward_plan(T, g1, g2)
            torch.cuda.synchronize()
            out.append((c1.clone(), c2.clone(), f.clone(), m.flat_g.clone()))
        for a, b in zip(*out):
            assert torch.equal(a, b)
        assert float(out[0][3].abs().sum()) > 0


def test_small_map_batchnorm_path_matches_the_general_kernels():
    """small_bn: the PPM branches' BatchNorms through rgda_bn_train_small / rgda_bn_bwd_small (statistics summed by the
    kernel itself from the stored values) against the general path (accumulators of the convolution epilogue, reduce +
    apply): same logits and gradients up to the rounding of another summation order, running statistics included."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=4)
    gen = torch.Generator().manual_seed(22)
    x = [torch.randn(2, 3, 64, 64, generator=gen).cuda(), torch.randn(2, 3, 64, 64, generator=gen).cuda()]
    g1, g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda(), torch.randn(4, 6, 4, 4, generator=gen).cuda()
    ones = torch.ones(2, 512)
    out = []
    for flag in (True, False):
        m = build(rt)
        m.small_bn = flag
        m.load_state_dict(sd, strict=True)
        m.train()
        m.set_drop_masks(ones, ones)
        m.flat_g.zero_()
        T = m.new_tape(groups=2)
        with torch.no_grad():
            c1, c2, f = m._forward_plan(x, T)
            m._backward_plan(T, g1, g2)
        torch.cuda.synchronize()
        bnb = {k: v.clone() for k, v in m.state_dict().items() if 'ppm' in k and ('running' in k or 'num_batches' in k)}
        out.append((c1.clone(), c2.clone(), m.flat_g.clone(), bnb))
    (a1, a2, ga, ba), (b1, b2, gb, bb) = out
    assert l2(a1, b1) < 5e-3 and l2(a2, b2) < 5e-3
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos > 0.999 and ga.norm().item() == pytest.approx(gb.norm().item(), rel=5e-3)
    assert len(ba) >= 8 * 3
    for k in ba:
        torch.testing.assert_close(ba[k].float(), bb[k].float(), rtol=1e-4, atol=1e-6, msg=k)


def test_factored_ppm_maps_match_the_one_pass_maps():
    """The separable two-stage form of the PPM heads' tap-shifted bilinear maps (csrc/mix_kernels.hip) against the
    one-pass sparse maps it replaces: same logits and same gradients up to bf16 rounding of the stored tensors
    (forward BN statistics are accumulated with atomics, so not even two runs of ONE path agree bitwise)."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=1)
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(4, 3, 128, 128, generator=gen).cuda()
    masks = (torch.ones(4, 512), torch.ones(4, 512))
    outs = []
    for factored in (True, False):
        m = build(rt)
        m.factored_ppm = factored
        m.load_state_dict(sd, strict=True)
        m.train()
        m.set_drop_masks(*masks)
        x1, x2, feat = m(xs)
        (x1.square().mean() + x2.square().mean()).backward()
        outs.append((x1.detach(), x2.detach(), m.flat_g.clone()))
    (a1, a2, ga), (b1, b2, gb) = outs
    assert l2(a1, b1) < 2e-2 and l2(a2, b2) < 2e-2
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos > 0.99 and ga.norm().item() == pytest.approx(gb.norm().item(), rel=3e-2)       # the mask test's bound


def test_feature_output_is_differentiable_through_the_module_api():
    """d(loss on feat)/dW through `model(x)` + torch.autograd (the stage-2 losses of tools/train_align_reg.py act on
    the third output): against autograd on the bf16-emulating oracle, and against the logits-only gradient (must
    differ).  The reference's `feat` is an ordinary differentiable output (Encoder.py:146-151)."""
    rt = 'resnet17t'
    m = build(rt)
    sd = omodel.init_state_dict(rt, 6, seed=5)
    m.load_state_dict(sd, strict=True)
    m.train()
    gen = torch.Generator().manual_seed(17)
    xs = torch.randn(2, 3, 64, 64, generator=gen)
    R = torch.randn(2, 2048, 4, 4, generator=gen)
    ones = (torch.ones(2, 512), torch.ones(2, 512))
    m.set_drop_masks(*ones)
    names = omodel.param_names(sd)
    backbone = [k for k in names if k.startswith('encoder.')]

    def grads(use_feat, use_logits):
        m.zero_grad(set_to_none=True)
        x1, x2, feat = m(xs.cuda())
        assert feat.requires_grad
        loss = 0
        if use_feat:
            loss = loss + (feat * R.cuda()).mean()
        if use_logits:
            loss = loss + x1.square().mean() + x2.mean()
        loss.backward()
        named = dict(m.named_parameters())
        return torch.cat([named[k].grad.float().cpu().reshape(-1) for k in backbone])
    g_feat = grads(True, False)
    g_both = grads(True, True)
    g_log = grads(False, True)
    assert g_feat.norm().item() > 0
    # linearity of the backward pass in the output gradients
    cos = (g_both @ (g_feat + g_log) / (g_both.norm() * (g_feat + g_log).norm())).item()
    assert cos > 0.995 and g_both.norm().item() == pytest.approx((g_feat + g_log).norm().item(), rel=0.03)
    # oracle: autograd on the same loss
    sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    r1, r2, rf = omodel.forward(sdr, xs, True, ones, rt, {}, None, emulate_bf16=True)
    gref = torch.autograd.grad((rf * R).mean(), [sdr[k] for k in backbone])
    b = torch.cat([g.reshape(-1) for g in gref])
    cos = (g_feat @ b / (g_feat.norm() * b.norm())).item()
    assert cos > 0.97 and g_feat.norm().item() == pytest.approx(b.norm().item(), rel=0.05), (cos, g_feat.norm().item(), b.norm().item())


def test_resnet101_layerwise_forward_bound():
    """Every conv + BatchNorm(+ ReLU) unit of ResNet-101 (train mode, 2 BatchNorm groups like the SSL step): the raw
    conv output `c` and the normalised activation `y` the HIP path stored on its tape, recomputed by torch in fp32
    from the unit's OWN stored input -- one layer of bf16 rounding each, so the bound is tight where the whole-network
    comparisons (chaotic over 101 layers) cannot be: relative L2 < 4e-3 for c, < 1e-2 for y."""
    rt = 'resnet101'
    m = build(rt)
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=7), strict=True)
    m.train()
    gen = torch.Generator().manual_seed(23)
    xs = [torch.randn(2, 3, 128, 128, generator=gen).cuda(), (torch.randn(2, 3, 128, 128, generator=gen) * 1.5).cuda()]
    T = m.new_tape(groups=2)
    with torch.no_grad():
        m._forward_plan(xs, T)
    torch.cuda.synchronize()
    worst_c, worst_y, n = (0.0, ''), (0.0, ''), 0
    for p, inpl, planes, stride, dil, ds in m.blocks:
        units = [('.1', '.conv1', '.bn1', True), ('.2', '.conv2', '.bn2', True)]
        if ds:
            units.append(('.d', '.downsample.0', '.downsample.1', False))
        for tag, cname, bname, relu in units:
            conv, bn = m.convs[p + cname], m.bns[p + bname]
            x, c, y, mi, (N, H, W, Ho, Wo), nscale, rmask = T[p + tag]
            xt = x[:, :conv.ci].float().reshape(N, H, W, conv.ci).permute(0, 3, 1, 2).cpu()
            wt = conv.wb.float().reshape(conv.co, conv.k, conv.k, conv.ci).permute(0, 3, 1, 2).cpu()
            ref_c = F.conv2d(xt, wt, None, conv.stride, conv.pad, conv.dil)
            got_c = c.float().reshape(N, Ho, Wo, conv.co).permute(0, 3, 1, 2)
            ec = l2(got_c, ref_c)
            # BatchNorm per group (source / target halves of the batch) on the STORED conv output
            cs = got_c.cpu()
            ref_y = torch.cat([F.batch_norm(cs[g * N // 2:(g + 1) * N // 2], None, None, bn.gamma.cpu(), bn.beta.cpu(), True, 0.1, 1e-5)
                               for g in range(2)])
            if relu:
                ref_y = ref_y.relu()
            ey = l2(y.float().reshape(N, Ho, Wo, conv.co).permute(0, 3, 1, 2), ref_y)
            worst_c = max(worst_c, (ec, p + tag))
            worst_y = max(worst_y, (ey, p + tag))
            n += 1
            if rmask is not None and relu:
                assert torch.equal(unpack_bits(rmask, conv.co), (y.float() > 0).cpu()), p + tag
    assert n >= 66 + 4
    assert worst_c[0] < 4e-3, worst_c
    assert worst_y[0] < 1e-2, worst_y


def unpack_bits(mask, C):
    bits = (mask.cpu().unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1
    return bits.reshape(mask.shape[0], C).bool()
@pytest.mark.gpu
def test_weight_gradient_flush_per_layer_waits_for_the_small_map_batchnorm_backward():
    """A flush of the queued weight gradients must not fire between a PPM branch unit being QUEUED for the small-map
    BatchNorm backward (which writes its `dc`, the weight gradient's operand) and that launch: with one flush per layer
    (wgrad_group_gflop = 0), the default grouping and no weight-gradient stream, every gradient -- the branch convolutions'
    in particular -- equals the run with 500-GFLOP groups and the run through the general BatchNorm kernels."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=4)
    gen = torch.Generator().manual_seed(23)
    x = [torch.randn(2, 3, 64, 64, generator=gen).cuda(), torch.randn(2, 3, 64, 64, generator=gen).cuda()]
    g1, g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda(), torch.randn(4, 6, 4, 4, generator=gen).cuda()
    ones = torch.ones(2, 512)
    out = []
    for gflop, small in ((0.0, True), (500.0, True), (0.0, False)):
        m = build(rt)
        m.wgrad_group_gflop = gflop
        m.small_bn = small
        assert m.group_small_convs
        m.load_state_dict(sd, strict=True)
        m.train()
        m.set_drop_masks(ones, ones)
        m.flat_g.zero_()
        T = m.new_tape(groups=2)
        assert T.get('wgrad_stream') is None
        with torch.no_grad():
            m._forward_plan(x, T)
            m._backward_plan(T, g1, g2)
        torch.cuda.synchronize()
        out.append({k: v.clone() for k, v in m._gviews.items()})
    per_layer, grouped, general = out
    names = [k for k in per_layer if '.ppm.' in k and k.endswith('.1.weight')]
    assert len(names) == 8
    for k in per_layer:
        a, b = per_layer[k].float(), grouped[k].float()
        assert l2(a, b) < 1e-5, k                        # same operands, another split of the pixel sum at most
    for k in names:
        if '.ppm.0.' not in k:          # (the scale-1 branch's gradient is identically zero on this fixture: its BatchNorm gain is 0)
            assert float(per_layer[k].abs().sum()) > 0
        assert l2(per_layer[k].float(), general[k].float()) < 2e-2, k      # another BatchNorm summation order (bf16 dc)


@pytest.mark.gpu
def test_every_block_and_the_heads_against_the_oracles_own_tensors_full_size(capsys):
    """One tight whole-backward check (regda/_resnets.py:92-112, regda/models/Encoder.py:8-65,146-155): ResNet-101 at
    BASELINE config[0]'s geometry (2 + 2 images of 512 x 512, two BatchNorm groups).  The fp32 oracle runs the two forwards
    and ONE backward on the CPU and keeps, for every one of the 33 bottleneck blocks and for the heads, the unit's input, its
    output, the gradient that arrives at its output and the gradient it sends to its input.  Each HIP unit (`_block_fwd` /
    `_block_bwd`, `_heads_fwd` / `_heads_bwd`: the methods the plans are made of) then gets the ORACLE's input and the ORACLE's
    upstream gradient, rounded once to bf16 -- nothing a unit sees was produced by another HIP unit, so no depth-compounded
    noise enters -- and its output, data gradient, weight gradients and BatchNorm gradients are compared with

      (E) the oracle's OWN unit (oracle.model.bottleneck / heads: the functions oracle.model.forward is made of) run on the same
          tensors with bf16 rounding at the HIP path's storage points (`emulate_bf16='grad'`): relative L2 <= 2 % per
          quantity of a block, <= 3 % for the heads (measured: <= 0.8 % / 1.5 % for bn3's bias / 2.3 %).  The two round to the same neighbours almost everywhere; a wrong term, sign or scale is 100 %;
      (F) the fp32 oracle itself: <= 3 N + 0.5 %, N = |E - fp32| of that quantity computed here on the CPU.

    Why (F) cannot be "<= 1.5 %" for every quantity: the ReLU sign of a unit is taken from its STORED (bf16) convolution
    output; ~0.3 % of the elements lie so close to zero that the rounding flips them, and a flipped element carries its
    whole gradient as error -- sqrt(0.003) = 5.5 %.  conv3 / bn3 see one such layer (1 - 3 %), conv2 / bn2 two, conv1 / bn1
    three (6 - 10 %), in the emulating oracle and on the GPU alike (the test prints both columns)."""
    from oracle import labelpath as opath, labels as olab
    from regda_amd.synthetic import make_batch
    rt = 'resnet101'
    sd = omodel.init_state_dict(rt, 6, seed=5, res_gamma=0.02)
    b = make_batch(b=2, size=512, seed=77, device='cpu')
    ones = torch.ones(2, 512)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    names = omodel.param_names(sd)
    sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    taps = [{}, {}]
    s1, s2, _ = omodel.forward(sdr, b['images_s'], True, (ones, ones), rt, {}, taps[0])
    t1, t2, _ = omodel.forward(sdr, b['images_t'], True, (ones, ones), rt, {}, taps[1])
    hard = torch.from_numpy(olab.pseudo_selection(b['soft_t'].numpy(), 0.8, 0.6, -1))
    loss = opath.loss_calc([s1, s2], b['label_s'], -1) + opath.loss_calc([t1, t2], hard, -1)
    specs = omodel.layer_specs(rt)
    blocks = [sp[0] for sp in specs]
    keys = ['pool'] + blocks
    wrt = [sdr[k] for k in names] + [tp[k] for tp in taps for k in keys] + [s1, s2, t1, t2]
    gr = torch.autograd.grad(loss, wrt)
    gpar = dict(zip(names, gr[:len(names)]))
    gtap = [dict(zip(keys, gr[len(names) + d * len(keys):len(names) + (d + 1) * len(keys)])) for d in range(2)]
    glog = gr[-4:]
    RB = omodel._RoundBoth.apply
    sdq = {k: (omodel._rb(v) if (v.dim() == 4 and 'conv_last.4' not in k) else v) for k, v in sd.items()}

    def emu_unit(fn, prefix_ok, xs, gups):
        """The oracle's unit `fn` with bf16 rounding at the HIP storage points, per BatchNorm group (domain): returns
        (outputs per domain, dx per domain, parameter gradients summed over the domains)."""
        pn = [k for k in names if prefix_ok(k)]
        outs, dxs, gp = [], [], {k: 0.0 for k in pn}
        for x, gup in zip(xs, gups):
            w = dict(sdq)
            for k in pn:
                w[k] = sdq[k].clone().requires_grad_(True)
            x = x.detach().clone().requires_grad_(True)
            out = fn(RB(x), w)
            o_list = list(out) if isinstance(out, (tuple, list)) else [out]
            g = torch.autograd.grad(o_list, [x] + [w[k] for k in pn], list(gup) if isinstance(gup, (tuple, list)) else [gup],
                                    allow_unused=True)
            outs.append([o.detach() for o in o_list])
            dxs.append(g[0])
            for k, gk in zip(pn, g[1:]):
                if gk is not None:
                    gp[k] = gp[k] + gk
        return outs, dxs, gp

    def pxc(a, c):                       # two domains of NCHW fp32 -> one pixel-major bf16 matrix (source rows, then target rows)
        t = torch.cat([a, c]).detach()
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).to(torch.bfloat16).cuda().contiguous()

    def back(t, n, h, w):                # pixel-major -> NCHW fp32 on the host
        return t.float().reshape(n, h, w, -1).permute(0, 3, 1, 2).cpu()
    m = build(rt)
    m.load_state_dict(sd, strict=True)
    m.train()
    m.set_drop_masks(ones, ones)
    stream = torch.cuda.current_stream()
    N, rows = 4, []
    TIGHT, TIGHT_HEADS, FACTOR, FLOOR = 2e-2, 3e-2, 3.0, 5e-3     # measured: blocks <= 0.8 % (bn3.bias 1.5 %), heads <= 2.3 %

    def check(unit, what, got, emu, ref):
        got = got.float().cpu()
        rows.append((unit, what, l2(got, emu), l2(got, ref), l2(emu, ref)))
    with torch.no_grad():
        for bi, blk in enumerate(m.blocks):
            p, inpl, planes, stride, dil, ds = blk
            prev = keys[bi]
            h, w = taps[0][prev].shape[-2:]
            with torch.enable_grad():
                eo, edx, egp = emu_unit(lambda x, wts, sp=specs[bi]: omodel.bottleneck(x, wts, sp, True, {}, RB),
                                        lambda k, p=p: k.startswith(p + '.'), [taps[0][prev], taps[1][prev]],
                                        [omodel._rb(gtap[0][p]), omodel._rb(gtap[1][p])])
            T = m.new_tape(groups=2)
            y, h2, w2 = m._block_fwd(T, blk, pxc(taps[0][prev], taps[1][prev]), N, h, w, stream)
            check(p, 'out', back(y, N, h2, w2), torch.cat([eo[0][0], eo[1][0]]), torch.cat([taps[0][p], taps[1][p]]).detach())
            m.flat_g.zero_()
            m._begin_backward(T)
            gx = m._block_bwd(T, blk, pxc(gtap[0][p], gtap[1][p]), None)
            m._flush_wgrads(T)
            torch.cuda.synchronize()
            check(p, 'dx', back(gx, N, h, w), torch.cat(edx), torch.cat([gtap[0][prev], gtap[1][prev]]))
            for nm in egp:
                check(p, 'd ' + nm[len(p) + 1:], m._gviews[nm], egp[nm], gpar[nm])
        # ---- the heads (instance norm, pooling, PPM branches, the 3x3 convolution, classifier) as one unit
        last = blocks[-1]
        h, w = taps[0][last].shape[-2:]
        with torch.enable_grad():
            def hfn(x, wts):
                outs, feat = omodel.heads(x, wts, True, (ones, ones), {}, RB)
                return outs[0], outs[1]
            eo, edx, egp = emu_unit(hfn, lambda k: k.startswith('layer'), [taps[0][last], taps[1][last]],
                                    [(glog[0], glog[1]), (glog[2], glog[3])])
        T = m.new_tape(groups=2)
        x1, x2, feat = m._heads_fwd(T, pxc(taps[0][last], taps[1][last]), N, h, w, stream)
        check('heads', 'x1', x1, torch.cat([eo[0][0], eo[1][0]]), torch.cat([s1, t1]).detach())
        check('heads', 'x2', x2, torch.cat([eo[0][1], eo[1][1]]), torch.cat([s2, t2]).detach())
        ft = torch.cat([taps[0]['feat'], taps[1]['feat']]).detach()
        check('heads', 'feat', feat, ft, ft)
        m.flat_g.zero_()
        m._begin_backward(T)
        gy = m._heads_bwd(T, torch.cat([glog[0], glog[2]]).cuda().contiguous(), torch.cat([glog[1], glog[3]]).cuda().contiguous(),
                          None, stream)
        m._flush_wgrads(T)
        torch.cuda.synchronize()
        check('heads', 'dx', back(gy, N, h, w), torch.cat(edx), torch.cat([gtap[0][last], gtap[1][last]]))
        for nm in egp:
            if '.ppm.0.' not in nm:      # (the scale-1 branch is identically zero in the reference: noise)
                check('heads', 'd ' + nm, m._gviews[nm], egp[nm], gpar[nm])
    by = {}
    for unit, what, he, hf, ef in rows:
        by.setdefault(what if unit != 'heads' else 'heads: ' + what, []).append((he, hf, ef))
    with capsys.disabled():
        print('\n[per-unit forward / backward on the oracle\'s tensors, ResNet-101 2 + 2 x 512 x 512] %d quantities over %d units; '
              'relative L2, max over the units:' % (len(rows), len(m.blocks) + 1))
        print('   %-34s %10s %10s %12s' % ('quantity', 'HIP vs E', 'HIP vs F', 'N = E vs F'))
        for what, v in by.items():
            if not what.startswith('heads: d layer6'):
                print('   %-34s %10.4f %10.4f %12.4f' % (what, max(x[0] for x in v), max(x[1] for x in v), max(x[2] for x in v)))
    if os.environ.get('RGDA_UNIT_DUMP'):
        with open(os.environ['RGDA_UNIT_DUMP'], 'w') as f:
            for r in sorted(rows, key=lambda r: -r[2]):
                f.write('%-34s %-28s %.5f %.5f %.5f\n' % r)
    bad = [r for r in rows if not (r[2] <= (TIGHT_HEADS if r[0] == 'heads' else TIGHT) and r[3] <= FACTOR * r[4] + FLOOR)]
    assert not bad, bad[:8]
    assert len(rows) >= 33 * 11
