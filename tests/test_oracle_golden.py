"""CPU: the oracle (oracle/*.py) against the golden vectors minted from the
reference's own Python (tests/golden/make_goldens.py).  Integer paths must be
bit-exact; float paths carry an explicit tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import labels, labelpath, model, step

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_lrh_bit_exact(gold):
    g = gold('lrh.npz')
    n = int(g['n'])
    assert n >= 19
    for i in range(n):
        out = labels.homogenize(g[f'lab{i}'].astype(np.int64), g[f'reg{i}'].astype(np.int64),
                                float(g[f'pct{i}']), 6, -1)
        assert np.array_equal(out, g[f'out{i}'].astype(np.int64)), f'case {i}'


def test_lrh_tie_contract(gold):
    """exact 50/50 ties are homogenised (to the LOWEST tied id) only when n >= 256."""
    g = gold('lrh.npz')
    for i, n in zip(range(4, 16), [2, 4, 100, 128, 200, 254, 256, 258, 400, 512, 2000, 262144]):
        lab, out = g[f'lab{i}'].reshape(-1)[:n], g[f'out{i}'].reshape(-1)[:n]
        if n >= 256:
            assert (out == 2).all(), n
        else:
            assert np.array_equal(lab, out), n


def test_pseudo_selection_bit_exact(gold):
    g = gold('pseudo.npz')
    for i in range(int(g['n'])):
        out = labels.pseudo_selection(g[f'in{i}'], 0.8, 0.6, -1)
        assert np.array_equal(out, g[f'out{i}'].astype(np.int64)), f'case {i}'
    assert list(g['out3'].reshape(-1)) == [0, -1, -1, -1]


def test_pseudo_selection_asserts_range():
    with pytest.raises(AssertionError):
        labels.pseudo_selection(np.full((1, 2, 2, 2), 1.5, np.float32))


def test_downscale_label_bit_exact(gold):
    g = gold('downscale.npz')
    out = labels.downscale_label(g['lab'].astype(np.int64), 16, 6, -1, 0.75)
    assert np.array_equal(out, g['out'].astype(np.int64))
    assert out[0, 0, 0, 0] == 2 and out[0, 0, 0, 1] == -1 and out[0, 0, 0, 2] == -1 and out[0, 0, 0, 3] == -1


def test_label_refine_and_prototypes(gold):
    g = gold('refine.npz')
    t = lambda k: torch.from_numpy(g[k])
    k = g['feat_t'].shape[1]
    dist = labelpath.pearson_dist(t('feat_t').permute(0, 2, 3, 1).reshape(-1, k), t('protos'))
    # the feature that equals a prototype has dist ~1e-7 whose RELATIVE error is O(1) in fp32
    # for any summation order; everything else is tight.
    d_ref = g['dist']
    big = d_ref > 1e-4
    np.testing.assert_allclose(dist.numpy()[big], d_ref[big], rtol=1e-5, atol=1e-6)
    out = labelpath.label_refine(t('feat_t'), t('protos'), [t('p1'), t('p2')], t('soft'))
    np.testing.assert_allclose(out.numpy(), g['out'], rtol=2e-5, atol=1e-6)
    # the other modes the reference offers with label_t_sup=None, and the single-tensor prediction branch
    for key, preds, mode, temp in (('out_p', [t('p1'), t('p2')], 'p', 2.0), ('out_l', [t('p1'), t('p2')], 'l', 1.5),
                                   ('out_1', t('p1'), 'all', 2.0)):
        o = labelpath.label_refine(t('feat_t'), t('protos'), preds, t('soft'), True, mode, temp)
        np.testing.assert_allclose(o.numpy(), g[key], rtol=2e-5, atol=1e-6, err_msg=key)
    assert labelpath.label_refine(t('feat_t'), t('protos'), [t('p1'), t('p2')], t('soft'), True, 'n') is not None
    new, ds = labelpath.update_prototype(t('feat_s'), t('lab_s').long(), t('protos'))
    assert np.array_equal(ds.numpy(), g['ds'].astype(np.int64))
    np.testing.assert_allclose(new.numpy(), g['protos_new'], rtol=1e-5, atol=1e-6)


def test_label_refine_superpixel_view(gold):
    """The superpixel view (alignment.py:238-258), modes 'all' and 's', against outputs of the reference's own Aligner
    (make_goldens.gold_refine_sup; the inputs are refine.npz's)."""
    g, gs = gold('refine.npz'), gold('refine_sup.npz')
    t = lambda k: torch.from_numpy(g[k])
    sup = torch.from_numpy(gs['sup'].astype(np.int64)).reshape(2, 1, 64, 64)
    for key, mode, temp in (('out_all', 'all', 2.0), ('out_s', 's', 1.5)):
        o = labelpath.label_refine(t('feat_t'), t('protos'), [t('p1'), t('p2')], t('soft'), True, mode, temp, label_t_sup=sup)
        np.testing.assert_allclose(o.numpy(), gs[key], rtol=2e-5, atol=1e-6, err_msg=key)
    # modes 'p' / 'l' / 'n' do not look at the superpixels
    for mode in ('p', 'l', 'n'):
        a = labelpath.label_refine(t('feat_t'), t('protos'), [t('p1'), t('p2')], t('soft'), True, mode, 2.0, label_t_sup=sup)
        b = labelpath.label_refine(t('feat_t'), t('protos'), [t('p1'), t('p2')], t('soft'), True, mode, 2.0)
        assert torch.equal(a, b)


def test_loss_and_grad(gold):
    g = gold('loss.npz')
    p1 = torch.from_numpy(g['p1']).requires_grad_(True)
    p2 = torch.from_numpy(g['p2']).requires_grad_(True)
    lab = torch.from_numpy(g['lab'].astype(np.int64))
    loss = labelpath.loss_calc([p1, p2], lab, -1)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(p1.grad.numpy(), g['g1'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(p2.grad.numpy(), g['g2'], rtol=1e-5, atol=1e-8)
    # class-balanced variant
    cb = labelpath.ClassBalanceState(6, -1, 0.99, 2.0)
    q1 = torch.from_numpy(g['p1']).requires_grad_(True)
    q2 = torch.from_numpy(g['p2']).requires_grad_(True)
    lossb = labelpath.loss_calc([q1, q2], lab, -1, cb)
    lossb.backward()
    np.testing.assert_allclose(cb.freq.numpy(), g['freq'], rtol=1e-6)      # updated once per head
    np.testing.assert_allclose(lossb.item(), g['lossb'], rtol=1e-6)
    np.testing.assert_allclose(q1.grad.numpy(), g['gb1'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(q2.grad.numpy(), g['gb2'], rtol=1e-5, atol=1e-8)


def test_lr_and_ema(gold):
    g = gold('lr_ema.npz')
    for it, lr in zip(g['its'], g['lrs']):
        assert labelpath.lr_at(int(it)) == pytest.approx(float(lr), rel=1e-12, abs=0)
    sh = labelpath.ema_update(torch.from_numpy(g['w0']), torch.from_numpy(g['w1']), 0.99)
    np.testing.assert_allclose(sh.numpy(), g['shadow'], rtol=1e-6)
    # only parameters (not BN buffers) are shadowed (ema.py:41-44)
    assert list(g['shadow_keys']) == ['0.bias', '0.weight', '1.bias', '1.weight']


def test_state_dict_manifest():
    with open(os.path.join(GOLD, 'state_dict_manifest.json')) as f:
        man = json.load(f)
    sd = model.init_state_dict('resnet101', 6, seed=0)
    assert len(man) == 688
    assert [m[0] for m in man] == list(sd.keys())
    for (k, shape, dt), v in zip(man, sd.values()):
        assert list(v.shape) == shape and str(v.dtype).replace('torch.', '') == dt, k
    assert sum(v.numel() for k, v in sd.items() if k in model.param_names(sd)) == 88653900


@pytest.fixture(scope='module')
def small_case(gold):
    g = gold('model_small.npz')
    sd = model.init_state_dict('resnet101', 6, seed=1)
    return g, sd


def test_model_forward_small(small_case):
    g, sd = small_case
    ns = {}
    m5, m6 = torch.from_numpy(g['m5']), torch.from_numpy(g['m6'])
    s1, s2, fs = model.forward(sd, torch.from_numpy(g['xs']), True, (m5[0], m6[0]), 'resnet101', ns)
    np.testing.assert_allclose(s1.numpy(), g['s1'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(s2.numpy(), g['s2'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(fs.numpy()[:, :32], g['feat_s'], rtol=1e-3, atol=2e-4)


def test_full_step_small(small_case):
    """End-to-end: oracle.step.CpuStep vs the reference's step composed as
    tools/train_ssl_reg.py:198-241 (losses, labels, prototypes, grads, BN buffers, teacher)."""
    g, sd = small_case
    t = lambda k: torch.from_numpy(g[k])
    st = step.CpuStep(sd, t('protos'), lr=0.0)
    m5, m6 = t('m5'), t('m6')
    r = st.step(t('xs'), t('lab_s').long(), t('xt'), t('soft_t'), t('regs').long(),
                (m5[0], m6[0]), (m5[1], m6[1]))
    np.testing.assert_allclose(r['preds'][2].numpy(), g['t1'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(r['soft'].numpy(), g['soft2'], rtol=1e-3, atol=1e-5)
    # labels: the float inputs differ by summation-order noise, so allow a handful of flips
    assert (r['hard'].numpy() != g['hard2'].astype(np.int64)).mean() < 2e-3
    assert r['loss_source'] == pytest.approx(float(g['loss_s']), rel=1e-4)
    assert r['loss_target'] == pytest.approx(float(g['loss_t']), rel=2e-3)
    assert r['grad_norm'] == pytest.approx(float(g['grad_norm']), rel=5e-3)
    np.testing.assert_allclose(st.prototypes.numpy(), g['protos_new'], rtol=1e-4, atol=1e-6)
    for k in ['encoder.resnet.conv1.weight', 'layer5.conv_last.4.weight', 'layer5.conv_last.4.bias',
              'encoder.resnet.bn1.bias', 'layer6.ppm.3.2.bias']:
        ref = g['grad:' + k]
        got = r['grads'][k].numpy()
        assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7, k
    np.testing.assert_allclose(st.sd['encoder.resnet.bn1.running_mean'].numpy(), g['bn1_rm'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st.sd['layer5.conv_last.1.running_var'].numpy(), g['l5bn_rv'], rtol=1e-4, atol=1e-6)
    assert int(st.sd['encoder.resnet.bn1.num_batches_tracked']) == int(g['nbt']) == 2
    with torch.no_grad():
        sd_eval = {k: v.detach() for k, v in st.sd.items()}
        probs = model.forward(sd_eval, t('xt'), False)
    np.testing.assert_allclose(probs.numpy(), g['probs'], rtol=2e-3, atol=2e-5)


def _slice_of(name, tensors):
    """'a.b.weight[:8]' -> tensors['a.b.weight'][:8]."""
    base, _, sl = name.partition('[')
    t = tensors[base]
    return t[:int(sl[1:-1])] if sl else t


def test_full_step_mid_size(gold):
    """oracle.step.CpuStep against the reference's step on the better-conditioned ResNet-101 fixture (model_mid.npz: minted
    by make_goldens.gold_model128 at 2 x 3 x 128 x 128, residual gain 0.02): losses, refined soft labels, pseudo labels,
    gradient norm, prototypes and all 17 stored gradient tensors over the depth of the network."""
    g = gold('model_mid.npz')
    sd = model.init_state_dict('resnet101', 6, seed=3, res_gamma=0.02)
    t = lambda k: torch.from_numpy(g[k])
    st = step.CpuStep(sd, t('protos'), lr=0.0)
    m5, m6 = t('m5'), t('m6')
    r = st.step(t('xs'), t('lab_s').long(), t('xt'), t('soft_t'), t('regs').long(), (m5[0], m6[0]), (m5[1], m6[1]))
    np.testing.assert_allclose(r['preds'][2].numpy(), g['t1'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(r['soft'].numpy(), g['soft2'].astype(np.float32), rtol=2e-3, atol=1e-3)     # stored as fp16
    assert (r['hard'].numpy() != g['hard2'].astype(np.int64)).mean() < 2e-3
    assert r['loss_source'] == pytest.approx(float(g['loss_s']), rel=1e-4)
    assert r['loss_target'] == pytest.approx(float(g['loss_t']), rel=2e-3)
    assert r['grad_norm'] == pytest.approx(float(g['grad_norm']), rel=5e-3)
    np.testing.assert_allclose(st.prototypes.numpy(), g['protos_new'], rtol=1e-4, atol=1e-6)
    keys = [k[5:] for k in g.files if k.startswith('grad:')]
    assert len(keys) >= 12
    for k in keys:
        ref = g['grad:' + k]
        got = _slice_of(k, r['grads']).numpy()
        assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max() + 1e-7, k
    np.testing.assert_allclose(st.sd['encoder.resnet.bn1.running_mean'].numpy(), g['bn1_rm'], rtol=1e-4, atol=1e-6)


def test_teacher_harness_oracle_vs_reference(gold):
    """oracle/teacher.py (pre_slide, tta_predict, soft-label resize) against the reference's own functions
    (tests/golden/tta.npz; ttach restated, see the oracle's header)."""
    import torch.nn.functional as F
    from oracle import teacher
    g = gold('tta.npz')
    wgt, bias = torch.from_numpy(g['wgt']), torch.from_numpy(g['bias'])

    def model(x):
        return torch.softmax(F.conv2d(x, wgt, bias, padding=1), dim=1)
    for i in range(3):
        img, tile = torch.from_numpy(g[f'img{i}']), tuple(int(v) for v in g[f'tile{i}'])
        for tta in (0, 1):
            got = teacher.pre_slide(model, img, num_classes=5, tile_size=tile, tta=bool(tta))
            np.testing.assert_allclose(got.numpy(), g[f'probs{i}_tta{tta}'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(teacher.tta_predict(model, torch.from_numpy(g['img0'])).numpy(), g['tta_single'],
                               rtol=0, atol=1e-6)
    np.testing.assert_allclose(teacher.soft_label(torch.from_numpy(g['probs1_tta1']), (64, 48)).numpy(), g['resized'],
                               rtol=0, atol=1e-6)
    assert teacher.windows(40, 24, (16, 16))[-1] == (24, 40, 8, 24)      # last window re-aligned to the border


def test_eval_path_oracle_definitions():
    """oracle/evalpath.py on a hand-checked 3-class confusion matrix (no reference vector exists for this row: the
    reference's metric class sits on the un-vendored `ever` package -- see the oracle's header)."""
    from oracle import evalpath
    y_true = np.array([0, 0, 1, 1, 1, 2, -1, 2])
    y_pred = np.array([0, 1, 1, 1, 0, 2, 2, 1])
    cm = evalpath.confusion_matrix(y_true, y_pred, 3)
    assert cm.tolist() == [[1, 1, 0], [1, 2, 0], [0, 1, 1]]
    s = evalpath.summary(cm, ignore_labels=[0])
    assert s['iou'] == [0.4, 0.5] and s['miou'] == 0.45                 # class 0 dropped: (2/5 + 1/2) / 2
    assert s['recall'] == [round(2 / 3, 5), 0.5] and s['precision'] == [0.5, 1.0]


def test_prototype_contrastive_loss_oracle_vs_reference(gold):
    """oracle.labelpath.prototype_contrastive_loss against regda/loss.py's PrototypeContrastiveLoss (loss and the
    gradient w.r.t. the features), tests/golden/pcl.npz."""
    g = gold('pcl.npz')
    for i in range(3):
        feat = torch.from_numpy(g[f'feat{i}']).requires_grad_(True)
        loss = labelpath.prototype_contrastive_loss(torch.from_numpy(g[f'protos{i}']), feat, torch.from_numpy(g[f'lab{i}']),
                                                    temperature=float(g[f'temp{i}']), ignore_label=-1)
        loss.backward()
        np.testing.assert_allclose(loss.detach().numpy(), g[f'loss{i}'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(feat.grad.numpy(), g[f'gfeat{i}'], rtol=1e-5, atol=1e-8)


def test_full_stage2_step_small(small_case, gold):
    """End-to-end: oracle.step.CpuAlignStep vs the reference's stage-2 iteration composed as
    tools/train_align_reg.py:144-196 (tests/golden/align_small.npz; same inputs as the SSL fixture)."""
    g, sd = small_case
    a = gold('align_small.npz')
    t = lambda k: torch.from_numpy(g[k])        # noqa: E731
    st = step.CpuAlignStep(sd, t('protos'), lr=0.0, proto_decay=0.999)
    m5, m6 = torch.from_numpy(a['m5']), torch.from_numpy(a['m6'])
    r = st.step(t('xs'), t('lab_s').long(), t('xt'), t('regs').long(), (m5[0], m6[0]), (m5[1], m6[1]))
    assert np.array_equal(r['label_s_down'].numpy(), a['label_s_down'].astype(np.int64))
    assert (r['hard'].numpy() != a['hard'].astype(np.int64)).mean() < 2e-3
    assert (r['label_t'].numpy() != a['label_t'].astype(np.int64)).mean() < 0.07      # 32 cells: at most two flips
    np.testing.assert_allclose(st.prototypes.numpy(), a['protos_new'], rtol=1e-4, atol=1e-6)
    assert r['loss_seg'] == pytest.approx(float(a['loss_seg']), rel=1e-4)
    assert r['loss_align'] == pytest.approx(float(a['loss_align']), rel=1e-3)
    assert r['grad_norm'] == pytest.approx(float(a['grad_norm']), rel=5e-3)
    for k in ['encoder.resnet.conv1.weight', 'layer5.conv_last.4.weight', 'encoder.resnet.bn1.bias',
              'encoder.resnet.layer4.2.bn3.bias', 'layer6.ppm.3.2.bias']:
        ref, got = a['grad:' + k], r['grads'][k].numpy()
        assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7, k


# ---------------------------------------------------------------- ASPP heads
def test_aspp_head_matches_reference_module(gold):
    """oracle.model.aspp_head vs the reference's Classifier_Module (Encoder.py:68-84): output and every gradient."""
    g = gold('aspp.npz')
    x = torch.from_numpy(g['cm_x']).requires_grad_(True)
    ws = [torch.from_numpy(g[f'cm_w{i}']).requires_grad_(True) for i in range(4)]
    bs = [torch.from_numpy(g[f'cm_b{i}']).requires_grad_(True) for i in range(4)]
    y = model.aspp_head(x, ws, bs)
    np.testing.assert_allclose(y.detach().numpy(), g['cm_y'], rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(g['cm_gy'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['cm_gx'], rtol=1e-4, atol=1e-6)
    for i in range(4):
        np.testing.assert_allclose(ws[i].grad.numpy(), g[f'cm_gw{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bs[i].grad.numpy(), g[f'cm_gb{i}'], rtol=1e-4, atol=1e-5)


def test_aspp_model_matches_reference(gold):
    """The use_ppm=False network: state_dict layout, train outputs, loss, gradients and eval probabilities against
    the reference Deeplabv2 (tests/golden/make_goldens.py gold_aspp)."""
    g = gold('aspp.npz')
    sd = model.init_state_dict('resnet101', 6, seed=2, head='aspp')
    assert list(g['keys']) == list(sd.keys())
    names = model.param_names(sd)
    sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    ns = {}
    xs, lab = torch.from_numpy(g['xs']), torch.from_numpy(g['lab']).long()
    x1, x2, feat = model.forward(sdr, xs, True, None, 'resnet101', ns)
    np.testing.assert_allclose(x1.detach().numpy(), g['x1'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(x2.detach().numpy(), g['x2'], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(feat.detach().numpy()[:, :32], g['feat'], rtol=1e-3, atol=2e-4)
    loss = labelpath.loss_calc([x1, x2], lab, -1)
    assert loss.item() == pytest.approx(float(g['loss']), rel=1e-4)
    grads = dict(zip(names, torch.autograd.grad(loss, [sdr[k] for k in names])))
    for k in ('layer5.conv2d_list.0.bias', 'layer6.conv2d_list.3.bias', 'encoder.resnet.bn1.weight'):
        ref = g['grad:' + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7, k
    ref = g['grad:layer5.conv2d_list.1.weight[:, :32]']
    assert np.abs(grads['layer5.conv2d_list.1.weight'][:, :32].numpy() - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7
    sd_eval = {k: ns.get(k, v).detach() for k, v in sd.items()}
    with torch.no_grad():
        probs = model.forward(sd_eval, xs, False, None, 'resnet101')
    np.testing.assert_allclose(probs.numpy(), g['probs'], rtol=1e-3, atol=1e-4)


def test_eval_metrics_against_scikit_learn_definitions():
    """`ever`'s PixelMetric (the base class of PixelMetricIgnore, regda/gast/metrics.py:19-45) is un-vendored and not
    installed, so no vector of the reference itself exists for the evaluation row.  What CAN be pinned is that the
    restated confusion matrix and per-class IoU / precision / recall / F1 are the standard definitions with rows =
    truth: an independent implementation (scikit-learn) on random labels, ignored pixels (label -1) excluded like
    eval.py:45-48, including a class that never occurs."""
    from sklearn import metrics as skm
    from oracle import evalpath
    rng = np.random.default_rng(11)
    C = 6
    y_true = rng.integers(-1, C - 1, size=20000)             # class C-1 never occurs in the truth
    y_pred = np.where(rng.random(20000) < 0.7, np.maximum(y_true, 0), rng.integers(0, C - 1, size=20000))
    cm = evalpath.confusion_matrix(y_true, y_pred, C)
    keep = y_true >= 0
    ref_cm = skm.confusion_matrix(y_true[keep], y_pred[keep], labels=list(range(C)))
    assert np.array_equal(cm, ref_cm)
    iou, f1, prec, rec = evalpath.per_class(cm)
    labels = list(range(C - 1))
    np.testing.assert_allclose(iou[:-1], skm.jaccard_score(y_true[keep], y_pred[keep], labels=labels, average=None), rtol=1e-12)
    np.testing.assert_allclose(prec[:-1], skm.precision_score(y_true[keep], y_pred[keep], labels=labels, average=None), rtol=1e-12)
    np.testing.assert_allclose(rec[:-1], skm.recall_score(y_true[keep], y_pred[keep], labels=labels, average=None), rtol=1e-12)
    np.testing.assert_allclose(f1[:-1], skm.f1_score(y_true[keep], y_pred[keep], labels=labels, average=None), rtol=1e-12)
    assert np.isnan(iou[-1])                                  # 0 / 0 for the absent class, like numpy does in `ever`
    s = evalpath.summary(cm[:-1, :-1], ignore_labels=[0])
    assert s['miou'] == np.round(np.mean(np.round(iou[1:-1], 5)), 5)


def test_sam_region_map_assembly(gold):
    """oracle/regions.py against the reference's own loop (SAM.get_local_regions, local_region_homog.py:51-56) run on
    synthetic annotations: ids = generator index + 1, area threshold inclusive, later masks over earlier ones."""
    from oracle import regions as oreg
    g = gold('regions.npz')
    for i in range(int(g['n'])):
        out = oreg.regions_from_masks(g[f'masks{i}'], g[f'areas{i}'], int(g[f'thr{i}']))
        assert out.dtype == np.int32 and np.array_equal(out, g[f'regions{i}']), i
    k = g['masks0'].shape[0]
    assert (g['regions0'] != 1).all() and (g['regions0'] == k // 2 + 1).any()      # below / exactly at the threshold


def test_bf16_tolerance_table_is_what_the_rounding_model_gives():
    """tests/golden/bf16_tolerances.json (the rounding-noise units N the GPU suite states its step- and model-level
    tolerances in) is the output of tests/golden/derive_tolerances.py: recompute the shallow-topology fixtures here on
    the CPU.  N is itself the outcome of a chaotic amplification (another thread count re-associates the fp32 sums), so
    the check is agreement within a factor 1.5, key by key."""
    import importlib.util
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('derive_tolerances', os.path.join(here, 'derive_tolerances.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    table = json.load(open(os.path.join(here, 'bf16_tolerances.json')))
    assert table['factor'] == 3.0
    for name, fn in (('shallow_model', mod.shallow_model_fixture), ('shallow_step', lambda: mod.shallow_fixture(False))):
        got = fn()
        for k, v in table[name].items():
            if isinstance(v, dict):
                continue
            w = got[k]
            if 'cos' in k:
                v, w = 1.0 - v, 1.0 - w
            assert w <= 1.5 * v + 1e-6 and v <= 1.5 * w + 1e-6, (name, k, v, w)


def test_eval_branch_on_warm_statistics_oracle_vs_reference():
    """oracle.model.forward in eval mode against the reference's own probabilities on WARM BatchNorm statistics
    (tests/golden/eval_warm.npz, make_goldens.gold_eval_warm: the reference ResNet-101 after 20 train-mode forwards, all of
    its BatchNorm buffers in the fixture): the pin behind tests/test_model_gpu.py::
    test_eval_branch_on_warm_statistics_vs_reference_golden."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('derive_tolerances', os.path.join(here, 'derive_tolerances.py'))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    sd, xe, ref = D.eval_warm_inputs()
    assert int(sd['encoder.resnet.bn1.num_batches_tracked']) == 20
    with torch.no_grad():
        got = model.forward(sd, xe, False, None, 'resnet101')
    assert got.shape == ref.shape == (1, 6, 128, 128)
    assert float((got - ref).abs().max()) < 1e-5
    # a soft, well-conditioned output (the point of the fixture): no class above 0.9 anywhere, several classes in use
    assert float(ref.max()) < 0.9 and len(np.unique(ref.argmax(1).numpy())) >= 4
