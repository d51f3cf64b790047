"""Mint golden vectors from the reference's OWN Python (imported behind stubs).

Run in the build container only (needs /root/reference):
    python tests/golden/make_goldens.py
Writes small .npz / .json fixtures next to this file.  The fixtures are data
(inputs + the reference's outputs); no reference source is copied.  The
reference ships no tests or known-answer vectors for this path, so
these are what pins the oracle (oracle/*.py) and, through it, the HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refstubs  # noqa: E402

_refstubs.install()

from regda.utils.local_region_homog import Homogenizer  # noqa: E402
from regda.gast.pseudo_generation import pseudo_selection  # noqa: E402
from regda.gast.alignment import Aligner, DownscaleLabel  # noqa: E402
from regda.gast.balance import CrossEntropy, ClassBalance  # noqa: E402
from regda.utils.tools import loss_calc, adjust_learning_rate  # noqa: E402
from regda.utils.ema import ExponentialMovingAverage  # noqa: E402
from regda.models.Encoder import Deeplabv2  # noqa: E402

from oracle import model as omodel  # noqa: E402  (only for the seeded weight generator)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in arrs.items()})


class _Log:
    def info(self, *a, **k):
        pass


def random_regions(rng, b, h, w, nreg, zero_frac=0.2):
    """Random rectangles painted over a zero background, later over earlier
    (mirrors local_region_homog.py:51-56)."""
    regs = np.zeros((b, h, w), np.int64)
    for i in range(b):
        for r in range(1, nreg + 1):
            y0, x0 = rng.integers(0, h), rng.integers(0, w)
            hh, ww = rng.integers(1, max(2, h // 2)), rng.integers(1, max(2, w // 2))
            if rng.random() > zero_frac:
                regs[i, y0:y0 + hh, x0:x0 + ww] = r
    return regs


def gold_lrh():
    rng = np.random.default_rng(2333)
    out = {}
    cases = []
    # random cases
    for ci, (b, h, w, nreg, pct) in enumerate([(2, 64, 64, 40, 0.5), (3, 48, 80, 25, 0.9),
                                                (1, 33, 17, 7, 0.5), (2, 64, 64, 300, 0.5)]):
        lab = rng.integers(-1, 6, size=(b, h, w)).astype(np.int64)
        # make regions label-correlated so that some regions do homogenise
        regs = random_regions(rng, b, h, w, nreg)
        for r in range(1, nreg + 1, 2):
            lab[(regs == r) & (rng.random((b, h, w)) < 0.7)] = r % 6
        cases.append((lab, regs, pct))
    # tie cases: one region of n pixels, exactly half class 4 and half class 2
    for n in [2, 4, 100, 128, 200, 254, 256, 258, 400, 512, 2000, 262144]:
        side = int(np.ceil(np.sqrt(n)))
        lab = np.full((1, side, side + 1), 3, np.int64)
        regs = np.zeros((1, side, side + 1), np.int64)
        flat_l, flat_r = lab.reshape(-1), regs.reshape(-1)
        flat_r[:n] = 1
        flat_l[:n // 2] = 4
        flat_l[n // 2:n] = 2
        cases.append((lab, regs, 0.5))
    # all-ignored region, region 0 only, sparse ids (max id >> count)
    lab = rng.integers(0, 6, size=(1, 16, 16)).astype(np.int64)
    regs = np.ones((1, 16, 16), np.int64)
    lab[:] = -1
    cases.append((lab.copy(), regs.copy(), 0.5))
    lab = rng.integers(-1, 6, size=(2, 16, 16)).astype(np.int64)
    cases.append((lab.copy(), np.zeros((2, 16, 16), np.int64), 0.5))
    regs = np.zeros((1, 16, 16), np.int64)
    regs[0, :8] = 1000
    regs[0, 8:, :8] = 7
    lab = np.where(rng.random((1, 16, 16)) < 0.8, 1, 5).astype(np.int64)
    cases.append((lab, regs, 0.5))
    for i, (lab, regs, pct) in enumerate(cases):
        h = Homogenizer(percent=pct, class_num=6, ignore_label=-1)
        res = h(torch.from_numpy(lab.copy()), torch.from_numpy(regs.copy())).numpy()
        out[f'lab{i}'] = lab.astype(np.int8)
        out[f'reg{i}'] = regs.astype(np.int32)
        out[f'pct{i}'] = np.float64(pct)
        out[f'out{i}'] = res.astype(np.int8)
    out['n'] = np.int64(len(cases))
    save('lrh.npz', **out)


def gold_pseudo():
    rng = np.random.default_rng(7)
    out = {}
    cases = []
    for (b, h, w, sharp) in [(2, 32, 32, 3.0), (1, 17, 9, 8.0), (2, 16, 16, 0.5)]:
        logits = rng.normal(size=(b, 6, h, w)).astype(np.float32) * sharp
        p = torch.softmax(torch.from_numpy(logits), 1).numpy()
        cases.append(p)
    # the toy from SURVEY Appendix B + exact-threshold equality + class never above 0.6
    toy = np.zeros((1, 4, 1, 4), np.float32)
    toy[0, :, 0, 0] = [0.9, 0.05, 0.03, 0.02]
    toy[0, :, 0, 1] = [0.7, 0.2, 0.05, 0.05]
    toy[0, :, 0, 2] = [0.5, 0.3, 0.1, 0.1]
    toy[0, :, 0, 3] = [0.34, 0.33, 0.2, 0.13]
    cases.append(toy)
    eq = np.zeros((1, 3, 1, 4), np.float32)
    eq[0, :, 0, 0] = [1.0, 0.0, 0.0]
    eq[0, :, 0, 1] = [np.float32(1.0) * np.float32(0.8), 0.1, 0.1]   # exactly at the threshold -> rejected
    eq[0, :, 0, 2] = [np.nextafter(np.float32(0.8), np.float32(1)), 0.1, 0.05]
    eq[0, :, 0, 3] = [0.3, 0.61, 0.09]                                # class 1 max .61 -> t=.6 -> .61 passes
    cases.append(eq)
    two = np.zeros((1, 2, 1, 2), np.float32)                          # two classes pass -> ambiguous
    two[0, :, 0, 0] = [0.7, 0.7]
    two[0, :, 0, 1] = [0.65, 0.1]
    cases.append(two)
    for i, p in enumerate(cases):
        res = pseudo_selection(torch.from_numpy(p.copy()), 0.8, 0.6, 'tensor', -1).numpy()
        out[f'in{i}'] = p
        out[f'out{i}'] = res.astype(np.int8)
    out['n'] = np.int64(len(cases))
    save('pseudo.npz', **out)


def gold_downscale():
    rng = np.random.default_rng(11)
    lab = rng.integers(-1, 6, size=(2, 64, 96)).astype(np.int64)
    # block-constant regions so some blocks pass 0.75
    blk = rng.integers(-1, 6, size=(2, 4, 6)).astype(np.int64)
    big = np.kron(blk, np.ones((16, 16), np.int64))
    lab = np.where(rng.random(lab.shape) < 0.8, big, lab)
    # exact cases in image 0, block (0,0): exactly 192/256 class 2 (kept); block (0,1): 191/256 (-> -1)
    b00 = np.full(256, 5, np.int64); b00[:192] = 2
    lab[0, :16, :16] = b00.reshape(16, 16)
    b01 = np.full(256, 5, np.int64); b01[:191] = 2
    lab[0, :16, 16:32] = b01.reshape(16, 16)
    b02 = np.full(256, 1, np.int64); b02[:128] = 3      # 50/50 tie -> -1
    lab[0, :16, 32:48] = b02.reshape(16, 16)
    b03 = np.full(256, -1, np.int64); b03[:10] = 3      # mostly ignored -> -1
    lab[0, :16, 48:64] = b03.reshape(16, 16)
    ds = DownscaleLabel(scale_factor=16, n_classes=6, ignore_label=-1, min_ratio=0.75)
    res = ds(torch.from_numpy(lab.copy())).numpy()
    save('downscale.npz', lab=lab.astype(np.int8), out=res.astype(np.int8))


def gold_refine():
    torch.manual_seed(5)
    b, k, h, w, C = 2, 64, 4, 4, 6
    al = Aligner(logger=_Log(), feat_channels=k, class_num=C, ignore_label=-1, decay=0.996, resume=None)
    protos = torch.randn(C, k)
    al.prototypes = protos.clone()
    feat_t = torch.randn(b, k, h, w)
    feat_t[0, :, 0, 0] = protos[2]          # 1/pearson ~ 1e7 case
    p1, p2 = torch.randn(b, C, h, w) * 2, torch.randn(b, C, h, w) * 2
    soft = torch.softmax(torch.randn(b, C, 64, 64) * 3, 1)
    out = al.label_refine(None, feat_t, [p1, p2], soft, refine=True, mode='all', temp=2.0)
    dist = al._pearson_dist(feat_t.permute(0, 2, 3, 1).reshape(-1, k), protos)
    # the other refine modes with label_t_sup=None, and the single-tensor prediction branch (no random numbers drawn)
    out_p = al.label_refine(None, feat_t, [p1, p2], soft, refine=True, mode='p', temp=2.0)
    out_l = al.label_refine(None, feat_t, [p1, p2], soft, refine=True, mode='l', temp=1.5)
    out_1 = al.label_refine(None, feat_t, p1, soft, refine=True, mode='all', temp=2.0)
    out_n = al.label_refine(None, feat_t, [p1, p2], soft, refine=True, mode='n', temp=2.0)
    assert out_n is soft
    feat_s = torch.randn(b, k, h, w)
    lab_s = torch.from_numpy(np.kron(np.random.default_rng(3).integers(-1, 6, size=(b, 4, 4)),
                                     np.ones((16, 16), np.int64)))
    lab_s[0, :16, :16] = torch.from_numpy(np.random.default_rng(4).integers(-1, 6, size=(16, 16)))
    ds = al.update_prototype(feat_s, lab_s)
    # class absent from the batch keeps its old prototype
    save('refine.npz', feat_t=feat_t.numpy(), protos=protos.numpy(), p1=p1.numpy(), p2=p2.numpy(),
         soft=soft.numpy(), out=out.numpy(), out_p=out_p.numpy(), out_l=out_l.numpy(), out_1=out_1.numpy(),
         dist=dist.numpy(), feat_s=feat_s.numpy(),
         lab_s=lab_s.numpy().astype(np.int8), ds=ds.numpy().astype(np.int8),
         protos_new=al.prototypes.numpy())


def gold_refine_sup():
    """label_refine WITH the superpixel view (alignment.py:238-258), modes 'all' and 's', from the reference's own Aligner
    (torch_scatter stubbed, _refstubs.py).  Same inputs as gold_refine (refine.npz holds them) plus a superpixel map: painted rectangles over a
    zero background; image 0 does not hold the batch's largest id, image 1 does (its pixels are `ignored`)."""
    torch.manual_seed(5)
    b, k, h, w, C = 2, 64, 4, 4, 6
    al = Aligner(logger=_Log(), feat_channels=k, class_num=C, ignore_label=-1, decay=0.996, resume=None)
    protos = torch.randn(C, k)
    al.prototypes = protos.clone()
    feat_t = torch.randn(b, k, h, w)
    feat_t[0, :, 0, 0] = protos[2]
    p1, p2 = torch.randn(b, C, h, w) * 2, torch.randn(b, C, h, w) * 2
    soft = torch.softmax(torch.randn(b, C, 64, 64) * 3, 1)
    rng = np.random.default_rng(17)
    sup = random_regions(rng, b, 64, 64, 23, zero_frac=0.1)
    sup[0][sup[0] == sup.max()] = 1
    sup[1, 40:50, 8:30] = 31                     # the largest id of the batch: ignored
    sup[0, :2, :] = np.arange(64)[None, :] % 29   # single-pixel superpixels / ids changing inside a wave
    sup_t = torch.from_numpy(sup).reshape(b, 1, 64, 64)
    out_all = al.label_refine(sup_t, feat_t, [p1, p2], soft, refine=True, mode='all', temp=2.0)
    out_s = al.label_refine(sup_t, feat_t, [p1, p2], soft, refine=True, mode='s', temp=1.5)
    out_p = al.label_refine(sup_t, feat_t, [p1, p2], soft, refine=True, mode='p', temp=2.0)      # does not look at sup
    assert torch.equal(out_p, al.label_refine(None, feat_t, [p1, p2], soft, refine=True, mode='p', temp=2.0))
    base = np.load(os.path.join(HERE, 'refine.npz'))             # same seed, same draws: the inputs are refine.npz's
    for key, val in (('feat_t', feat_t), ('protos', protos), ('p1', p1), ('p2', p2), ('soft', soft)):
        assert np.array_equal(base[key], val.numpy()), key
    save('refine_sup.npz', sup=sup.astype(np.int8), out_all=out_all.numpy(), out_s=out_s.numpy())


def gold_regions():
    """SAM.get_local_regions (regda/utils/local_region_homog.py:41-64) -- the reference's OWN assembly of the region map
    from the mask generator's output: masks in generator order, those with area >= area_thrshold painted one over the
    other as ids i + 1.  The third-party generator (segment_anything) and the image reader are replaced by synthetic
    annotations; the loop that runs is the reference's."""
    import cv2
    from regda.utils.local_region_homog import SAM
    rng = np.random.default_rng(77)
    out = {}
    for ci, (h, w, k, thr) in enumerate([(64, 64, 12, 100), (48, 80, 40, 64), (96, 96, 5, 1024), (64, 64, 0, 100)]):
        anns = []
        for i in range(k):
            m = np.zeros((h, w), bool)
            y0, x0 = rng.integers(0, h - 2), rng.integers(0, w - 2)
            hh, ww = rng.integers(2, h // 2), rng.integers(2, w // 2)
            m[y0:y0 + hh, x0:x0 + ww] = True
            m &= rng.random((h, w)) < 0.9                       # ragged masks
            anns.append({'segmentation': m, 'area': int(m.sum())})
        if k:
            anns[k // 2]['area'] = thr                          # exactly at the threshold: kept (>=)
            anns[0]['area'] = thr - 1                           # just below: dropped

        class _Gen:
            def generate(self, image, _a=anns):
                return _a
        sam = SAM.__new__(SAM)
        sam.model = _Gen()
        cv2.imread = lambda p, _s=(h, w): np.zeros(_s + (3,), np.uint8)
        cv2.cvtColor = lambda im, code: im
        cv2.COLOR_BGR2RGB = 4
        reg = sam.get_local_regions(image_path='unused.png', area_thrshold=thr, save=False, show=False)
        assert reg.dtype == np.int32 and reg.shape == (h, w)
        out[f'masks{ci}'] = np.stack([a['segmentation'] for a in anns]).astype(np.uint8) if k else np.zeros((0, h, w), np.uint8)
        out[f'areas{ci}'] = np.array([a['area'] for a in anns], np.int64)
        out[f'thr{ci}'] = np.int64(thr)
        out[f'regions{ci}'] = reg
    out['n'] = np.int64(4)
    save('regions.npz', **out)


def gold_loss():
    torch.manual_seed(9)
    b, C = 2, 6
    p1 = (torch.randn(b, C, 4, 4) * 2).requires_grad_(True)
    p2 = (torch.randn(b, C, 4, 4) * 2).requires_grad_(True)
    lab = torch.randint(-1, C, (b, 64, 64))
    lab[0, :32] = -1                                   # many ignored pixels: checks the mean-over-all denominator
    ce = CrossEntropy(ignore_label=-1, class_balancer=None)
    loss = loss_calc([p1, p2], lab, loss_fn=ce, multi=True)
    loss.backward()
    # class-balanced variant (--bcs 1): ClassBalance decay .99, temperature 2.0
    cb = ClassBalance(class_num=C, ignore_label=-1, decay=0.99, temperature=2.0)
    ceb = CrossEntropy(ignore_label=-1, class_balancer=cb)
    q1, q2 = p1.detach().clone().requires_grad_(True), p2.detach().clone().requires_grad_(True)
    lossb = loss_calc([q1, q2], lab, loss_fn=ceb, multi=True)
    lossb.backward()
    save('loss.npz', p1=p1.detach().numpy(), p2=p2.detach().numpy(), lab=lab.numpy().astype(np.int8),
         loss=loss.detach().numpy(), g1=p1.grad.numpy(), g2=p2.grad.numpy(),
         lossb=lossb.detach().numpy(), gb1=q1.grad.numpy(), gb2=q2.grad.numpy(),
         freq=cb.freq.numpy())


def gold_lr_ema():
    class Cfg:
        LEARNING_RATE = 1e-2
        POWER = 0.9
        NUM_STEPS = 6000 * 1.5
        PREHEAT_STEPS = int(6000 / 20)

    class Opt:
        param_groups = [dict(lr=0.0)]
    its = [0, 1, 299, 300, 301, 3000, 5999]
    lrs = [adjust_learning_rate(Opt(), i, Cfg) for i in its]
    lin = torch.nn.Linear(3, 2)
    bn = torch.nn.BatchNorm1d(2)
    m = torch.nn.Sequential(lin, bn)
    ema = ExponentialMovingAverage(m, 0.99)
    ema.register()
    w0 = lin.weight.detach().clone().numpy()
    with torch.no_grad():
        lin.weight.add_(1.0)
    ema.update()
    save('lr_ema.npz', its=np.array(its), lrs=np.array(lrs, np.float64), w0=w0,
         w1=lin.weight.detach().numpy(), shadow=ema.shadow['0.weight'].numpy(),
         shadow_keys=np.array(sorted(ema.shadow.keys())))


def build_ref_model(resnet_type='resnet101'):
    return Deeplabv2(dict(backbone=dict(resnet_type=resnet_type, output_stride=16, pretrained=False),
                          multi_layer=True, cascade=False, use_ppm=True,
                          ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                          inchannels=2048, num_classes=6, is_ins_norm=True))


def gold_model():
    """ResNet-101 DeepLabV2(PPM) at 2x3x64x64, seeded weights (oracle.model.init_state_dict(seed=1)),
    one full SSL step composed exactly like tools/train_ssl_reg.py:198-241."""
    m = build_ref_model()
    manifest = [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, 'state_dict_manifest.json'), 'w') as f:
        json.dump(manifest, f)
    sd = omodel.init_state_dict('resnet101', 6, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train()
    masks = {}

    def hook(name):
        def fn(mod, inp, out):
            i, o = inp[0].detach(), out.detach()
            keep = ((o != 0).flatten(2).any(-1) | (i == 0).flatten(2).all(-1))
            masks.setdefault(name, []).append(keep.numpy().astype(np.uint8))
        return fn
    m.layer5.conv_last[3].register_forward_hook(hook('m5'))
    m.layer6.conv_last[3].register_forward_hook(hook('m6'))
    g = torch.Generator().manual_seed(2333)
    b = 2
    xs = torch.randn(b, 3, 64, 64, generator=g)
    xt = torch.randn(b, 3, 64, 64, generator=g).clamp(max=1.0)
    rng = np.random.default_rng(2333)
    lab_s = torch.from_numpy(np.kron(rng.integers(-1, 6, size=(b, 4, 4)), np.ones((16, 16), np.int64)))
    soft_t = torch.softmax(torch.randn(b, 6, 64, 64, generator=g) * 3, 1)
    regs = torch.from_numpy(random_regions(rng, b, 64, 64, 12))[:, None]
    protos = torch.randn(6, 2048, generator=g)
    al = Aligner(logger=_Log(), feat_channels=2048, class_num=6, ignore_label=-1, decay=0.996, resume=None)
    al.prototypes = protos.clone()
    hom = Homogenizer(percent=0.5, class_num=6, ignore_label=-1)
    ce = CrossEntropy(ignore_label=-1, class_balancer=None)
    torch.manual_seed(2333)
    s1, s2, fs = m(xs)
    t1, t2, ft = m(xt)
    soft2 = al.label_refine(None, ft, [t1, t2], soft_t, refine=True, mode='all', temp=2.0)
    hard = pseudo_selection(soft2, 0.8, 0.6, 'tensor', -1)
    hard2 = hom(hard, regs.squeeze(1))
    al.update_prototype(fs, lab_s)
    ls = loss_calc([s1, s2], lab_s, loss_fn=ce, multi=True)
    lt = loss_calc([t1, t2], hard2, loss_fn=ce, multi=True)
    loss = ls + lt
    loss.backward()
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values())).item()
    sel = ['encoder.resnet.conv1.weight', 'encoder.resnet.bn1.weight', 'encoder.resnet.bn1.bias',
           'layer5.conv_last.4.weight', 'layer5.conv_last.4.bias', 'layer6.conv_last.1.weight',
           'encoder.resnet.layer4.2.bn3.bias', 'layer5.ppm.0.2.weight', 'layer6.ppm.3.2.bias']
    grads = {('grad:' + k): named[k].grad.numpy() for k in sel}
    grads['grad:layer3.10.conv2.weight[:2]'] = named['encoder.resnet.layer3.10.conv2.weight'].grad[:2].numpy()
    grads['grad:layer5.conv_last.0.weight[:1,:64]'] = named['layer5.conv_last.0.weight'].grad[:1, :64].numpy()
    sdn = m.state_dict()
    # eval-mode teacher output with the (twice updated) running stats
    m.eval()
    with torch.no_grad():
        probs = m(xt)
    save('model_small.npz', xs=xs.numpy(), xt=xt.numpy(), lab_s=lab_s.numpy().astype(np.int8),
         soft_t=soft_t.numpy(), regs=regs.numpy().astype(np.int32), protos=protos.numpy(),
         m5=np.stack(masks['m5']), m6=np.stack(masks['m6']),
         s1=s1.detach().numpy(), s2=s2.detach().numpy(), t1=t1.detach().numpy(), t2=t2.detach().numpy(),
         feat_s=fs.detach().numpy()[:, :32], feat_t=ft.detach().numpy()[:, :32],
         soft2=soft2.detach().numpy(), hard=hard.numpy().astype(np.int8), hard2=hard2.numpy().astype(np.int8),
         protos_new=al.prototypes.numpy(), loss_s=ls.detach().numpy(), loss_t=lt.detach().numpy(),
         grad_norm=np.float64(gn),
         bn1_rm=sdn['encoder.resnet.bn1.running_mean'].numpy(), bn1_rv=sdn['encoder.resnet.bn1.running_var'].numpy(),
         l5bn_rm=sdn['layer5.conv_last.1.running_mean'].numpy(), l5bn_rv=sdn['layer5.conv_last.1.running_var'].numpy(),
         nbt=sdn['encoder.resnet.bn1.num_batches_tracked'].numpy(), probs=probs.numpy(), **grads)


MID_GRADS = ['encoder.resnet.conv1.weight', 'encoder.resnet.bn1.weight', 'encoder.resnet.layer1.0.conv1.weight',
             'encoder.resnet.layer1.2.bn3.weight', 'encoder.resnet.layer2.1.conv2.weight[:8]', 'encoder.resnet.layer2.3.bn3.bias',
             'encoder.resnet.layer3.0.downsample.0.weight[:8]', 'encoder.resnet.layer3.5.conv2.weight[:4]',
             'encoder.resnet.layer3.11.bn2.weight', 'encoder.resnet.layer3.17.conv1.weight[:16]',
             'encoder.resnet.layer3.22.conv3.weight[:64]', 'encoder.resnet.layer4.0.conv2.weight[:2]',
             'encoder.resnet.layer4.2.bn3.bias', 'layer5.ppm.3.1.weight[:8]', 'layer5.conv_last.0.weight[:8]',
             'layer6.conv_last.1.weight', 'layer6.conv_last.4.weight']
MID_RES_GAMMA = 0.02        # residual-branch gain of this fixture's weights (tests/golden/derive_tolerances.py: FULL_RES_GAMMA)


def gold_model128():
    """The reference's SSL step (tools/train_ssl_reg.py:198-241, composed as in gold_model) on ResNet-101 at 2 x 3 x 128 x 128
    -- 8 x 8 feature maps: the deep BatchNorm layers see 128 values per channel and domain instead of the 32 of
    model_small.npz -- with weights whose residual branches have gain 0.02 (oracle.model.init_state_dict(seed=3,
    res_gamma=0.02)): the better-conditioned reference-minted fixture of the GPU suite.  Stores the inputs, both losses,
    the gradient norm, refined soft labels (fp16), pseudo labels, prototypes and 17 gradient tensors (slices of the large
    ones) spread over the depth of the network."""
    m = build_ref_model()
    sd = omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=MID_RES_GAMMA)
    m.load_state_dict(sd, strict=True)
    m.train()
    masks = {}

    def hook(name):
        def fn(mod, inp, out):
            i, o = inp[0].detach(), out.detach()
            keep = ((o != 0).flatten(2).any(-1) | (i == 0).flatten(2).all(-1))
            masks.setdefault(name, []).append(keep.numpy().astype(np.uint8))
        return fn
    m.layer5.conv_last[3].register_forward_hook(hook('m5'))
    m.layer6.conv_last[3].register_forward_hook(hook('m6'))
    g = torch.Generator().manual_seed(4242)
    b, S = 2, 128
    xs = torch.randn(b, 3, S, S, generator=g)
    xt = torch.randn(b, 3, S, S, generator=g).clamp(max=1.0)
    rng = np.random.default_rng(4242)
    lab_s = torch.from_numpy(np.kron(rng.integers(-1, 6, size=(b, S // 16, S // 16)), np.ones((16, 16), np.int64)))
    soft_t = torch.softmax(torch.randn(b, 6, S, S, generator=g) * 3, 1)
    regs = torch.from_numpy(random_regions(rng, b, S, S, 24))[:, None]
    protos = torch.randn(6, 2048, generator=g)
    al = Aligner(logger=_Log(), feat_channels=2048, class_num=6, ignore_label=-1, decay=0.996, resume=None)
    al.prototypes = protos.clone()
    hom = Homogenizer(percent=0.5, class_num=6, ignore_label=-1)
    ce = CrossEntropy(ignore_label=-1, class_balancer=None)
    torch.manual_seed(4242)
    s1, s2, fs = m(xs)
    t1, t2, ft = m(xt)
    soft2 = al.label_refine(None, ft, [t1, t2], soft_t, refine=True, mode='all', temp=2.0)
    hard = pseudo_selection(soft2, 0.8, 0.6, 'tensor', -1)
    hard2 = hom(hard, regs.squeeze(1))
    al.update_prototype(fs, lab_s)
    ls = loss_calc([s1, s2], lab_s, loss_fn=ce, multi=True)
    lt = loss_calc([t1, t2], hard2, loss_fn=ce, multi=True)
    (ls + lt).backward()
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values())).item()
    grads = {}
    for k in MID_GRADS:
        name, _, sl = k.partition('[')
        gk = named[name].grad
        if sl:
            gk = gk[:int(sl[1:-1])]
        # (single output channels are a poor statistic -- per-channel cosines under bf16 noise spread from 0.89 to 0.999 --
        #  so slices keep several; the large head-conv slice is stored as fp16)
        grads['grad:' + k] = gk.numpy().astype(np.float16) if gk.numel() > 100000 else gk.numpy()
    sdn = m.state_dict()
    save('model_mid.npz', xs=xs.numpy(), xt=xt.numpy(), lab_s=lab_s.numpy().astype(np.int8),
         soft_t=soft_t.numpy(), regs=regs.numpy().astype(np.int32), protos=protos.numpy(),
         m5=np.stack(masks['m5']), m6=np.stack(masks['m6']),
         s1=s1.detach().numpy(), s2=s2.detach().numpy(), t1=t1.detach().numpy(), t2=t2.detach().numpy(),
         soft2=soft2.detach().numpy().astype(np.float16), hard=hard.numpy().astype(np.int8), hard2=hard2.numpy().astype(np.int8),
         protos_new=al.prototypes.numpy(), loss_s=ls.detach().numpy(), loss_t=lt.detach().numpy(),
         grad_norm=np.float64(gn), bn1_rm=sdn['encoder.resnet.bn1.running_mean'].numpy(),
         nbt=sdn['encoder.resnet.bn1.num_batches_tracked'].numpy(), **grads)


def gold_align():
    """One stage-2 iteration on the reference model, composed exactly like tools/train_align_reg.py:144-196 (defaults
    --align-domain 0, --refine-label 1, --refine-mode all, --refine-temp 2, --sam-refine, --pcl-temp 8), same inputs as
    gold_model so the fixture stays small: only what differs is stored."""
    import torch.nn.functional as tnf
    from regda.loss import PrototypeContrastiveLoss
    src = np.load(os.path.join(HERE, 'model_small.npz'))
    m = build_ref_model()
    sd = omodel.init_state_dict('resnet101', 6, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train()
    masks = {}

    def hook(name):
        def fn(mod, inp, out):
            i, o = inp[0].detach(), out.detach()
            keep = ((o != 0).flatten(2).any(-1) | (i == 0).flatten(2).all(-1))
            masks.setdefault(name, []).append(keep.numpy().astype(np.uint8))
        return fn
    m.layer5.conv_last[3].register_forward_hook(hook('m5'))
    m.layer6.conv_last[3].register_forward_hook(hook('m6'))
    xs, xt = torch.from_numpy(src['xs']), torch.from_numpy(src['xt'])
    lab_s = torch.from_numpy(src['lab_s'].astype(np.int64))
    regs = torch.from_numpy(src['regs'].astype(np.int64))
    al = Aligner(logger=_Log(), feat_channels=2048, class_num=6, ignore_label=-1, decay=0.999, resume=None)
    al.prototypes = torch.from_numpy(src['protos']).clone()
    hom = Homogenizer(percent=0.5, class_num=6, ignore_label=-1)
    ce = CrossEntropy(ignore_label=-1, class_balancer=None)
    pcl = PrototypeContrastiveLoss(temperature=8.0, ignore_label=-1)
    torch.manual_seed(77)
    s1, s2, fs = m(xs)
    label_s_down = al.update_prototype(fs, lab_s)
    t1, t2, ft = m(xt)
    x1 = tnf.interpolate(t1, xt.shape[-2:], mode='bilinear', align_corners=True)
    x2 = tnf.interpolate(t2, xt.shape[-2:], mode='bilinear', align_corners=True)
    soft = ((x1.softmax(dim=1) + x2.softmax(dim=1)) * 0.5).detach()
    soft = al.label_refine(None, ft, [t1, t2], soft, refine=True, mode='all', temp=2.0)
    hard = pseudo_selection(soft, cutoff_top=0.8, cutoff_low=0.6, return_type='tensor', ignore_label=-1)
    hard = hom(hard, regs.squeeze(dim=1))
    label_t = al.downscale_gt(hard)
    loss_seg = loss_calc([s1, s2], lab_s, loss_fn=ce, multi=True)
    loss_align = (pcl(al.prototypes, fs, label_s_down) + pcl(al.prototypes, ft, label_t)) * 0.5
    loss = loss_seg + loss_align
    loss.backward()
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    sel = ['encoder.resnet.conv1.weight', 'encoder.resnet.bn1.bias', 'layer5.conv_last.4.weight',
           'encoder.resnet.layer4.2.bn3.bias', 'layer6.ppm.3.2.bias']
    grads = {('grad:' + k): named[k].grad.numpy() for k in sel}
    grads['grad:layer3.10.conv2.weight[:2]'] = named['encoder.resnet.layer3.10.conv2.weight'].grad[:2].numpy()
    save('align_small.npz', m5=np.stack(masks['m5']), m6=np.stack(masks['m6']),
         label_s_down=label_s_down.numpy().astype(np.int8), soft=soft.numpy(), hard=hard.numpy().astype(np.int8),
         label_t=label_t.numpy().astype(np.int8), protos_new=al.prototypes.numpy(),
         loss_seg=loss_seg.detach().numpy(), loss_align=loss_align.detach().numpy(), grad_norm=np.float64(gn), **grads)


def gold_tta():
    """The reference's own pre_slide / tta_predict (regda/utils/tools.py:61-97,132-152) on a small deterministic
    'model' that is not equivariant under flips / rotations, plus the align_corners=True soft-label resize of
    pseudo_generation.py:135.  ttach is the restatement in _refstubs (un-vendored dependency)."""
    import torch.nn.functional as F
    from regda.utils import tools as rtools
    torch.manual_seed(17)
    wgt, bias = torch.randn(5, 3, 3, 3) * 0.7, torch.randn(5) * 0.3

    def model(x):                       # stands in for Deeplabv2.eval(): per-pixel class probabilities
        return torch.softmax(F.conv2d(x, wgt, bias, padding=1), dim=1)
    out = dict(wgt=wgt.numpy(), bias=bias.numpy())
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference allocates with .cuda(); CPU-only box
    try:
        cases = [((1, 3, 16, 16), (16, 16)),      # one window, no padding (the 512 x 512 production case)
                 ((1, 3, 40, 24), (16, 16)),      # 4 x 2 windows, overlaps, last window re-aligned
                 ((1, 3, 12, 20), (16, 16))]      # image smaller than the tile in one dimension: zero padding
        for i, (shape, tile) in enumerate(cases):
            img = torch.randn(*shape)
            out[f'img{i}'] = img.numpy()
            out[f'tile{i}'] = np.array(tile)
            for tta in (False, True):
                out[f'probs{i}_tta{int(tta)}'] = rtools.pre_slide(model, img, num_classes=5, tile_size=tile, tta=tta).numpy()
        out['tta_single'] = rtools.tta_predict(model, torch.from_numpy(out['img0'])).numpy()
    finally:
        torch.Tensor.cuda = cuda
    cls = torch.from_numpy(out['probs1_tta1'])
    out['resized'] = F.interpolate(cls, (64, 48), mode='bilinear', align_corners=True).squeeze(dim=0).numpy()
    save('tta.npz', **out)


def gold_pcl():
    """PrototypeContrastiveLoss (regda/loss.py:10-47) and its gradient w.r.t. the features, from the reference class."""
    from regda.loss import PrototypeContrastiveLoss
    torch.manual_seed(23)
    b, K, h, w, C = 2, 64, 5, 7, 6
    out = {}
    for i, (temp, frac_ign) in enumerate([(8.0, 0.3), (2.0, 0.0), (8.0, 0.97)]):
        feat = (torch.randn(b, K, h, w) * 1.5 + 0.2).requires_grad_(True)
        protos = torch.randn(C, K)
        lab = torch.randint(0, C, (b, h, w))
        lab[torch.rand(b, h, w) < frac_ign] = -1
        if i == 2:
            lab[0, 0, 0] = 3
        loss = PrototypeContrastiveLoss(temperature=temp, ignore_label=-1)(protos, feat, lab)
        loss.backward()
        out.update({f'feat{i}': feat.detach().numpy(), f'protos{i}': protos.numpy(), f'lab{i}': lab.numpy(),
                    f'temp{i}': np.float32(temp), f'loss{i}': loss.detach().numpy(), f'gfeat{i}': feat.grad.numpy()})
    save('pcl.npz', **out)


def gold_aspp():
    """ASPP heads (use_ppm=False): (1) the reference's Classifier_Module alone (Encoder.py:68-84) on a small map,
    forward and all gradients; (2) the reference Deeplabv2(use_ppm=False) ResNet-101 at 2x3x64x64 with
    oracle.model.init_state_dict(seed=2, head='aspp'): train-mode outputs, a loss, selected gradients, eval probs."""
    from regda.models.Encoder import Classifier_Module
    torch.manual_seed(5)
    cm = Classifier_Module(64, [6, 12, 18, 24], [6, 12, 18, 24], 6)
    x = torch.randn(2, 64, 20, 28, requires_grad=True)
    gy = torch.randn(2, 6, 20, 28)
    y = cm(x)
    (y * gy).sum().backward()
    out = dict(cm_x=x.detach().numpy(), cm_gy=gy.numpy(), cm_y=y.detach().numpy(), cm_gx=x.grad.numpy())
    for i, m in enumerate(cm.conv2d_list):
        out[f'cm_w{i}'], out[f'cm_b{i}'] = m.weight.detach().numpy(), m.bias.detach().numpy()
        out[f'cm_gw{i}'], out[f'cm_gb{i}'] = m.weight.grad.numpy(), m.bias.grad.numpy()
    m = Deeplabv2(dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False),
                       multi_layer=True, cascade=False, use_ppm=False, inchannels=2048, num_classes=6,
                       is_ins_norm=True))
    sd = omodel.init_state_dict('resnet101', 6, seed=2, head='aspp')
    m.load_state_dict(sd, strict=True)
    out['keys'] = np.array(list(m.state_dict().keys()))
    m.train()
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(2, 3, 64, 64, generator=g)
    lab = torch.from_numpy(np.kron(np.random.default_rng(7).integers(-1, 6, size=(2, 4, 4)), np.ones((16, 16), np.int64)))
    x1, x2, feat = m(xs)
    loss = loss_calc([x1, x2], lab, loss_fn=CrossEntropy(ignore_label=-1, class_balancer=None), multi=True)
    loss.backward()
    named = dict(m.named_parameters())
    for k in ('layer5.conv2d_list.0.bias', 'layer6.conv2d_list.3.bias', 'encoder.resnet.bn1.weight'):
        out['grad:' + k] = named[k].grad.numpy()
    out['grad:layer5.conv2d_list.1.weight[:, :32]'] = named['layer5.conv2d_list.1.weight'].grad[:, :32].numpy()
    m.eval()
    with torch.no_grad():
        probs = m(xs)
    out.update(xs=xs.numpy(), lab=lab.numpy().astype(np.int8), x1=x1.detach().numpy(), x2=x2.detach().numpy(),
               feat=feat.detach().numpy()[:, :32], loss=loss.detach().numpy(), probs=probs.numpy())
    save('aspp.npz', **out)


EVAL_WARM_CLS_GAIN = 0.1


def gold_eval_warm():
    """The eval branch (Encoder.py:152-155 -- the teacher's output) on WARM BatchNorm statistics: the reference ResNet-101
    (oracle.model.init_state_dict(seed=3, res_gamma=0.02), the better-conditioned weights of model_mid.npz) sees 20
    train-mode forwards of random 2 x 3 x 128 x 128 batches (running statistics 0.9^20 = 12 % away from what they converge
    to; model_small.npz's eval output was taken two updates from the (0, 1) initialisation, where the logits saturate), then
    one eval forward.  Saved: the eval input, the reference's probabilities and EVERY BatchNorm buffer of the reference at
    that point (so that the HIP test starts from the reference's exact state and isolates the eval path)."""
    m = build_ref_model()
    sd = omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=MID_RES_GAMMA)
    # classifier gain 0.1: random-init 512 -> 6 classifiers give logits of +-30 and one-hot heads; at 0.1 the probabilities
    # are soft (median top probability 0.28) and a probability error is visible instead of clipped
    for head in ('layer5', 'layer6'):
        sd[f'{head}.conv_last.4.weight'] = sd[f'{head}.conv_last.4.weight'] * EVAL_WARM_CLS_GAIN
    m.load_state_dict(sd, strict=True)
    m.train()
    g = torch.Generator().manual_seed(4242)
    torch.manual_seed(4242)                 # Dropout2d draws of the train-mode forwards
    with torch.no_grad():
        for _ in range(20):
            m(torch.randn(2, 3, 128, 128, generator=g))
        m.eval()
        xe = torch.randn(1, 3, 128, 128, generator=g).clamp(max=1.0)
        probs = m(xe)
    sdn = m.state_dict()
    bufs = {k: v.numpy() for k, v in sdn.items() if k.endswith(('running_mean', 'running_var'))}
    names = sorted(bufs)
    flat = np.concatenate([bufs[k].ravel() for k in names]).astype(np.float32)
    save('eval_warm.npz', xe=xe.numpy(), probs=probs.numpy(), cls_gain=np.float32(EVAL_WARM_CLS_GAIN), buffer_names=np.array(names),
         buffer_sizes=np.array([bufs[k].size for k in names]), buffers=flat,
         nbt=sdn['encoder.resnet.bn1.num_batches_tracked'].numpy())
    conf = probs.max(1)[0]
    print('eval_warm: argmax histogram', np.bincount(probs.argmax(1).numpy().ravel(), minlength=6).tolist(),
          'max-prob quantiles', np.quantile(conf.numpy(), [0.1, 0.5, 0.9]).round(3).tolist())


if __name__ == '__main__':
    which = sys.argv[1:] or ['lrh', 'pseudo', 'downscale', 'refine', 'loss', 'lr_ema', 'model', 'model128', 'tta', 'pcl', 'align', 'aspp', 'regions', 'refine_sup', 'eval_warm']
    for w in which:
        globals()['gold_' + w]()
