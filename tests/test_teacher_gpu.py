"""GPU: the teacher / pseudo-label harness through the C ABI against torch permutations (bit-exact)
and against the golden vectors minted from the reference's own pre_slide / tta_predict (tests/golden/tta.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from regda_amd import ops
    return ops


def fake_model(g):
    wgt, bias = torch.from_numpy(g['wgt']).cuda(), torch.from_numpy(g['bias']).cuda()

    class M:
        num_classes = 5

        def eval(self):
            return self

        def __call__(self, x):
            return torch.softmax(F.conv2d(x, wgt, bias, padding=1), dim=1)
    return M()


@pytest.mark.parametrize('shape', [(2, 3, 8, 8), (1, 5, 6, 10)])
def test_dihedral_views_are_exact_permutations(ops, shape):
    x = torch.randn(*shape, device='cuda')
    for f in (False, True):
        for k in range(4):
            aug = ops.dihedral(x, f, k, True)
            ref = torch.rot90(x.flip(3) if f else x, k, (2, 3))
            assert torch.equal(aug, ref), (f, k)
            back = ops.dihedral(aug, f, (4 - k) % 4, False)          # deaugment_mask: rot90(-k), then the flip
            assert torch.equal(back, x), (f, k)
    acc = ops.dihedral(x, True, 1, True, scale=0.5)
    ops.dihedral(x, True, 1, True, dst=acc, scale=0.25, accumulate=True)
    assert torch.allclose(acc, 0.75 * torch.rot90(x.flip(3), 1, (2, 3)), rtol=0, atol=1e-6)


def test_pre_slide_and_tta_match_the_reference_goldens(gold):
    from regda_amd.utils.tools import pre_slide, tta_predict
    g = gold('tta.npz')
    model = fake_model(g)
    for i in range(3):
        img, tile = torch.from_numpy(g[f'img{i}']).cuda(), tuple(int(v) for v in g[f'tile{i}'])
        for tta in (0, 1):
            got = pre_slide(model, img, num_classes=5, tile_size=tile, tta=bool(tta))
            # fp32 everywhere; the only difference to the CPU reference is conv2d's summation order
            np.testing.assert_allclose(got.cpu().numpy(), g[f'probs{i}_tta{tta}'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tta_predict(model, torch.from_numpy(g['img0']).cuda()).cpu().numpy(), g['tta_single'],
                               rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        tta_predict(model, torch.zeros(2, 3, 16, 16, device='cuda'))


def test_soft_label_resize_align_corners(ops, gold):
    g = gold('tta.npz')
    cls = torch.from_numpy(g['probs1_tta1']).cuda()
    got = ops.resize_bilinear_ac(cls, (64, 48)).squeeze(0)
    np.testing.assert_allclose(got.cpu().numpy(), g['resized'], rtol=0, atol=1e-6)
    x = torch.randn(2, 3, 7, 9, device='cuda')
    for size in ((7, 9), (1, 1), (13, 4), (20, 31)):
        ref = F.interpolate(x, size, mode='bilinear', align_corners=True)
        assert torch.allclose(ops.resize_bilinear_ac(x, size), ref, rtol=0, atol=1e-6), size


def test_gener_target_pseudo_writes_the_soft_label_files(tmp_path, gold):
    """save_prob=True: one `<fname>.pt` per tile holding a (C, H, W) fp32 CPU tensor = resize(pre_slide(..., tta=True))
    (pseudo_generation.py:128-136, read back by basedata.py:86)."""
    from regda_amd.gast.pseudo_generation import gener_target_pseudo
    from oracle import teacher
    g = gold('tta.npz')
    model = fake_model(g)
    imgs = [torch.from_numpy(g['img1']), torch.from_numpy(g['img1']).flip(2)]
    loader = [(im, {'fname': [f'tile_{i}.tif']}) for i, im in enumerate(imgs)]

    class Cfg:
        NUM_CLASSES = 5
        SNAPSHOT_DIR = None
    out = str(tmp_path / 'pseudo_label')
    gener_target_pseudo(Cfg, model, loader, out, slide=True, save_prob=True, size=(64, 48), ignore_label=-1)
    wgt, bias = torch.from_numpy(g['wgt']), torch.from_numpy(g['bias'])
    cpu_model = lambda x: torch.softmax(F.conv2d(x, wgt, bias, padding=1), dim=1)       # noqa: E731
    for i, im in enumerate(imgs):
        t = torch.load(os.path.join(out, f'tile_{i}.tif.pt'))
        assert t.device.type == 'cpu' and t.dtype == torch.float32 and tuple(t.shape) == (5, 64, 48)
        ref = teacher.soft_label(teacher.pre_slide(cpu_model, im, num_classes=5, tile_size=(512, 512), tta=True), (64, 48))
        np.testing.assert_allclose(t.numpy(), ref.numpy(), rtol=0, atol=3e-6)
    # save_prob=False (pseudo_generation.py:143-150): the selected hard labels + 1 as a uint8 image named <fname>
    from PIL import Image
    from oracle import labels as olab
    hard_dir = str(tmp_path / 'pseudo_hard')
    for select in (True, False):
        Cfg.PSEUDO_SELECT = select
        # slide=False: the 40 x 24 fixture is smaller than pre_slide's 512 x 512 tile, for which the reference's window
        # arithmetic yields no window at all (NaN probabilities, reproduced by the soft-label leg above)
        gener_target_pseudo(Cfg, model, loader, hard_dir, slide=False, save_prob=False, size=(40, 24), ignore_label=-1)   # (no resize on this branch: size = the tile's)
        for i, im in enumerate(imgs):
            arr = np.array(Image.open(os.path.join(hard_dir, f'tile_{i}.tif')))
            assert arr.dtype == np.uint8 and arr.shape == (40, 24)
            probs = cpu_model(im)
            want = olab.pseudo_selection(probs.numpy(), 0.8, 0.6, -1) if select else probs.argmax(1).numpy()
            # (probabilities within 3e-6 of the oracle's: a pixel sitting exactly on a threshold / an exact tie may flip)
            assert (arr != (want + 1).reshape(40, 24)).mean() < 2e-3


def test_batched_tta_equals_view_by_view_on_the_real_network():
    """tta_predict feeds the eight views through Deeplabv2.eval() as one batch; per-view forwards must agree (same
    kernels, eval-mode BN) up to the bf16 activations' tile-order noise."""
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.utils.tools import tta_predict
    from regda_amd import ops
    from oracle import model as omodel
    rt = 'resnet17t'
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True, cascade=False,
                       use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6,
                       is_ins_norm=True))
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=3), strict=True)
    m.eval()
    img = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(8)).cuda()
    got = tta_predict(m, img)
    acc = torch.zeros_like(got)
    for f in (False, True):
        for k in range(4):
            p = m(ops.dihedral(img, f, k, True))
            ops.dihedral(p.contiguous(), f, (4 - k) % 4, False, dst=acc, scale=0.125, accumulate=True)
    assert tuple(got.shape) == (1, 6, 64, 64)
    assert torch.allclose(got.sum(1), torch.ones_like(got.sum(1)), atol=1e-4)
    assert (got - acc).abs().max().item() < 2e-2


def test_argmax_and_confusion_matrix_are_exact(ops):
    from oracle import evalpath
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn(3, 6, 37, 29, generator=g) * 2, 1)
    probs[0, :, 5, 5] = 0.25                                     # a tie: the first maximum wins
    pred = ops.argmax_nchw(probs.cuda())
    assert torch.equal(pred.cpu(), probs.argmax(1)) and int(pred[0, 5, 5]) == 0
    y_true = torch.randint(-1, 6, (3, 37, 29), generator=g)
    cm = torch.zeros(6, 6, dtype=torch.int64, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.confusion_accumulate(y_true.cuda(), pred, cm, flag)
    ops.confusion_accumulate(y_true.cuda(), pred, cm, flag)      # accumulates
    ref = evalpath.confusion_matrix(y_true.numpy(), pred.cpu().numpy(), 6)
    assert np.array_equal(cm.cpu().numpy(), 2 * ref) and int(flag) == 0
    big_t = torch.randint(-1, 6, (4, 512, 512), generator=g)
    big_p = torch.randint(0, 6, (4, 512, 512), generator=g)
    cm.zero_()
    ops.confusion_accumulate(big_t.cuda(), big_p.cuda(), cm, flag)
    assert np.array_equal(cm.cpu().numpy(), evalpath.confusion_matrix(big_t.numpy(), big_p.numpy(), 6))
    assert int(cm.sum()) == int((big_t >= 0).sum())
    ops.confusion_accumulate(torch.full((8,), 6, dtype=torch.int64).cuda(), torch.zeros(8, dtype=torch.int64).cuda(), cm, flag)
    assert int(flag) == 1                                        # label out of range: flagged, not counted


def test_evaluate_returns_the_oracle_miou(gold):
    """regda_amd.utils.eval.evaluate over a two-tile loader: table + mIoU with class 0 dropped (IsprsDA) equal the
    oracle's summary of the oracle's predictions."""
    from regda_amd.utils.eval import evaluate
    from oracle import evalpath, teacher
    g = gold('tta.npz')
    model = fake_model(g)
    gen = torch.Generator().manual_seed(4)
    imgs = [torch.randn(1, 3, 40, 24, generator=gen) for _ in range(2)]
    gts = [torch.randint(-1, 5, (1, 40, 24), generator=gen) for _ in range(2)]
    loader = [(im, {'cls': gt, 'fname': ['t.tif']}) for im, gt in zip(imgs, gts)]

    class Cfg:
        DATASETS = 'IsprsDA'
        NUM_CLASSES = 5
        SNAPSHOT_DIR = None
    table, miou = evaluate(model, Cfg, is_training=True, dataloader=loader, slide=True, tta=False)
    wgt, bias = torch.from_numpy(g['wgt']), torch.from_numpy(g['bias'])
    cpu_model = lambda x: torch.softmax(F.conv2d(x, wgt, bias, padding=1), dim=1)       # noqa: E731
    cm = np.zeros((5, 5), np.int64)
    for im, gt in zip(imgs, gts):
        pred = teacher.pre_slide(cpu_model, im, num_classes=5, tile_size=(512, 512), tta=False).argmax(1)
        cm += evalpath.confusion_matrix(gt.numpy(), pred.numpy(), 5)
    ref = evalpath.summary(cm, ignore_labels=[0])
    assert miou == ref['miou'] and 'mean' in table and len(table.splitlines()) == 1 + 4 + 1
    with pytest.raises(ValueError):
        evaluate(model, Cfg, is_training=True)
