"""GPU: the ASPP heads (Classifier_Module, use_ppm=False) -- the HIP form (one 1x1 convolution for all
taps + dilated gather / scatter, csrc/aspp_kernels.hip) against the reference goldens and the oracle.

Tolerances: Z and the activations are bf16, accumulation fp32 -> relative L2 < 1 % on the head alone (inputs rounded
to bf16 first on both sides), the same network-level bounds as tests/test_model_gpu.py for the whole model."""
import numpy as np
import pytest
import torch

from oracle import labelpath as opath
from oracle import model as omodel

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DIL = (6, 12, 18, 24)


def l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def build(rt):
    from regda_amd.models.Encoder import Deeplabv2
    return Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                          cascade=False, use_ppm=False, inchannels=2048, num_classes=6, is_ins_norm=True))


def test_head_ops_against_reference_module(gold):
    """Forward and all three gradients of the head, composed from the C-ABI calls exactly like the model does,
    against the reference Classifier_Module's golden (K = 64 input channels, 20x28 map, so every dilation both hits
    and misses the border).  Head 2 runs the same filters scaled by -0.5 (checked against the oracle)."""
    from regda_amd import ops
    g = gold('aspp.npz')
    dev = 'cuda'
    N, K, h, w, C = 2, 64, 20, 28, 6
    rb = lambda t: t.to(BF).float()
    x = rb(torch.from_numpy(g['cm_x']))
    ws = [rb(torch.from_numpy(g[f'cm_w{i}'])) for i in range(4)]
    bs = [torch.from_numpy(g[f'cm_b{i}']) for i in range(4)]
    gy = rb(torch.from_numpy(g['cm_gy']))
    # references on the rounded operands (fp32 CPU); the golden itself differs from them only by that rounding
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in ws]
    y_ref = omodel.aspp_head(xr, wr, bs)
    assert l2(y_ref, torch.from_numpy(g['cm_y'])) < 1e-2
    y2_ref = omodel.aspp_head(x, [-0.5 * t for t in ws], bs)
    (y_ref * gy).sum().backward()
    # stacked filter [ZC][1][K]: row ((head*4 + d)*C + c)*9 + tap
    rows = 2 * 4 * C * 9
    zc = (rows + 63) // 64 * 64
    wz = torch.zeros(zc, 1, K)
    for hd, scale in ((0, 1.0), (1, -0.5)):
        for d in range(4):
            blk = (ws[d] * scale).permute(0, 2, 3, 1).reshape(C * 9, K)         # [C][3][3][K]
            wz[(hd * 4 + d) * C * 9:(hd * 4 + d + 1) * C * 9, 0] = blk
    wz = wz.to(BF).to(dev)
    xp = x.permute(0, 2, 3, 1).reshape(N * h * w, K).to(BF).to(dev)
    z = torch.empty(N * h * w, zc, dtype=BF, device=dev)
    ops.conv2d(xp, wz, z, N, h, w, h, w, 1, 1, 1, 0, 1, 0)
    biases = [b.to(dev) for b in bs] * 2
    o1 = torch.empty(N, C, h, w, device=dev)
    o2 = torch.empty(N, C, h, w, device=dev)
    ops.aspp_gather(z, biases, o1, o2, N, h, w, C, DIL)
    assert l2(o1, y_ref.detach()) < 1e-2 and l2(o2, y2_ref) < 1e-2
    # backward: gradient only through head 1
    dz = torch.full((N * h * w, zc), 7.0, dtype=BF, device=dev)
    dbs = [torch.zeros(C, device=dev) for _ in range(8)]
    ops.aspp_scatter(gy.to(dev), torch.zeros(N, C, h, w, device=dev), dz, dbs, N, h, w, C, DIL)
    assert float(dz[:, rows:].abs().max()) == 0.0 and float(dz[:, rows // 2:rows].abs().max()) == 0.0
    dx = torch.empty(N * h * w, K, dtype=BF, device=dev)
    wzt = wz.view(zc, K).t().contiguous().view(K, 1, zc)
    ops.conv2d(dz, wzt, dx, N, h, w, h, w, 1, 1, 1, 0, 1, 0)
    gx = dx.float().reshape(N, h, w, K).permute(0, 3, 1, 2)
    assert l2(gx, xr.grad) < 1e-2 and l2(gx, torch.from_numpy(g['cm_gx'])) < 2e-2
    gz = torch.zeros(zc, 1, K, device=dev)
    ops.conv2d_wgrad(xp, dz, gz, N, h, w, h, w, 1, 1, 1, 0, 1)
    for d in range(4):
        got = gz[d * C * 9:(d + 1) * C * 9, 0].reshape(C, 3, 3, K).permute(0, 3, 1, 2)
        assert l2(got, wr[d].grad) < 1e-2 and l2(got, torch.from_numpy(g[f'cm_gw{d}'])) < 2e-2, d
        np.testing.assert_allclose(dbs[d].cpu().numpy(), gy.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-4)
        assert l2(dbs[d], torch.from_numpy(g[f'cm_gb{d}'])) < 1e-2
        assert float(dbs[4 + d].abs().max()) == 0.0
    assert float(gz[rows // 2:].abs().max()) == 0.0


def test_state_dict_layout_and_resnet101_golden(gold):
    """ResNet-101 with ASPP heads: the reference's state_dict keys in order, train-mode logits / feat and eval-mode
    probabilities against the reference golden (relative L2 < 10 %, the fp32-vs-bf16 bound of test_model_gpu.py)."""
    g = gold('aspp.npz')
    m = build('resnet101')
    sd = omodel.init_state_dict('resnet101', 6, seed=2, head='aspp')
    m.load_state_dict(sd, strict=True)
    assert list(m.state_dict().keys()) == list(g['keys'])
    for k, v in sd.items():
        assert torch.equal(m.state_dict()[k].cpu(), v), k
    m.train()
    xs = torch.from_numpy(g['xs']).cuda()
    x1, x2, feat = m(xs)
    assert l2(x1, torch.from_numpy(g['x1'])) < 0.10 and l2(x2, torch.from_numpy(g['x2'])) < 0.10
    assert l2(feat[:, :32], torch.from_numpy(g['feat'])) < 0.10
    m.eval()
    with torch.no_grad():
        probs = m(xs)
    assert probs.shape == (2, 6, 64, 64)
    assert (probs.cpu() - torch.from_numpy(g['probs'])).abs().max() < 0.05


def test_shallow_topology_forward_backward():
    """Whole network, forward + backward + every parameter gradient, against the bf16-emulating oracle."""
    rt = 'resnet17t'
    m = build(rt)
    sd = omodel.init_state_dict(rt, 6, seed=4, head='aspp')
    m.load_state_dict(sd, strict=True)
    m.train()
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(4, 3, 128, 128, generator=gen)
    lab = torch.from_numpy(np.kron(np.random.default_rng(0).integers(-1, 6, size=(4, 8, 8)), np.ones((16, 16), np.int64)))
    names = omodel.param_names(sd)
    sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    r1, r2, rf = omodel.forward(sdr, xs, True, None, rt, {}, None, emulate_bf16=True)
    lref = opath.loss_calc([r1, r2], lab, -1)
    gref = dict(zip(names, torch.autograd.grad(lref, [sdr[k] for k in names])))
    from regda_amd.gast.balance import CrossEntropy
    from regda_amd.utils.tools import loss_calc
    m.zero_grad(set_to_none=True)
    x1, x2, feat = m(xs.cuda())
    loss = loss_calc([x1, x2], lab.cuda(), CrossEntropy(-1), multi=True)
    loss.backward()
    assert l2(x1, r1) < 0.03 and l2(x2, r2) < 0.03 and l2(feat, rf) < 0.03
    assert loss.item() == pytest.approx(lref.item(), rel=0.015)
    named = dict(m.named_parameters())
    a = torch.cat([named[k].grad.float().cpu().reshape(-1) for k in names])
    b = torch.cat([gref[k].reshape(-1) for k in names])
    assert (a @ b / (a.norm() * b.norm())).item() > 0.98 and a.norm().item() == pytest.approx(b.norm().item(), rel=0.04)
    for k in names:
        if 'conv2d_list' in k:
            assert l2(named[k].grad, gref[k]) < 0.05, k


def test_ssl_step_runs_with_aspp_heads():
    """The SSL step (teacher, label path, DDP bucket cuts, optimizer) is head-agnostic: one step on the ASPP model."""
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    m = build('resnet17t')
    m.load_state_dict(omodel.init_state_dict('resnet17t', 6, seed=3, head='aspp'), strict=True)
    b = make_batch(b=2, size=64, seed=7)
    st = SSLStep(m, torch.randn(6, 2048), ema_decay=0.99)
    p0 = m.flat_p.clone()
    ls, lt, gn = st.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], lr=1e-3)
    torch.cuda.synchronize()
    assert np.isfinite(ls.item()) and np.isfinite(lt.item()) and np.isfinite(gn.item()) and gn.item() > 0
    assert not torch.equal(p0, m.flat_p)
