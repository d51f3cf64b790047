"""GPU: stage-2 ("align") kernels through the C ABI -- PrototypeContrastiveLoss forward + feature gradient against the
golden vectors minted from regda/loss.py (tests/golden/pcl.npz) and, at the production shape, against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def run(ops, feat, protos, lab, temp, **kw):
    b, K, h, w = feat.shape
    df = torch.zeros(b * h * w, K, dtype=BF, device='cuda')
    loss = ops.pcl_loss(feat.cuda(), lab.cuda(), protos.cuda(), temperature=temp, ignore_label=-1, dfeat=df, **kw)
    g = df.float().reshape(b, h, w, K).permute(0, 3, 1, 2).cpu()
    return float(loss.item()), g


def test_pcl_loss_matches_the_reference_goldens(gold):
    from regda_amd import ops
    g = gold('pcl.npz')
    for i in range(3):
        feat, protos, lab = (torch.from_numpy(g[k + str(i)]) for k in ('feat', 'protos', 'lab'))
        loss, grad = run(ops, feat, protos, lab, float(g[f'temp{i}']))
        # fp32 arithmetic in another summation order; the gradient is stored in bf16 (relative 2^-9)
        np.testing.assert_allclose(loss, g[f'loss{i}'], rtol=2e-5)
        ref = torch.from_numpy(g[f'gfeat{i}'])
        assert (grad - ref).abs().max().item() <= 4e-3 * ref.abs().max().item() + 1e-8, i
        # ignored pixels get exactly zero gradient
        assert torch.equal(grad.permute(0, 2, 3, 1)[lab == -1], torch.zeros_like(grad.permute(0, 2, 3, 1)[lab == -1]))


def test_pcl_loss_production_shape_weight_and_accumulate():
    from regda_amd import ops
    from oracle import labelpath
    gen = torch.Generator().manual_seed(77)
    b, K, h, w, C = 4, 2048, 32, 32, 6
    feat = torch.randn(b, K, h, w, generator=gen)
    protos = torch.randn(C, K, generator=gen)
    lab = torch.randint(-1, C, (b, h, w), generator=gen)
    fr = feat.clone().requires_grad_(True)
    ref = labelpath.prototype_contrastive_loss(protos, fr, lab, temperature=8.0, ignore_label=-1)
    ref.backward()
    loss, grad = run(ops, feat, protos, lab, 8.0, weight=0.5)
    assert abs(loss - 0.5 * ref.item()) <= 2e-5 * abs(ref.item())
    rg = 0.5 * fr.grad
    assert ((grad - rg).norm() / rg.norm()).item() < 3e-3
    # accumulate onto an existing gradient; loss tensor accumulates too
    df = torch.ones(b * h * w, K, dtype=BF, device='cuda')
    acc = torch.full((1,), 2.0, device='cuda')
    ops.pcl_loss(feat.cuda(), lab.cuda(), protos.cuda(), 8.0, -1, 0.5, loss=acc, dfeat=df, accumulate=True)
    assert abs(acc.item() - 2.0 - 0.5 * ref.item()) <= 1e-4
    got = df.float().reshape(b, h, w, K).permute(0, 3, 1, 2).cpu() - 1.0
    assert ((got - rg).abs().max() < 1e-2)
    # nothing kept: NaN loss like nn.CrossEntropyLoss over zero elements, zero gradient
    df.zero_()
    none = ops.pcl_loss(feat.cuda(), torch.full((b, h, w), -1).cuda(), protos.cuda(), 8.0, -1, dfeat=df)
    assert torch.isnan(none).all() and float(df.float().abs().max()) == 0.0
    with pytest.raises(ValueError):
        ops.pcl_loss(feat.cuda(), lab.cuda(), torch.randn(5, K).cuda())


def test_align_step_matches_the_oracle_stage2_step():
    """regda_amd.align.AlignStep (stage 2, tools/train_align_reg.py:144-196) end to end against oracle.step.CpuAlignStep
    on the shallow topology: losses, labels, gradient norm, prototype update, SGD direction."""
    from oracle import model as omodel
    from oracle.step import CpuAlignStep
    from regda_amd.align import AlignStep
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=6)
    b = make_batch(b=4, size=128, seed=11, device='cpu')
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(1))
    ones = torch.ones(4, 512)
    cpu = CpuAlignStep(sd, protos, resnet_type=rt, lr=1e-3, proto_decay=0.999)
    ref = cpu.step(b['images_s'], b['label_s'], b['images_t'], b['regs_t'], (ones, ones), (ones, ones))
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True, cascade=False,
                       use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6,
                       is_ins_norm=True))
    m.load_state_dict(sd, strict=True)
    m.set_drop_masks(ones, ones)
    st = AlignStep(m, protos)
    g = {k: v.cuda() for k, v in b.items()}
    lseg, lal, gn = st.step(g['images_s'], g['label_s'], g['images_t'], g['regs_t'], 1e-3)
    # stated tolerances: bf16 network
    assert lseg.item() == pytest.approx(ref['loss_seg'], rel=0.02)
    assert lal.item() == pytest.approx(ref['loss_align'], rel=0.02)
    assert gn.sqrt().item() == pytest.approx(ref['grad_norm'], rel=0.06)
    # the source-side integer path is exact (same labels in), the target side may move a few borderline pixels
    assert torch.equal(st.last_label_s_down.cpu(), ref['label_s_down'])
    assert (st.last_hard.cpu() != ref['hard']).float().mean().item() < 0.03
    assert (st.last_label_t.cpu() != ref['label_t']).float().mean().item() < 0.05
    assert ((st.prototypes.cpu() - cpu.prototypes).norm() / cpu.prototypes.norm()).item() < 2e-3
    named = dict(m.named_parameters())
    for k, tol in (('encoder.resnet.layer4.1.conv3.weight', 0.97), ('encoder.resnet.conv1.weight', 0.9)):
        d_ref = cpu.sd[k].detach() - sd[k]
        d_got = named[k].detach().cpu() - sd[k]
        cos = (d_ref.flatten() @ d_got.flatten() / (d_ref.norm() * d_got.norm())).item()
        assert cos > tol, (k, cos)
    # the classifier sees only the source CE: its update must match closely
    k = 'layer5.conv_last.4.weight'
    d_ref, d_got = cpu.sd[k].detach() - sd[k], named[k].detach().cpu() - sd[k]
    assert ((d_got - d_ref).norm() / d_ref.norm()).item() < 0.08
