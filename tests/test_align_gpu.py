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
