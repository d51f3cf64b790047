"""GPU: every conv-stack kernel (through the C ABI) against a plain PyTorch fp32 reference of the same
op computed on the bf16-rounded operands.  Tolerances: outputs are bf16 (relative 2^-8 per rounding);
weight gradients are fp32 accumulations of bf16 products."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from regda_amd import ops
    return ops


def to_pxc(x):       # NCHW f32 -> [N*H*W, C] bf16 (cuda)
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(-1, c).to(BF).cuda().contiguous()


def from_pxc(t, n, h, w):
    return t.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2)


def rbf(x):          # round to bf16 and back
    return x.to(BF).float()


def relerr(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def _case_seed(case):
    # NOT hash(case): a tuple holding a str hashes differently in every interpreter (PYTHONHASHSEED), so a failure could
    # not be re-run on the data that produced it
    import zlib
    return zlib.crc32(repr(case).encode()) % 1000


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 64, 3, 1, 1, 1),
    (2, 16, 16, 128, 256, 1, 1, 0, 1),
    (1, 32, 32, 256, 128, 3, 1, 2, 2),
    (2, 17, 13, 64, 128, 3, 2, 1, 1),
    (2, 16, 16, 64, 72, 1, 2, 0, 1),
    (8, 32, 32, 256, 256, 3, 1, 1, 1),
    (1, 4, 4, 512, 512, 3, 1, 1, 1),
    (3, 1, 1, 2048, 512, 1, 1, 0, 1),
    (4, 32, 32, 1024, 256, 1, 1, 0, 1),    # layer-3 1x1 at a quarter of its size: several K splits
    # full-batch 32-wide 3x3 layers (layer3 / layer4 / head geometry at reduced channel counts), dilation 2, odd batch
    (16, 32, 32, 256, 256, 3, 1, 1, 1),
    (16, 32, 32, 128, 512, 3, 1, 1, 1),
    (8, 32, 32, 128, 512, 3, 1, 2, 2),
    (7, 16, 32, 192, 1024, 3, 1, 1, 1),
    # long-K 3x3 on 32-wide maps (layer 4 / the heads at reduced size): the halo kernel, dilation 1 and 2, forward and data
    # gradient; 5 tiles per image + ragged Cout (forward only)
    (16, 32, 32, 512, 512, 3, 1, 1, 1),
    (16, 32, 32, 512, 512, 3, 1, 2, 2),
    (8, 40, 32, 512, 1032, 3, 1, 2, 2),
    # the same long-K 3x3 shapes on maps WIDER than 32 (1024 x 1024 tiles at stride 16: 64 columns; 96 = three bands):
    # conv3x3_halo_wide_kernel, tiles of 8 / 4 image rows x one 32-column band, dilation 1 and 2, forward and data gradient
    (4, 32, 64, 512, 512, 3, 1, 1, 1),
    (4, 32, 64, 512, 512, 3, 1, 2, 2),
    (2, 40, 96, 512, 512, 3, 1, 1, 1),
    (4, 32, 64, 256, 256, 3, 1, 1, 1),
    # layer1 geometry (64 -> 64 on 128-wide maps): the weights-resident rolling-window kernel, 2 and 5 rows per workgroup
    (4, 128, 128, 64, 64, 3, 1, 1, 1),
    (10, 128, 128, 64, 64, 3, 1, 1, 1),
    (16, 128, 128, 64, 64, 3, 1, 1, 1),    # the step's batch: 8 rows per workgroup
    # tap-fused 3x3 weight-gradient geometries: 64-pixel K tiles of R rows x WT columns
    (2, 64, 64, 128, 256, 3, 1, 1, 1),     # WT=64, R=1
    (1, 8, 128, 256, 128, 3, 1, 1, 1),     # WT=64, two tiles per image row
    (1, 16, 64, 128, 256, 3, 1, 2, 2),     # WT=64, dilation 2
    (2, 8, 16, 256, 136, 3, 1, 2, 2),      # WT=16, R=4, dilation 2, ragged Cout
    (3, 32, 32, 256, 128, 3, 1, 1, 1),     # WT=32, R=2, odd image count (split-K tail)
]


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd_dgrad_wgrad(ops, case):
    N, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(_case_seed(case))
    x = rbf(torch.randn(N, Cin, H, W, generator=g))
    w = rbf(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    ref = F.conv2d(x, w, None, s, p, d)
    xg = to_pxc(x)
    wg = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin).to(BF).cuda().contiguous()
    y = torch.zeros(N * Ho * Wo, Cout, dtype=BF, device='cuda')
    stats = ops.new_stats(8, 2, Cout)
    ops.conv2d(xg, wg, y, N, H, W, Ho, Wo, k, k, s, p, d, 0, None, stats, 1)
    out = from_pxc(y, N, Ho, Wo)
    assert relerr(out, ref) < 1e-2, 'forward'
    # BatchNorm statistics fused in the epilogue (of the bf16-rounded outputs)
    yf = y.float()
    torch.testing.assert_close(ops.stats_value(stats).sum(0)[0].float().cpu(), yf.sum(0).cpu(), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(ops.stats_value(stats).sum(0)[1].float().cpu(), (yf * yf).sum(0).cpu(), rtol=1e-3, atol=1e-2)
    # data gradient = conv in mode 1 with [Cin][tap][Cout] weights (+ residual add in the epilogue)
    dy = rbf(torch.randn(N, Cout, Ho, Wo, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, s, p, d).backward(dy)
    dyg = to_pxc(dy)
    wt = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).to(BF).cuda().contiguous()
    res = rbf(torch.randn(N, Cin, H, W, generator=g))
    dx = torch.zeros(N * H * W, Cin, dtype=BF, device='cuda')
    if Cout % 64 == 0:
        ops.conv2d(dyg, wt, dx, N, Ho, Wo, H, W, k, k, s, p, d, 1, to_pxc(res), None)
        assert relerr(from_pxc(dx, N, H, W), xr.grad + res) < 1.5e-2, 'dgrad'
    # weight gradient (accumulating: run twice -> 2x)
    dw = torch.zeros(Cout, k * k, Cin, device='cuda')
    ops.conv2d_wgrad(xg, dyg, dw, N, H, W, Ho, Wo, k, k, s, p, d)
    ops.conv2d_wgrad(xg, dyg, dw, N, H, W, Ho, Wo, k, k, s, p, d)
    dwr = wr.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin) * 2
    assert relerr(dw.cpu(), dwr) < 2e-3, 'wgrad'


def test_wgrad_grouped_matches_reference_and_single_launches(ops):
    """rgda_conv2d_wgrad_grouped over a mixed list (every kernel family, 20 layers of one family so a group
    overflows into a second launch, one layer listed twice = accumulated twice) against F.conv2d's weight
    gradient, and against one rgda_conv2d_wgrad call per layer."""
    g = torch.Generator().manual_seed(21)
    layers = list(CASES) + [(2, 16, 16, 128, 256, 1, 1, 0, 1)] * 19
    items, refs, singles = [], [], []
    for (N, H, W, Cin, Cout, k, s, p, d) in layers:
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
        x = rbf(torch.randn(N, Cin, H, W, generator=g))
        dy = rbf(torch.randn(N, Cout, Ho, Wo, generator=g))
        wr = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        F.conv2d(x, wr, None, s, p, d).backward(dy)
        refs.append(wr.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin))
        xg, dyg = to_pxc(x), to_pxc(dy)
        dw = torch.zeros(Cout, k * k, Cin, device='cuda')
        items.append((xg, dyg, dw, N, H, W, Ho, Wo, k, k, s, p, d))
        one = torch.zeros(Cout, k * k, Cin, device='cuda')
        ops.conv2d_wgrad(xg, dyg, one, N, H, W, Ho, Wo, k, k, s, p, d)
        singles.append(one)
    ops.conv2d_wgrad_grouped(items + [items[0]])
    refs[0] = refs[0] * 2
    singles[0] = singles[0] * 2
    for i, (it, ref, one) in enumerate(zip(items, refs, singles)):
        assert relerr(it[2].cpu(), ref) < 2e-3, (i, layers[i])
        # same products, another split of the pixel sum: fp32 re-association only
        assert relerr(it[2].cpu(), one.cpu()) < 1e-5, (i, layers[i])
    ops.conv2d_wgrad_grouped([])
    with pytest.raises(ValueError):     # a bad entry anywhere: nothing is launched
        bad = list(items[1])
        bad[3] = 0
        before = items[2][2].clone()
        ops.conv2d_wgrad_grouped([items[2], tuple(bad)])
    torch.cuda.synchronize()
    assert torch.equal(before, items[2][2])


def test_wgrad_into_channel_slices_and_stacked_filters(ops):
    """rgda_wgrad_desc.lddw / co_split (ABI 5): a weight gradient written straight into a channel slice of a wider
    [Cout][taps][C] tensor, and a 1x1 layer whose output rows are T stacked filters of S channels landing channel-major
    ([s][t][ci]) in such a slice -- the head convolution's feature half and PPM branches.  Both against the dense call
    (bit for bit: same kernels, same summation order), untouched elements stay untouched, generic and tap-fused kernels."""
    g = torch.Generator().manual_seed(61)
    N, H, W = 2, 16, 32
    M = N * H * W
    for (Ci, Co, k) in [(128, 256, 3), (64, 64, 3), (128, 128, 1)]:
        x = rbf(torch.randn(M, Ci, generator=g)).to(BF).cuda()
        dy = rbf(torch.randn(M, Co, generator=g)).to(BF).cuda()
        dense = torch.zeros(Co, k * k, Ci, device='cuda')
        ops.conv2d_wgrad(x, dy, dense, N, H, W, H, W, k, k, 1, k // 2, 1)
        wide = torch.full((Co, k * k, Ci + 96), 7.0, device='cuda')
        view = wide[:, :, 32:32 + Ci]
        view.zero_()
        ops.conv2d_wgrad_grouped([(x, dy, view, N, H, W, H, W, k, k, 1, k // 2, 1)])
        assert torch.equal(view, dense), (Ci, Co, k)
        assert float(wide[:, :, :32].min()) == 7.0 and float(wide[:, :, 32 + Ci:].min()) == 7.0
    # stacked 1x1 filters: T = 3 filters of S = 64 channels, output row r = t * 64 + s -> wide[s][t][16 + ci]
    S, T, Ci = 64, 3, 128
    x = rbf(torch.randn(M, Ci, generator=g)).to(BF).cuda()
    dy = rbf(torch.randn(M, S * T, generator=g)).to(BF).cuda()
    dense = torch.zeros(S * T, 1, Ci, device='cuda')
    ops.conv2d_wgrad(x, dy, dense, N, H, W, H, W, 1, 1, 1, 0, 1)
    wide = torch.full((S, T, Ci + 48), 7.0, device='cuda')
    view = wide[:, :, 16:16 + Ci]
    view.zero_()
    ops.conv2d_wgrad_grouped([(x, dy, view, N, H, W, H, W, 1, 1, 1, 0, 1)])
    assert torch.equal(view, dense.view(T, S, Ci).permute(1, 0, 2))
    assert float(wide[:, :, :16].min()) == 7.0 and float(wide[:, :, 16 + Ci:].min()) == 7.0


def test_conv_strided_views_and_row_tail(ops):
    """ld > C on both sides (channel-slice views of a concat buffer) and M not a tile multiple."""
    g = torch.Generator().manual_seed(4)
    N, H, W, Cin, Cout = 1, 9, 7, 64, 64
    x = rbf(torch.randn(N, Cin, H, W, generator=g))
    w = rbf(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1)
    big = torch.zeros(N * H * W, 192, dtype=BF, device='cuda')
    big[:, 64:128] = to_pxc(x)
    outb = torch.full((N * H * W, 256), 7.0, dtype=BF, device='cuda')
    ops.conv2d(big[:, 64:128], w.reshape(Cout, 1, Cin).to(BF).cuda(), outb[:, 128:192], N, H, W, H, W, 1, 1, 1, 0, 1)
    assert relerr(from_pxc(outb[:, 128:192], N, H, W), F.conv2d(x, w)) < 1e-2
    assert (outb[:, :128] == 7).all() and (outb[:, 192:] == 7).all()


def test_stem_im2col_and_gemm(ops):
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 32, 48
    img = torch.randn(N, 3, H, W, generator=g)
    w = rbf(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    Ho, Wo = H // 2, W // 2
    col = torch.empty(N * Ho * Wo, 192, dtype=BF, device='cuda')
    ops.stem_im2col(img.cuda(), col, N, H, W, Ho, Wo)
    wp = torch.zeros(64, 192, dtype=BF, device='cuda')
    ops.pad_cast_bf16(w.permute(0, 2, 3, 1).reshape(64, 147).contiguous().cuda(), wp, 64, 147, 192)
    y = torch.empty(N * Ho * Wo, 64, dtype=BF, device='cuda')
    ops.conv2d(col, wp.reshape(64, 1, 192), y, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1)
    ref = F.conv2d(rbf(img), w, None, 2, 3)
    assert relerr(from_pxc(y, N, Ho, Wo), ref) < 1e-2


@pytest.mark.parametrize('N,H,W,groups', [(2, 40, 128, 1), (4, 128, 256, 2), (2, 18, 384, 2)])
def test_fused_stem_conv_equals_the_im2col_route(ops, N, H, W, groups):
    """rgda_stem_conv (the 7x7 / stride-2 stem straight from the image, Wo % 64 == 0) against (a) F.conv2d on the
    bf16-rounded operands and (b) the rgda_stem_im2col + rgda_conv2d route: the same bf16 products summed in another
    order inside the fp32 accumulator (outputs agree to one bf16 rounding, statistics to 1e-4); the inference-BatchNorm
    form against rgda_conv2d_bneval; odd heights / row blocks, per-group statistics; other widths are refused."""
    g = torch.Generator().manual_seed(31)
    img = torch.randn(N, 3, H, W, generator=g)
    w = rbf(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    M = N * Ho * Wo
    wp = torch.zeros(64, 192, dtype=BF, device='cuda')
    ops.pad_cast_bf16(w.permute(0, 2, 3, 1).reshape(64, 147).contiguous().cuda(), wp, 64, 147, 192)
    wb = wp.reshape(64, 1, 192)
    imgc = img.cuda()
    # the fused kernel takes one image batch per statistics group (the step has one tensor per domain)
    y = torch.empty(M, 64, dtype=BF, device='cuda')
    st = ops.new_stats(groups, 8, 2, 64)
    Ng, Mg = N // groups, M // groups
    for gi in range(groups):
        ops.stem_conv(imgc[gi * Ng:(gi + 1) * Ng].contiguous(), wb, y[gi * Mg:(gi + 1) * Mg], st[gi], Ng, H, W, Ho, Wo)
    ref = F.conv2d(rbf(img), w, None, 2, 3)
    assert relerr(from_pxc(y, N, Ho, Wo), ref) < 1e-2
    col = torch.empty(M, 192, dtype=BF, device='cuda')
    ops.stem_im2col(imgc, col, N, H, W, Ho, Wo)
    y2 = torch.empty(M, 64, dtype=BF, device='cuda')
    st2 = ops.new_stats(groups, 8, 2, 64)
    ops.conv2d(col, wb, y2, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1, 0, None, st2, groups)
    assert rel_l2(y, y2.float()) < 2e-3 and (y.float() - y2.float()).abs().max().item() <= 2.0 ** -7 * y2.float().abs().max().item()
    a, b = ops.stats_value(st).sum(1), ops.stats_value(st2).sum(1)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-3 * float(b.abs().max()))
    yg = y.float().view(groups, Mg, 64)
    torch.testing.assert_close(a[:, 0].float(), yg.sum(1), rtol=1e-4, atol=0.5)      # statistics of the stored (rounded) rows
    torch.testing.assert_close(a[:, 1].float(), (yg * yg).sum(1), rtol=1e-4, atol=0.5)
    # inference form
    rm, rv = torch.randn(64, generator=g).cuda() * 0.1, (torch.rand(64, generator=g) + 0.5).cuda()
    gam, bet = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda() * 0.1
    e1, e2 = torch.empty(M, 64, dtype=BF, device='cuda'), torch.empty(M, 64, dtype=BF, device='cuda')
    ops.stem_conv_bneval(imgc, wb, e1, rm, rv, gam, bet, True, N, H, W, Ho, Wo)
    ops.conv2d_bneval(col, wb, e2, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1, rm, rv, gam, bet, True)
    assert rel_l2(e1, e2.float()) < 3e-3 and float(e1.float().min()) >= 0.0
    with pytest.raises(ValueError):
        ops.stem_conv(imgc[:, :, :, :96].contiguous(), wb, y, None, N, H, 96, Ho, 48)


@pytest.mark.parametrize('N,H,W', [(2, 40, 128), (8, 128, 256), (3, 18, 384)])
def test_fused_stem_weight_gradient(ops, N, H, W):
    """rgda_stem_wgrad (the stem's weight gradient straight from the image, Wo % 64 == 0) against (a) autograd of F.conv2d on
    the bf16-rounded operands and (b) the rgda_stem_im2col + rgda_conv2d_wgrad route (the same bf16 products, another
    summation order); accumulating (two calls = twice), bit-identical from run to run; other widths are refused."""
    g = torch.Generator().manual_seed(33)
    img = torch.randn(N, 3, H, W, generator=g)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    M = N * Ho * Wo
    dy = rbf(torch.randn(N, 64, Ho, Wo, generator=g))
    imgc, dyg = img.cuda(), to_pxc(dy)
    dw = torch.zeros(64, 147, device='cuda')
    ops.stem_wgrad(imgc, dyg, dw, N, H, W, Ho, Wo)
    xr = rbf(img).requires_grad_(False)
    wr = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(xr, wr, None, 2, 3).backward(dy)
    ref = wr.grad.permute(0, 2, 3, 1).reshape(64, 147)
    assert relerr(dw.cpu(), ref) < 2e-3
    col = torch.empty(M, 192, dtype=BF, device='cuda')
    ops.stem_im2col(imgc, col, N, H, W, Ho, Wo)
    dwp = torch.zeros(64, 1, 192, device='cuda')
    ops.conv2d_wgrad(col, dyg, dwp, N, Ho, Wo, Ho, Wo, 1, 1, 1, 0, 1)
    assert relerr(dw.cpu(), dwp.view(64, 192)[:, :147].cpu()) < 1e-4
    first = dw.clone()
    ops.stem_wgrad(imgc, dyg, dw, N, H, W, Ho, Wo)
    torch.testing.assert_close(dw, 2 * first, rtol=0, atol=0)          # x + x is exact: the second pass summed identically
    # a strided gradient view (64 channels of a wider buffer)
    wide = torch.zeros(M, 96, dtype=BF, device='cuda')
    wide[:, 16:80] = dyg
    dw2 = torch.zeros(64, 147, device='cuda')
    ops.stem_wgrad(imgc, wide[:, 16:80], dw2, N, H, W, Ho, Wo)
    assert torch.equal(dw2, first)
    with pytest.raises(ValueError):
        ops.stem_wgrad(imgc[:, :, :, :96].contiguous(), dyg, dw, N, H, 96, Ho, 48)


@pytest.mark.parametrize('N,H,W', [(2, 32, 48), (1, 38, 270), (3, 16, 130)])
def test_stem_im2col_columns_exact(ops, N, H, W):
    """Every column vector, bit-exact: col[(n,ho,wo)][(kh*7+kw)*3 + c] = bf16(img[n, c, 2ho-3+kh, 2wo-3+kw]) or 0
    (padding, and the columns 147..191); widths that are not a multiple of the 64-pixel workgroup tile included."""
    g = torch.Generator().manual_seed(4)
    img = torch.randn(N, 3, H, W, generator=g)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    col = torch.full((N * Ho * Wo, 192), 7.0, dtype=BF, device='cuda')
    ops.stem_im2col(img.cuda(), col, N, H, W, Ho, Wo)
    unf = F.unfold(img.to(BF).float(), 7, 1, 3, 2)                     # (N, 3*49, Ho*Wo), row = c*49 + kh*7 + kw
    ref = unf.view(N, 3, 49, Ho * Wo).permute(0, 3, 2, 1).reshape(N * Ho * Wo, 147)
    got = col.float().cpu()
    assert torch.equal(got[:, :147], ref)
    assert float(got[:, 147:].abs().max()) == 0.0


@pytest.mark.parametrize('N,C,H,W', [(4, 64, 8, 8), (2, 256, 8, 8), (2, 2048, 4, 4), (3, 72, 5, 7), (2, 1024, 16, 16)])
def test_batchnorm_fwd_bwd(ops, N, C, H, W):
    """Standalone statistics / apply / backward-reduce / backward-apply kernels against torch autograd; the wide rows
    (C > 128) are split over several 128-channel workgroups, C = 72 leaves a partial channel vector block."""
    g = torch.Generator().manual_seed(2)
    M = N * H * W
    x = rbf(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    res = rbf(torch.randn(N, C, H, W, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = rm.clone(), rv.clone()
    yr = F.relu(F.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5) + res)
    go = rbf(torch.randn(N, C, H, W, generator=g))
    yr.backward(go)
    xg = to_pxc(x)
    stats = ops.new_stats(8, 2, C)
    ops.bn_stats(xg, stats, M, C)
    mi = torch.empty(2, C, device='cuda')
    rmg, rvg, nbt = rm.cuda(), rv.cuda(), torch.zeros((), dtype=torch.int64, device='cuda')
    ops.bn_finalize(stats, mi, rmg, rvg, nbt, M, C)
    torch.testing.assert_close(rmg.cpu(), rm2, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rvg.cpu(), rv2, rtol=1e-4, atol=1e-5)
    assert int(nbt) == 1
    y = torch.empty(M, C, dtype=BF, device='cuda')
    ops.bn_apply(xg, mi, gamma.cuda(), beta.cuda(), y, M, C, True, to_pxc(res))
    assert relerr(from_pxc(y, N, H, W), yr.detach()) < 1e-2
    sums = ops.new_stats(8, 2, C)
    gg = to_pxc(go)
    ops.bn_bwd_reduce(gg, y, xg, mi, sums, M, C, True)
    dx = torch.empty(M, C, dtype=BF, device='cuda')
    gm = torch.empty(M, C, dtype=BF, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_bwd_apply(gg, y, xg, mi, gamma.cuda(), sums, dx, M, C, True, gm, dgam, dbet)
    assert relerr(from_pxc(dx, N, H, W), xr.grad) < 2e-2
    assert relerr(dgam.cpu(), gr.grad) < 1e-2 and relerr(dbet.cpu(), br.grad) < 1e-2
    # gmask = gradient reaching the residual branch
    assert relerr(from_pxc(gm, N, H, W), go * (yr.detach() > 0)) < 1e-2
    # eval-mode finalize
    mi2 = torch.empty(2, C, device='cuda')
    ops.bn_finalize(None, mi2, rmg, rvg, None, M, C)
    torch.testing.assert_close(mi2[1].cpu(), 1 / torch.sqrt(rvg.cpu() + 1e-5), rtol=1e-5, atol=1e-6)


def unpack_mask(mask, C):
    """uint8 [M][C/8] sign bits -> bool [M][C]"""
    bits = (mask.cpu().unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1
    return bits.reshape(mask.shape[0], C).bool()


@pytest.mark.parametrize('C', [64, 256])
def test_relu_sign_mask_replaces_y_in_the_backward_kernels(ops, C):
    """bn_train_apply's relu_mask must be exactly [y > 0] bit for bit, and the three backward consumers
    (bn_bwd_reduce, bn_bwd_apply, conv2d_bnbwd's epilogue) must give IDENTICAL results from the mask as from y."""
    g = torch.Generator().manual_seed(31)
    N, H, W, G = 4, 8, 8, 2
    M = N * H * W
    x = torch.randn(M, C, generator=g).to(BF).cuda()
    res = torch.randn(M, C, generator=g).to(BF).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    ns = ((torch.rand(N, C, generator=g) > 0.2).float() / 0.8).cuda()
    stats = ops.new_stats(G, 8, 2, C)
    for gi in range(G):
        ops.bn_stats(x[gi * M // G:(gi + 1) * M // G], stats[gi], M // G, C)
    mi = torch.empty(G, 2, C, device='cuda')
    rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    y = torch.empty(M, C, dtype=BF, device='cuda')
    mask = torch.zeros(M, C // 8, dtype=torch.uint8, device='cuda')
    ops.bn_train_apply(x, stats, mi, rm, rv, nbt, gamma, beta, y, M, C, True, res, ns, H * W, groups=G, relu_mask=mask)
    assert torch.equal(unpack_mask(mask, C), (y.float() > 0).cpu())
    assert 0.2 < unpack_mask(mask, C).float().mean() < 0.8
    with pytest.raises(ValueError):      # a mask without ReLU is a caller bug
        ops.bn_train_apply(x, stats, mi, rm, rv, nbt, gamma, beta, y, M, C, False, res, ns, H * W, groups=G, relu_mask=mask)
    go = torch.randn(M, C, generator=g).to(BF).cuda()
    out = {}
    for tag, kw in (('y', dict()), ('mask', dict(relu_mask=mask))):
        yy = y if tag == 'y' else None
        sums = ops.new_stats(G, 8, 2, C)
        ops.bn_bwd_reduce(go, yy, x, mi, sums, M, C, True, ns, H * W, groups=G, **kw)
        dx = torch.empty(M, C, dtype=BF, device='cuda')
        gm = torch.empty(M, C, dtype=BF, device='cuda')
        dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        ops.bn_bwd_apply(go, yy, x, mi, gamma, sums, dx, M, C, True, gm, dgam, dbet, ns, H * W, groups=G, **kw)
        out[tag] = (ops.stats_value(sums, backward=True).sum(1).float(), dx, gm, dgam, dbet)
    for a, b in zip(out['y'][1:3], out['mask'][1:3]):
        assert torch.equal(a, b)
    for a, b in zip((out['y'][0],) + out['y'][3:], (out['mask'][0],) + out['mask'][3:]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)         # atomics: summation order only
    if C % 64 == 0:
        k, Cb = 3, 128
        dy = torch.randn(M, Cb, generator=g).to(BF).cuda()
        wt = (torch.randn(C, k * k, Cb, generator=g) * 0.05).to(BF).cuda()
        got = []
        for tag in ('y', 'mask'):
            dx = torch.empty(M, C, dtype=BF, device='cuda')
            sums = ops.new_stats(G, 8, 2, C)
            ops.conv2d_bnbwd(dy, wt, dx, N, H, W, H, W, k, k, 1, 1, 1, 1, res, sums, G, y if tag == 'y' else None, x, mi,
                             True, ns, H * W, relu_mask=mask if tag == 'mask' else None)
            got.append((dx, ops.stats_value(sums, backward=True).sum(1).float()))
        assert torch.equal(got[0][0], got[1][0])
        torch.testing.assert_close(got[0][1], got[1][1], rtol=1e-5, atol=1e-3)


def test_bn_dropout_scale(ops):
    g = torch.Generator().manual_seed(9)
    N, C, HW = 2, 64, 16
    M = N * HW
    x = rbf(torch.randn(M, C, generator=g))
    mi = torch.stack([torch.zeros(C), torch.ones(C)]).cuda()
    ns = (torch.rand(N, C, generator=g) > 0.3).float() / 0.9
    y = torch.empty(M, C, dtype=BF, device='cuda')
    one, zero = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_apply(x.to(BF).cuda(), mi, one, zero, y, M, C, True, None, ns.cuda(), HW)
    ref = F.relu(x).reshape(N, HW, C) * ns[:, None, :]
    assert relerr(y.float().cpu().reshape(N, HW, C), ref) < 1e-2


def test_maxpool(ops):
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 2, 64, 12, 10
    x = rbf(torch.randn(N, C, H, W, generator=g)).clamp(min=0)        # post-ReLU: many ties at 0
    x = (x * 4).round() / 4                                            # ... and ties at non-zero values
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    go = rbf(torch.randn_like(yr))
    yr.backward(go)
    Ho, Wo = yr.shape[-2:]
    y = torch.empty(N * Ho * Wo, C, dtype=BF, device='cuda')
    idx = torch.empty(N * Ho * Wo, C, dtype=torch.uint8, device='cuda')
    ops.maxpool_fwd(to_pxc(x), y, idx, N, H, W, C, Ho, Wo)
    assert torch.equal(from_pxc(y, N, Ho, Wo), yr.detach())
    gx = torch.empty(N * H * W, C, dtype=BF, device='cuda')
    ops.maxpool_bwd(to_pxc(go), idx, gx, N, H, W, C, Ho, Wo)
    assert relerr(from_pxc(gx, N, H, W), xr.grad) < 1e-2


def test_instnorm(ops):
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 128, 5, 7
    HW = H * W
    x = rbf(torch.randn(N, C, H, W, generator=g) * 3 + 1)
    xr = x.clone().requires_grad_(True)
    yr = F.instance_norm(xr, eps=1e-5)
    ga, gb = rbf(torch.randn(N, C, H, W, generator=g)), rbf(torch.randn(N, C, H, W, generator=g))
    gc = torch.randn(N, C, H, W, generator=g)
    yr.backward(ga + gb + rbf(gc))
    cat = torch.zeros(N * HW, 256, dtype=BF, device='cuda')
    feat = torch.empty(N, C, H, W, device='cuda')
    mi = torch.empty(N, 2, C, device='cuda')
    ops.instnorm_fwd(to_pxc(x), cat[:, :C], None, feat, mi, N, HW, C)
    torch.testing.assert_close(feat.cpu(), yr.detach(), rtol=1e-4, atol=1e-4)
    assert relerr(from_pxc(cat[:, :C], N, H, W), yr.detach()) < 1e-2
    dx = torch.empty(N * HW, C, dtype=BF, device='cuda')
    gc = rbf(gc)
    ops.instnorm_bwd(to_pxc(ga), to_pxc(gb), to_pxc(gc), to_pxc(x), mi, dx, N, HW, C)
    assert relerr(from_pxc(dx, N, H, W), xr.grad) < 2e-2


def test_spatial_mix_pool_and_upsample(ops):
    from regda_amd.models.Encoder import pool_matrix, upsample_matrix
    g = torch.Generator().manual_seed(6)
    N, C, H, W = 2, 64, 32, 32
    x = rbf(torch.randn(N, C, H, W, generator=g))
    xg = to_pxc(x)
    for s in (1, 2, 3, 6):
        P = pool_matrix(H, W, s)
        q = torch.empty(N * s * s, C, dtype=BF, device='cuda')
        ops.spatial_mix(xg, P.cuda(), q, N, s * s, H * W, C)
        refq = F.adaptive_avg_pool2d(x, s)
        assert relerr(from_pxc(q, N, s, s), refq) < 1e-2, s
        U = upsample_matrix(s, s, H, W)
        up = torch.empty(N * H * W, C, dtype=BF, device='cuda')
        ops.spatial_mix(q, U.cuda(), up, N, H * W, s * s, C)
        refu = F.interpolate(from_pxc(q, N, s, s), (H, W), mode='bilinear', align_corners=False)
        assert relerr(from_pxc(up, N, H, W), refu) < 1e-2, s


def test_spatial_mix_multi(ops):
    from regda_amd.models.Encoder import pool_matrix
    g = torch.Generator().manual_seed(16)
    N, C, H, W = 2, 64, 8, 8
    ins, mats, ref = [], [], 0
    for s in (1, 2, 3, 6):
        t = rbf(torch.randn(N, s * s, C, generator=g))
        Pt = pool_matrix(H, W, s).t().contiguous()
        ins.append(t.reshape(N * s * s, C).to(BF).cuda())
        mats.append(Pt.cuda())
        ref = ref + torch.einsum('ij,njc->nic', Pt, t)
    out = torch.empty(N * H * W, C, dtype=BF, device='cuda')
    ops.spatial_mix_multi(ins, mats, out, N, H * W, C)
    assert relerr(out.float().cpu().reshape(N, H * W, C), ref) < 1e-2


def test_classifier(ops):
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 2, 512, 4, 4
    hid = rbf(torch.randn(N, C, H, W, generator=g))
    w, b = torch.randn(6, C, generator=g) * 0.05, torch.randn(6, generator=g)
    hr, wr, br_ = hid.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = F.conv2d(hr, wr[:, :, None, None], br_)
    gl = torch.randn_like(out)
    out.backward(gl)
    hg = to_pxc(hid)
    logits = torch.empty(N, 6, H, W, device='cuda')
    ops.classifier_fwd(hg, w.cuda(), b.cuda(), logits, N, H * W, C, 6)
    torch.testing.assert_close(logits.cpu(), out.detach(), rtol=1e-4, atol=1e-4)
    dh = torch.empty(N * H * W, C, dtype=BF, device='cuda')
    dw, db = torch.zeros(6, C, device='cuda'), torch.zeros(6, device='cuda')
    ops.classifier_bwd(hg, w.cuda(), gl.cuda(), dh, dw, db, N, H * W, C, 6)
    assert relerr(from_pxc(dh, N, H, W), hr.grad) < 1e-2
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br_.grad, rtol=1e-4, atol=1e-4)


def test_optimizer_kernels(ops):
    g = torch.Generator().manual_seed(8)
    n = 4096 * 3 + 8
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    v = torch.zeros(n)
    sh = p.clone()
    pd, gd, vd, sd = p.cuda(), gr.cuda(), v.cuda(), sh.cuda()
    pb = torch.empty(n, dtype=BF, device='cuda')
    sb = torch.empty(n, dtype=BF, device='cuda')
    gn, ws = torch.empty(1, device='cuda'), torch.empty(1024, device='cuda')
    lr = torch.tensor([0.01], device='cuda')
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=5e-4)
    for step in range(2):
        ops.sumsq(gd, gn, ws)
        assert gn.item() == pytest.approx((gr.double() ** 2).sum().item(), rel=1e-5)
        ops.sgd_step(pd, gd, vd, sd, pb, gn, lr, 0.9, 5e-4, 32.0, 1.0, 0.99, step == 0, shadow_bf16=sb)
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], 32.0)
        opt.step()
        sh = 0.01 * pr.detach() + 0.99 * sh
    torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sd.cpu(), sh, rtol=1e-5, atol=1e-6)
    assert torch.equal(pb.cpu(), pd.cpu().to(BF))
    assert torch.equal(sb.cpu(), sd.cpu().to(BF))          # the EMA teacher's mirror, written in the same pass
    w = torch.randn(40, 9, 24, generator=g)
    wt = torch.empty(24, 9, 40, dtype=BF, device='cuda')
    ops.weight_transpose_bf16(w.cuda(), wt, 40, 9, 24)
    assert torch.equal(wt.cpu(), w.permute(2, 1, 0).to(BF))


@pytest.mark.parametrize('C,N,HW', [(512, 4, 64), (64, 2, 16)])
def test_batchnorm_bwd_with_dropout_scale(ops, C, N, HW):
    """conv_last BN + ReLU + Dropout2d (per (image, channel) scale) backward vs autograd."""
    g = torch.Generator().manual_seed(12)
    M = N * HW
    x = rbf(torch.randn(M, C, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ns = (torch.rand(N, C, generator=g) > 0.1).float() / 0.9
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    xn = xr.reshape(N, HW, C).permute(0, 2, 1)                    # (N, C, HW)
    yr = F.relu(F.batch_norm(xn, None, None, gr, br, True, 0.1, 1e-5)) * ns[:, :, None]
    go = rbf(torch.randn(N, C, HW, generator=g))
    yr.backward(go)
    xg = x.to(BF).cuda()
    stats = ops.new_stats(8, 2, C)
    ops.bn_stats(xg, stats, M, C)
    mi = torch.empty(2, C, device='cuda')
    ops.bn_finalize(stats, mi, None, None, None, M, C)
    y = torch.empty(M, C, dtype=BF, device='cuda')
    ops.bn_apply(xg, mi, gamma.cuda(), beta.cuda(), y, M, C, True, None, ns.cuda(), HW)
    ref_y = yr.detach().permute(0, 2, 1).reshape(M, C)
    assert relerr(y.float().cpu(), ref_y) < 1e-2
    gg = go.permute(0, 2, 1).reshape(M, C).to(BF).cuda().contiguous()
    sums = ops.new_stats(8, 2, C)
    ops.bn_bwd_reduce(gg, y, xg, mi, sums, M, C, True, ns.cuda(), HW)
    dx = torch.empty(M, C, dtype=BF, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_bwd_apply(gg, y, xg, mi, gamma.cuda(), sums, dx, M, C, True, None, dgam, dbet, ns.cuda(), HW)
    assert relerr(dx.float().cpu(), xr.grad) < 2e-2
    assert relerr(dgam.cpu(), gr.grad) < 1e-2 and relerr(dbet.cpu(), br.grad) < 1e-2


@pytest.mark.parametrize('groups', [1, 2])
def test_conv_dgrad_with_fused_bn_backward_reduction(ops, groups):
    """rgda_conv2d_bnbwd: the data-gradient conv also accumulates the consumer BatchNorm's backward sums; they
    must equal rgda_bn_bwd_reduce run on the stored gradient."""
    g = torch.Generator().manual_seed(21)
    N, H, W, Cf, Cb, k = 4, 16, 16, 128, 256, 3            # forward conv Cf -> Cb; its data gradient Cb -> Cf
    M = N * H * W
    dy = torch.randn(M, Cb, generator=g).to(BF).cuda()
    wt = (torch.randn(Cf, k * k, Cb, generator=g) * 0.05).to(BF).cuda()
    res = torch.randn(M, Cf, generator=g).to(BF).cuda()
    cy = torch.randn(M, Cf, generator=g).to(BF).cuda()      # consumer's activation (ReLU mask) and raw conv output
    cx = torch.randn(M, Cf, generator=g).to(BF).cuda()
    mi = torch.stack([torch.randn(groups, Cf, generator=g) * 0.1, torch.rand(groups, Cf, generator=g) + 0.5], 1).cuda().contiguous()
    ns = ((torch.rand(N, Cf, generator=g) > 0.2).float() / 0.8).cuda()
    dx = torch.empty(M, Cf, dtype=BF, device='cuda')
    sums = ops.new_stats(groups, 8, 2, Cf)
    ops.conv2d_bnbwd(dy, wt, dx, N, H, W, H, W, k, k, 1, 1, 1, 1, res, sums, groups, cy, cx, mi, True, ns, H * W)
    dx2 = torch.empty_like(dx)
    ops.conv2d(dy, wt, dx2, N, H, W, H, W, k, k, 1, 1, 1, 1, res, None)
    assert torch.equal(dx, dx2)
    ref = ops.new_stats(groups, 8, 2, Cf)
    ops.bn_bwd_reduce(dx, cy, cx, mi, ref, M, Cf, True, ns, H * W, groups=groups)
    torch.testing.assert_close(ops.stats_value(sums, backward=True).sum(1).float().cpu(),
                               ops.stats_value(ref, backward=True).sum(1).float().cpu(), rtol=2e-4, atol=2e-2)


def test_weight_layout_table_modes(ops):
    """rgda_weight_transpose_batched: transposed / tap-stacked / plain bf16 copies of channel slices of a wider fp32
    weight tensor, several rows in one launch."""
    g = torch.Generator().manual_seed(41)
    Co, T, Cw = 96, 9, 320
    w = torch.randn(Co, T, Cw, generator=g).cuda()
    rows, blk, outs = [], 0, []
    for (off, Ci, mode) in ((0, 128, 2), (128, 64, 1), (192, 128, 0), (0, 320, 0)):
        shape = {0: (Ci, T, Co), 1: (T, Co, Ci), 2: (Co, T, Ci)}[mode]
        dst = torch.zeros(shape, dtype=BF, device='cuda')
        rows.append([w.data_ptr() + 4 * off, dst.data_ptr(), Co, T, Ci, blk, Cw, mode])
        blk += -(-Ci // 64) * -(-Co // 64) * T        # RGDA_LAYOUT_TILE
        sl = w[:, :, off:off + Ci]
        ref = {0: sl.permute(2, 1, 0), 1: sl.permute(1, 0, 2), 2: sl}[mode]
        outs.append((dst, ref.to(BF)))
    table = torch.tensor(rows, dtype=torch.int64, device='cuda')
    ops.weight_transpose_batched(table, len(rows), blk)
    for dst, ref in outs:
        assert torch.equal(dst, ref.contiguous())


def test_conv_residual_gated_by_relu_mask(ops):
    """res_relu_mask: the residual enters the epilogue only where its sign bit is set (= adding a pre-gated copy)."""
    g = torch.Generator().manual_seed(51)
    N, H, W, Ci, Co = 2, 8, 8, 64, 128
    M = N * H * W
    x = torch.randn(M, Ci, generator=g).to(BF).cuda()
    w = (torch.randn(Co, 1, Ci, generator=g) * 0.1).to(BF).cuda()
    res = torch.randn(M, Co, generator=g).to(BF).cuda()
    keep = torch.rand(M, Co, generator=g) > 0.4
    mask = (keep.reshape(M, Co // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).cuda()
    gated = torch.where(keep.cuda(), res, torch.zeros_like(res))
    y1 = torch.empty(M, Co, dtype=BF, device='cuda')
    y2 = torch.empty(M, Co, dtype=BF, device='cuda')
    ops.conv2d(x, w, y1, N, H, W, H, W, 1, 1, 1, 0, 1, 0, res, None, res_mask=mask)
    ops.conv2d(x, w, y2, N, H, W, H, W, 1, 1, 1, 0, 1, 0, gated, None)
    assert torch.equal(y1, y2)
    with pytest.raises(ValueError):
        ops.conv2d(x, w, y1, N, H, W, H, W, 1, 1, 1, 0, 1, 0, None, None, res_mask=mask)


# ---------------------------------------------------------------- factored spatial maps (csrc/mix_kernels.hip)
@pytest.mark.parametrize('in_f32,out_f32', [(False, True), (True, False), (False, False)])
def test_group_mix_matches_dense(in_f32, out_f32):
    from regda_amd import ops
    g = torch.Generator().manual_seed(5)
    G, I, J, C = 37, 36, 32, 512
    W = torch.randn(I, J, generator=g)
    W[torch.rand(I, J, generator=g) < 0.4] = 0
    x = torch.randn(G * J, C, generator=g)
    xin = (x if in_f32 else x.to(BF)).cuda()
    out = torch.empty(G * I, C, dtype=torch.float32 if out_f32 else BF, device='cuda')
    ops.group_mix(xin, W.cuda(), out, G, I, J, C)
    ref = torch.einsum('ij,gjc->gic', W, xin.float().cpu().view(G, J, C)).reshape(G * I, C)
    tol = 1e-5 if out_f32 else 1e-2
    assert ((out.float().cpu() - ref).norm() / ref.norm()).item() < tol
    # a narrower channel count and odd I (the last output row has no partner)
    G, I, J, C = 5, 7, 9, 72
    W = torch.randn(I, J, generator=g)
    x = torch.randn(G * J, C, generator=g).to(BF).cuda()
    out = torch.empty(G * I, C, device='cuda')
    ops.group_mix(x, W.cuda(), out, G, I, J, C)
    ref = torch.einsum('ij,gjc->gic', W, x.float().cpu().view(G, J, C)).reshape(G * I, C)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_sparse_mix_matches_dense_multi_source():
    from regda_amd import ops
    from regda_amd.models.Encoder import _csr
    g = torch.Generator().manual_seed(6)
    N, I, C = 3, 50, 512
    Js = (9, 36, 81, 324)
    mats = []
    for J in Js:
        m = torch.randn(I, J, generator=g)
        m[torch.rand(I, J, generator=g) < 0.9] = 0
        mats.append(m)
    mats[0][7] = 0; mats[1][7] = 0; mats[2][7] = 0; mats[3][7] = 0          # an empty row
    ins = [torch.randn(N * J, C, generator=g).to(BF) for J in Js]
    csr = tuple(t.cuda() for t in _csr(mats))
    out = torch.empty(N * I, C, device='cuda')
    ops.sparse_mix([t.cuda() for t in ins], csr, out, N, C)
    ref = sum(torch.einsum('ij,njc->nic', m, t.float().view(N, -1, C)) for m, t in zip(mats, ins)).reshape(N * I, C)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    assert float(out.view(N, I, C)[:, 7].abs().max()) == 0.0
    # f32 in -> bf16 out, single source
    a = torch.randn(N * 81, C, generator=g)
    csr1 = tuple(t.cuda() for t in _csr([mats[2]]))
    o2 = torch.empty(N * I, C, dtype=BF, device='cuda')
    ops.sparse_mix([a.cuda()], csr1, o2, N, C)
    # the same rows split over three output tensors (20 + 1 + 29 rows per image)
    parts = [torch.empty(N * r, C, dtype=BF, device='cuda') for r in (20, 1, 29)]
    ops.sparse_mix([a.cuda()], csr1, parts, N, C)
    whole = o2.view(N, I, C)
    for t, (lo, hi) in zip(parts, ((0, 20), (20, 21), (21, 50))):
        assert torch.equal(t.view(N, hi - lo, C), whole[:, lo:hi])
    ref2 = torch.einsum('ij,njc->nic', mats[2], a.view(N, 81, C)).reshape(N * I, C)
    assert ((o2.float().cpu() - ref2).norm() / ref2.norm()).item() < 1e-2


def test_factored_ppm_maps_equal_the_one_pass_maps_per_op():
    """Same Z / dc in, both forms of V and V^T (models/Encoder.py:_head_last_fwd/_bwd): they differ only by the bf16
    rounding of the outputs (fp32 accumulation in a different order)."""
    from regda_amd import ops
    from regda_amd.models.Encoder import POOL_SCALES, _csr, ppm_factored_maps, ppm_tap_matrix
    g = torch.Generator().manual_seed(9)
    N, h, w, C = 3, 32, 32, 512
    Wx, Ay = ppm_factored_maps(h, w)
    R = Wx.shape[0]
    fwd = tuple(t.cuda() for t in _csr(Ay))
    zs = [torch.randn(N * s * s * 9, C, generator=g).to(BF).cuda() for s in POOL_SCALES]
    Vs = [ppm_tap_matrix(h, w, s).cuda().contiguous() for s in POOL_SCALES]
    one = torch.empty(N * h * w, C, dtype=BF, device='cuda')
    ops.spatial_mix_multi(zs, Vs, one, N, h * w, C)
    rows = torch.empty(N * h * R, C, device='cuda')
    ops.sparse_mix(zs, fwd, rows, N, C)
    two = torch.empty(N * h * w, C, dtype=BF, device='cuda')
    ops.group_mix(rows, Wx.t().contiguous().cuda(), two, N * h, w, R, C)
    ref = sum(torch.einsum('pj,njc->npc', V.cpu(), z.float().cpu().view(N, -1, C)) for V, z in zip(Vs, zs)).reshape(N * h * w, C)
    assert ((two.float().cpu() - ref).norm() / ref.norm()).item() < 4e-3
    assert ((two.float() - one.float()).norm() / one.float().norm()).item() < 5e-3
    dc = torch.randn(N * h * w, C, generator=g).to(BF).cuda()
    ops.group_mix(dc, Wx.cuda().contiguous(), rows, N * h, R, w, C)
    dzs = [torch.empty(N * 9 * s * s, C, dtype=BF, device='cuda') for s in POOL_SCALES]
    ops.sparse_mix([rows], tuple(t.cuda() for t in _csr([torch.cat([A.t() for A in Ay], 0).contiguous()])), dzs, N, C)
    for s, dz2, V in zip(POOL_SCALES, dzs, Vs):
        dz1 = torch.empty(N * 9 * s * s, C, dtype=BF, device='cuda')
        ops.spatial_mix(dc, V.t().contiguous(), dz1, N, 9 * s * s, h * w, C)
        ref = torch.einsum('pj,npc->njc', V.cpu(), dc.float().cpu().view(N, h * w, C)).reshape(N * 9 * s * s, C)
        assert ((dz2.float().cpu() - ref).norm() / ref.norm()).item() < 4e-3, s
        assert ((dz2.float() - dz1.float()).norm() / dz1.float().norm()).item() < 5e-3, s


# ---------------------------------------------------------------- the 1x1 layers of layer1 / layer2 at full size
# thousands of tiles per launch, several statistics groups per workgroup wave: every fused epilogue is checked at such a
# size against torch on the same bf16 operands.  64 / 128 input channels with >= 1024 tiles go through
# conv1x1_stream_kernel (persistent workgroups, T consecutive pixel tiles each, one instantiation per fused epilogue);
# the others through conv_igemm_kernel.
STREAM = [  # N, H, W, Cin, Cout
    (4, 128, 128, 64, 256),        # layer1 conv3: stream kernel, T = 4 tiles per workgroup
    (16, 128, 128, 64, 256),       # ... at the step's 8 + 8 batch: T = 16
    (6, 128, 128, 64, 256),        # T = 6 (not a power of two), 384 tiles per statistics group
    (8, 128, 128, 256, 64),        # layer1 conv1: BC = 64, four K slabs (conv_igemm_kernel)
    (4, 64, 64, 128, 1024),        # stream kernel with two K sub-tiles, eight channel tiles, T = 4
    (6, 128, 128, 64, 128),        # tiles_p = 768, one channel tile: three tiles per workgroup would be too few -> conv_igemm_kernel
    (16, 32, 32, 256, 1024),       # layer3 conv3 / conv1's data gradient: four K tiles (conv_igemm_kernel)
    (16, 64, 64, 128, 512),        # layer2 conv3 / conv1's data gradient: stream kernel, T = 8
    (8, 32, 32, 256, 1024),        # the teacher's batch
]


@pytest.mark.parametrize('N,H,W,Ci,Co', STREAM)
def test_large_1x1_conv_plain_residual_and_statistics(ops, N, H, W, Ci, Co):
    g = torch.Generator().manual_seed(Ci + Co)
    M = N * H * W
    x = torch.randn(M, Ci, generator=g).to(BF).cuda()
    w = (torch.randn(Co, 1, Ci, generator=g) * 0.1).to(BF).cuda()
    res = torch.randn(M, Co, generator=g).to(BF).cuda()
    ref = x.float() @ w.float().view(Co, Ci).t()
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    ops.conv2d(x, w, y, N, H, W, H, W, 1, 1, 1, 0, 1, 0)
    assert torch.equal(y, ref.to(BF)) or relerr(y.float().cpu(), ref.cpu()) < 4e-3
    # statistics of the STORED values, per group (2 groups = source / target batch), summed over the replicas
    groups = 2
    st = ops.new_stats(groups, 8, 2, Co)
    y2 = torch.empty_like(y)
    ops.conv2d(x, w, y2, N, H, W, H, W, 1, 1, 1, 0, 1, 0, stats=st, stat_groups=groups)
    assert torch.equal(y2, y)
    yg = y.float().view(groups, M // groups, Co)
    torch.testing.assert_close(ops.stats_value(st).sum(1)[:, 0].float().cpu(), yg.sum(1).cpu(), rtol=1e-4, atol=0.5)
    torch.testing.assert_close(ops.stats_value(st).sum(1)[:, 1].float().cpu(), (yg * yg).sum(1).cpu(), rtol=1e-4, atol=0.5)
    # residual added before the store
    y3 = torch.empty_like(y)
    ops.conv2d(x, w, y3, N, H, W, H, W, 1, 1, 1, 0, 1, 0, res, None)
    assert relerr(y3.float().cpu(), (ref + res.float()).cpu()) < 8e-3      # the conv is rounded to bf16 before the add
    # residual + statistics of the stored sum
    st3 = ops.new_stats(groups, 8, 2, Co)
    y4 = torch.empty_like(y)
    ops.conv2d(x, w, y4, N, H, W, H, W, 1, 1, 1, 0, 1, 0, res, stats=st3, stat_groups=groups)
    assert torch.equal(y4, y3)
    y3g = y3.float().view(groups, M // groups, Co)
    torch.testing.assert_close(ops.stats_value(st3).sum(1)[:, 0].float().cpu(), y3g.sum(1).cpu(), rtol=1e-4, atol=0.5)
    torch.testing.assert_close(ops.stats_value(st3).sum(1)[:, 1].float().cpu(), (y3g * y3g).sum(1).cpu(), rtol=1e-4, atol=0.5)
    # the same call twice: bit-identical output and accumulators (order-independent statistics)
    st4 = ops.new_stats(groups, 8, 2, Co)
    y5 = torch.empty_like(y)
    ops.conv2d(x, w, y5, N, H, W, H, W, 1, 1, 1, 0, 1, 0, res, stats=st4, stat_groups=groups)
    assert torch.equal(y5, y4) and torch.equal(ops.stats_value(st4).sum(1), ops.stats_value(st3).sum(1))
    # strided views on both sides
    xs = torch.randn(M, Ci + 64, generator=g).to(BF).cuda()
    ybig = torch.zeros(M, Co + 8, dtype=BF, device='cuda')
    ops.conv2d(xs[:, 64:], w, ybig[:, :Co], N, H, W, H, W, 1, 1, 1, 0, 1, 0)
    assert relerr(ybig[:, :Co].float().cpu(), (xs[:, 64:].float() @ w.float().view(Co, Ci).t()).cpu()) < 4e-3
    assert float(ybig[:, Co:].abs().max()) == 0.0


@pytest.mark.parametrize('N,H,W,Cb,Cf,use_mask', [(4, 128, 128, 64, 256, False), (16, 32, 32, 256, 1024, False),
                                                  (16, 32, 32, 256, 1024, True), (16, 64, 64, 128, 512, True)])
def test_large_1x1_conv_fused_bn_backward_sums_and_masks(ops, N, H, W, Cb, Cf, use_mask):
    """The data-gradient form at layer1 / layer3 / layer2 size (64 / 128 input channels: conv1x1_stream_kernel): residual gated by
    a sign mask + the consumer BatchNorm's backward sums (ReLU from y or from its sign mask), against the unfused kernels
    (rgda_bn_bwd_reduce on the stored gradient)."""
    g = torch.Generator().manual_seed(77)
    groups = 2                                                 # forward conv Cf -> Cb (1x1); its data gradient Cb -> Cf
    M = N * H * W
    dy = torch.randn(M, Cb, generator=g).to(BF).cuda()
    wt = (torch.randn(Cf, 1, Cb, generator=g) * 0.1).to(BF).cuda()
    res = torch.randn(M, Cf, generator=g).to(BF).cuda()
    keep = torch.rand(M, Cf, generator=g) > 0.4
    rmask = (keep.reshape(M, Cf // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).cuda()
    cy = torch.randn(M, Cf, generator=g).to(BF).cuda()
    cx = torch.randn(M, Cf, generator=g).to(BF).cuda()
    mi = torch.stack([torch.randn(groups, Cf, generator=g) * 0.1, torch.rand(groups, Cf, generator=g) + 0.5], 1).cuda().contiguous()
    dx = torch.empty(M, Cf, dtype=BF, device='cuda')
    sums = ops.new_stats(groups, 8, 2, Cf)
    ymask = ((cy > 0).reshape(M, Cf // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8, device='cuda')).sum(-1).to(torch.uint8)
    ops.conv2d_bnbwd(dy, wt, dx, N, H, W, H, W, 1, 1, 1, 0, 1, 1, res, sums, groups, None if use_mask else cy, cx, mi, True,
                     res_mask=rmask, relu_mask=ymask if use_mask else None)
    gated = torch.where(keep.cuda(), res, torch.zeros_like(res))
    ref = dy.float() @ wt.float().view(Cf, Cb).t() + gated.float()
    assert relerr(dx.float().cpu(), ref.cpu()) < 8e-3
    want = ops.new_stats(groups, 8, 2, Cf)
    ops.bn_bwd_reduce(dx, cy, cx, mi, want, M, Cf, True, groups=groups)
    torch.testing.assert_close(ops.stats_value(sums, backward=True).sum(1).float().cpu(),
                               ops.stats_value(want, backward=True).sum(1).float().cpu(), rtol=2e-4, atol=0.5)
    # the ReLU sign recomputed from the raw convolution output (relu = 2: the consumer's activation was never written)
    gamma, beta = (torch.rand(Cf, generator=g) + 0.5).cuda(), (torch.randn(Cf, generator=g) * 0.3).cuda()
    dx2 = torch.empty_like(dx)
    sums2 = ops.new_stats(groups, 8, 2, Cf)
    ops.conv2d_bnbwd(dy, wt, dx2, N, H, W, H, W, 1, 1, 1, 0, 1, 1, res, sums2, groups, None, cx, mi, 2, res_mask=rmask,
                     bn_gamma=gamma, bn_beta=beta)
    assert torch.equal(dx2, dx)                                # the stored gradient does not depend on the ReLU source
    want2 = ops.new_stats(groups, 8, 2, Cf)
    ops.bn_bwd_reduce(dx, None, cx, mi, want2, M, Cf, 2, groups=groups, gamma=gamma, beta=beta)
    torch.testing.assert_close(ops.stats_value(sums2, backward=True).sum(1).float().cpu(),
                               ops.stats_value(want2, backward=True).sum(1).float().cpu(), rtol=2e-4, atol=0.5)


@pytest.mark.parametrize('N,H,W,Ci,Co', [(4, 128, 128, 64, 256), (8, 32, 32, 256, 1024), (8, 64, 64, 128, 512)])
def test_large_1x1_conv_inference_batchnorm(ops, N, H, W, Ci, Co):
    """The EMA teacher's unit at layer1 / layer3 / layer2 size: conv + eval-mode BN + residual + ReLU in one kernel."""
    g = torch.Generator().manual_seed(78)
    M = N * H * W
    x = torch.randn(M, Ci, generator=g).to(BF).cuda()
    w = (torch.randn(Co, 1, Ci, generator=g) * 0.1).to(BF).cuda()
    res = torch.randn(M, Co, generator=g).to(BF).cuda()
    rm, rv = torch.randn(Co, generator=g).cuda() * 0.1, (torch.rand(Co, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda() * 0.1
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    ops.conv2d_bneval(x, w, y, N, H, W, H, W, 1, 1, 1, 0, 1, rm, rv, gamma, beta, True, res)
    c = (x.float() @ w.float().view(Co, Ci).t()).to(BF).float()         # the kernel normalises the bf16-rounded conv
    ref = torch.relu((c - rm) / torch.sqrt(rv + 1e-5) * gamma + beta + res.float())
    assert relerr(y.float().cpu(), ref.cpu()) < 1e-2


# ---------------------------------------------------------------- the production 3x3 kernels of the 32 x 32 maps
# conv3x3_halo_kernel (csrc/conv_kernels.hip: conv_use_halo) serves the 3x3 / stride-1 convolutions of the 32-wide maps from
# 100 tiles on: <D, 8> (128 x 256 tiles) for K >= 4096 -- the head's 3x3 2048 -> 512 and layer4's 512 -> 512 (dilation 1 / 2) --
# and <1, 4> (128 x 128 tiles) for layer3's 256 -> 256, at the full 8 + 8 batch and at the teacher's 8 images (half the tiles:
# half the CUs).  Every epilogue the step runs on them is checked here at exactly those geometries, forward and data gradient,
# against fp32 torch on the same bf16 operands (im2col + matmul on the GPU: 16384 x 18432 x 512 is too slow for the CPU suite).
BIG = [  # N, H, W, Cin, Cout, k, pad, dil          what it is in the step
    (16, 32, 32, 2048, 512, 3, 1, 1),              # head conv, feature half: forward (+ PPM residual, statistics)
    (16, 32, 32, 512, 512, 3, 2, 2),               # layer4.{1,2}.conv2: forward, atrous
    (16, 32, 32, 512, 512, 3, 1, 1),               # layer4.0.conv2
    (16, 32, 32, 256, 256, 3, 1, 1),               # layer3.*.conv2: four image rows per tile
    (8, 32, 32, 2048, 512, 3, 1, 1),               # the teacher's head conv: 128 tiles of eight rows
    (8, 32, 32, 256, 256, 3, 1, 1),                # the teacher's layer3 conv2: 128 tiles of four rows
]


def _ref_conv_gpu(x_pxc, w, N, H, W, k, pad, dil):
    """fp32 reference of a stride-1 'same' conv on pixel-major bf16 operands: unfold + matmul in fp32 on the GPU.
    x_pxc [N*H*W, Cin] bf16, w [Cout, k*k, Cin] bf16 -> [N*H*W, Cout] f32."""
    Cin, Cout = x_pxc.shape[1], w.shape[0]
    x = x_pxc.float().view(N, H, W, Cin).permute(0, 3, 1, 2)
    cols = F.unfold(x, k, dil, pad, 1)                                   # (N, Cin*k*k, H*W), row = ci*k*k + tap
    cols = cols.view(N, Cin, k * k, H * W).permute(0, 3, 2, 1).reshape(N * H * W, k * k * Cin)
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        return cols @ w.float().view(Cout, k * k * Cin).t()
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize('N,H,W,Ci,Co,k,pad,dil', BIG)
def test_big_tile_forward_statistics_residual(ops, N, H, W, Ci, Co, k, pad, dil):
    M, groups = N * H * W, 2
    g = torch.Generator().manual_seed(Ci + dil)
    x = torch.randn(M, Ci, generator=g).to(BF).cuda()
    w = (torch.randn(Co, k * k, Ci, generator=g) * (2.0 / (Ci * k * k)) ** 0.5).to(BF).cuda()
    res = torch.randn(M, Co, generator=g).to(BF).cuda()
    ref = _ref_conv_gpu(x, w, N, H, W, k, pad, dil)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, pad, dil, 0)
    assert relerr(y.float(), ref) < 6e-3 and rel_l2(y, ref) < 3e-3          # one bf16 rounding of the output
    # residual added before the store + per-group statistics of the STORED values (the head conv's epilogue)
    st = ops.new_stats(groups, 8, 2, Co)
    y2 = torch.empty_like(y)
    ops.conv2d(x, w, y2, N, H, W, H, W, k, k, 1, pad, dil, 0, res, st, groups)
    assert rel_l2(y2, ref + res.float()) < 4e-3
    yg = y2.float().view(groups, M // groups, Co)
    torch.testing.assert_close(ops.stats_value(st).sum(1)[:, 0].float(), yg.sum(1), rtol=1e-4, atol=0.5)
    torch.testing.assert_close(ops.stats_value(st).sum(1)[:, 1].float(), (yg * yg).sum(1), rtol=1e-4, atol=0.5)


@pytest.mark.parametrize('N,Ci,Co', [(16, 2048, 512), (8, 2048, 512), (8, 256, 256)])
def test_big_tile_inference_batchnorm_epilogue(ops, N, Ci, Co):
    """conv + eval-mode BN + residual + ReLU on the halo tiles (an eval forward of 16 images, e.g. two TTA batches; the
    teacher's 8 images: its head conv and layer3's conv2)."""
    H, W, k = 32, 32, 3
    M = N * H * W
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, Ci, generator=g).to(BF).cuda()
    w = (torch.randn(Co, k * k, Ci, generator=g) * (2.0 / (Ci * k * k)) ** 0.5).to(BF).cuda()
    res = torch.randn(M, Co, generator=g).to(BF).cuda()
    rm, rv = torch.randn(Co, generator=g).cuda() * 0.1, (torch.rand(Co, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda() * 0.1
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    ops.conv2d_bneval(x, w, y, N, H, W, H, W, k, k, 1, 1, 1, rm, rv, gamma, beta, True, res)
    c = _ref_conv_gpu(x, w, N, H, W, k, 1, 1).to(BF).float()                # the kernel normalises the bf16-rounded conv
    ref = torch.relu((c - rm) / torch.sqrt(rv + 1e-5) * gamma + beta + res.float())
    # a conv output one bf16 ulp off moves the normalised value by ~1 %: compare in norm, bound the worst element
    assert rel_l2(y, ref) < 6e-3 and relerr(y.float(), ref) < 3e-2


@pytest.mark.parametrize('Cf,Cb,k,pad,dil', [(2048, 512, 3, 1, 1), (512, 512, 3, 2, 2), (256, 256, 3, 1, 1)])
def test_big_tile_data_gradient_with_fused_bn_backward(ops, Cf, Cb, k, pad, dil):
    """The data gradient of the head conv (512 -> 2048, residual = the other head's feature gradient) and of layer4's
    atrous conv2 (residual-free in the step; here with one) on the 128 x 256 tile: plain, and with the consumer
    BatchNorm's backward sums + ReLU sign mask folded into the epilogue (what _cbr_bwd launches for layer4)."""
    N, H, W, groups = 16, 32, 32, 2
    M = N * H * W
    g = torch.Generator().manual_seed(Cf + dil)
    dy = torch.randn(M, Cb, generator=g).to(BF).cuda()
    w = (torch.randn(Cb, k * k, Cf, generator=g) * (2.0 / (Cf * k * k)) ** 0.5).to(BF).cuda()     # forward weights [Cout][tap][Cin]
    wt = w.permute(2, 1, 0).contiguous()                                                        # [Cin][tap][Cout]
    res = torch.randn(M, Cf, generator=g).to(BF).cuda()
    # reference: autograd through the fp32 im2col matmul
    xr = torch.zeros(M, Cf, device='cuda', requires_grad=True)
    cols = F.unfold(xr.view(N, H, W, Cf).permute(0, 3, 1, 2), k, dil, pad, 1)
    cols = cols.view(N, Cf, k * k, H * W).permute(0, 3, 2, 1).reshape(M, k * k * Cf)
    (cols @ w.float().view(Cb, k * k * Cf).t()).backward(dy.float())
    ref = xr.grad + res.float()
    dx = torch.empty(M, Cf, dtype=BF, device='cuda')
    ops.conv2d(dy, wt, dx, N, H, W, H, W, k, k, 1, pad, dil, 1, res, None)
    assert rel_l2(dx, ref) < 4e-3 and relerr(dx.float(), ref) < 1.5e-2
    # fused BatchNorm-backward reduction of the consumer (y sign from a mask, per-group statistics)
    cy = torch.randn(M, Cf, generator=g).to(BF).cuda()
    cx = torch.randn(M, Cf, generator=g).to(BF).cuda()
    keep = (cy.float() > 0).cpu()
    mask = (keep.reshape(M, Cf // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).cuda()
    mi = torch.stack([torch.randn(groups, Cf, generator=g) * 0.1, torch.rand(groups, Cf, generator=g) + 0.5], 1).cuda().contiguous()
    for use_mask in (False, True):
        dx2 = torch.empty_like(dx)
        sums = ops.new_stats(groups, 8, 2, Cf)
        ops.conv2d_bnbwd(dy, wt, dx2, N, H, W, H, W, k, k, 1, pad, dil, 1, res, sums, groups,
                         None if use_mask else cy, cx, mi, True, relu_mask=mask if use_mask else None)
        assert torch.equal(dx2, dx)
        want = ops.new_stats(groups, 8, 2, Cf)
        ops.bn_bwd_reduce(dx2, cy, cx, mi, want, M, Cf, True, groups=groups)
        torch.testing.assert_close(ops.stats_value(sums, backward=True).sum(1).float(),
                                   ops.stats_value(want, backward=True).sum(1).float(), rtol=3e-4, atol=0.5)


# ----------------------------------------------------------------------------------------------------------------------
# BatchNorm + ReLU on the consumer's operand path (rgda_conv2d_bnin, rgda_maxpool_fwd_bnin, relu == 2 in the backward
# kernels): regda/_resnets.py:92-112's conv -> bn -> relu -> conv chain without the activation ever being written.
BNIN_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, kernel kind   -- the production geometries (bottleneck conv2 / conv3)
    (16, 32, 32, 256, 1024, 1, 1, 0, 1, 'igemm'),    # layer 3 conv3 (the dominant 1x1 kernel's tile)
    (4, 32, 32, 512, 2048, 1, 1, 0, 1, 'igemm'),     # layer 4 conv3
    (4, 64, 64, 128, 512, 1, 1, 0, 1, 'igemm'),      # layer 2 conv3
    (2, 128, 128, 64, 256, 1, 1, 0, 1, 'igemm'),     # layer 1 conv3
    (16, 64, 64, 128, 128, 3, 1, 1, 1, 'igemm'),     # layer 2 conv2
    (16, 128, 128, 128, 128, 3, 2, 1, 1, 'igemm'),   # layer 2 block 0 conv2 (stride 2: padding rows on the operand path)
    (16, 32, 32, 256, 256, 3, 1, 1, 1, 'halo4'),     # layer 3 conv2
    (16, 32, 32, 512, 512, 3, 1, 1, 1, 'halo8'),     # layer 4 block 0 conv2
    (4, 32, 64, 256, 256, 3, 1, 1, 1, 'halo4 wide'), # layer 3 conv2 of a 1024 x 1024 tile (64-column map: 32-column bands)
    (4, 32, 64, 512, 512, 3, 1, 1, 1, 'halo8 wide'), # layer 4 block 0 conv2, likewise
]


def _bn_ref(c, gamma, beta, groups, eps=1e-5):
    """Train-mode BatchNorm of NCHW fp32 `c` per group of images (biased variance), fp64 inside."""
    outs, means, vars_ = [], [], []
    for cg in c.double().chunk(groups, 0):
        m = cg.mean((0, 2, 3))
        v = cg.var((0, 2, 3), unbiased=False)
        outs.append((cg - m[None, :, None, None]) / torch.sqrt(v + eps)[None, :, None, None] * gamma.double()[None, :, None, None]
                    + beta.double()[None, :, None, None])
        means.append(m)
        vars_.append(v)
    return torch.cat(outs, 0).float(), torch.stack(means), torch.stack(vars_)


@pytest.mark.parametrize('case', BNIN_CASES)
def test_conv_with_batchnorm_relu_on_the_operand_path(ops, case):
    """rgda_conv2d_bnin against fp32 F.conv2d(F.relu(bn(c))) on the bf16-rounded raw operand c: output, its fused
    statistics, (mean, invstd) and the running statistics (src then tgt, unbiased variance, momentum 0.1); then the
    backward side of the same unit: rgda_bn_bwd_apply(relu = 2) against autograd, and the activation it writes for the
    weight gradient fed to the PLAIN convolution reproduces the operand-path convolution bit for bit."""
    N, H, W, Cin, Cout, k, s, p, d, kind = case
    G = 2
    gen = torch.Generator().manual_seed(1 + _case_seed(case))
    # the raw output of the producing convolution: per-channel offsets / scales so that BatchNorm has something to do
    c = rbf(torch.randn(N, Cin, H, W, generator=gen) * (0.5 + torch.rand(1, Cin, 1, 1, generator=gen)) +
            torch.randn(1, Cin, 1, 1, generator=gen))
    gamma = 0.5 + torch.rand(Cin, generator=gen)
    gamma[::7] *= -1.0                                  # negative scales too
    beta = 0.3 * torch.randn(Cin, generator=gen)
    w = rbf(torch.randn(Cout, Cin, k, k, generator=gen) * (2.0 / (Cin * k * k)) ** 0.5)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    M = N * Ho * Wo
    assert ops.conv2d_bnin_supported(M, Cout, Cin, k, k, s, p, d, H, W, Ho, Wo, G), 'geometry must be served'
    bnr, mean_ref, var_ref = _bn_ref(c, gamma, beta, G)
    a_ref = rbf(F.relu(bnr))
    ref = F.conv2d(a_ref, w, None, s, p, d)
    cg = to_pxc(c)
    wg = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin).to(BF).cuda().contiguous()
    # the producer's accumulators, as its epilogue would have left them
    pst = ops.new_stats(G, 8, 2, Cin)
    rows = N * H * W // G
    for g_ in range(G):
        ops.bn_stats(cg[g_ * rows:(g_ + 1) * rows], pst[g_], rows, Cin)
    mi = torch.zeros(G, 2, Cin, device='cuda')
    rm = torch.full((Cin,), 0.25, device='cuda')
    rv = torch.full((Cin,), 2.0, device='cuda')
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    gam, bet = gamma.cuda(), beta.cuda()
    bnop = ops.bn_operand(pst, gam, bet, mi, rm, rv, nbt, G, True)
    y = torch.zeros(M, Cout, dtype=BF, device='cuda')
    stats = ops.new_stats(G, 8, 2, Cout)
    ops.conv2d_bnin(bnop, cg, wg, y, N, H, W, Ho, Wo, k, k, s, p, d, None, stats, G)
    out = from_pxc(y, N, Ho, Wo)
    assert relerr(out, ref) < 1.2e-2, ('forward', relerr(out, ref))
    yf = y.float().view(G, M // G, Cout)
    sv = ops.stats_value(stats).sum(1)
    torch.testing.assert_close(sv[:, 0].float().cpu(), yf.sum(1).cpu(), rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(sv[:, 1].float().cpu(), (yf * yf).sum(1).cpu(), rtol=1e-3, atol=2e-2)
    # nn.BatchNorm2d's train-mode side effects, written by one workgroup of the launch
    torch.testing.assert_close(mi[:, 0].cpu(), mean_ref.float(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mi[:, 1].cpu(), (1.0 / torch.sqrt(var_ref + 1e-5)).float(), rtol=1e-4, atol=1e-5)
    n_g = N * H * W // G
    rm_ref, rv_ref = torch.full((Cin,), 0.25).double(), torch.full((Cin,), 2.0).double()
    for g_ in range(G):
        rm_ref = 0.9 * rm_ref + 0.1 * mean_ref[g_]
        rv_ref = 0.9 * rv_ref + 0.1 * var_ref[g_] * n_g / (n_g - 1)
    torch.testing.assert_close(rm.cpu(), rm_ref.float(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), rv_ref.float(), rtol=1e-4, atol=1e-5)
    assert int(nbt.item()) == G
    # ---- backward of the deferred unit: ReLU sign from c, the activation written for the weight gradient
    gact = rbf(torch.randn(N, Cin, H, W, generator=gen))              # d(loss) / d(activation)
    cr = c.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    outs = []
    for cgrp in cr.chunk(G, 0):
        outs.append(F.relu(F.batch_norm(cgrp, None, None, gr, br, True, 0.1, 1e-5)))
    torch.cat(outs, 0).backward(gact)
    gg = to_pxc(gact)
    sums = ops.new_stats(G, 8, 2, Cin)
    ops.bn_bwd_reduce(gg, None, cg, mi, sums, N * H * W, Cin, 2, groups=G, gamma=gam, beta=bet)
    dcx = torch.zeros(N * H * W, Cin, dtype=BF, device='cuda')
    act = torch.zeros(N * H * W, Cin, dtype=BF, device='cuda')
    dgam, dbet = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    ops.bn_bwd_apply(gg, None, cg, mi, gam, sums, dcx, N * H * W, Cin, 2, None, dgam, dbet, groups=G, beta=bet, act_out=act)
    # (elements whose pre-ReLU value is within fp32 rounding of zero may take either sign -- fma(x, scale, shift) here,
    # (x - mean) * invstd * gamma + beta in the reference; with 3.4e7 elements in the largest case a few exist for most
    # seeds, and ONE flipped sign is a max-norm error of |g| k0.  They are excluded from the comparison.)
    tie = bnr.abs() < 2e-6 * (1.0 + c.abs() * gamma.abs().view(1, -1, 1, 1))
    assert tie.float().mean().item() < 1e-4
    dref = torch.where(tie, torch.zeros(()), cr.grad)
    dhip = torch.where(tie, torch.zeros(()), from_pxc(dcx, N, H, W))
    assert relerr(dhip, dref) < 1.5e-2, 'bn backward (relu sign from x)'
    assert relerr(dgam.cpu(), gr.grad) < 5e-3 and relerr(dbet.cpu(), br.grad) < 5e-3
    assert relerr(from_pxc(act, N, H, W), a_ref) < 1e-2
    # the same bits the operand path fed the matrix pipe: the plain convolution over `act` gives the identical result
    y2 = torch.zeros(M, Cout, dtype=BF, device='cuda')
    st2 = ops.new_stats(G, 8, 2, Cout)
    ops.conv2d(act, wg, y2, N, H, W, Ho, Wo, k, k, s, p, d, 0, None, st2, G)
    assert torch.equal(y2, y), 'operand-path activation != the activation written for the weight gradient'
    assert torch.equal(st2, stats)
    # the data-gradient convolution with the fused BatchNorm-backward sums of this unit (relu == 2: sign from c)
    dyn = rbf(torch.randn(N, Cout, Ho, Wo, generator=gen) * 0.1)
    ar = a_ref.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(ar, wr, None, s, p, d).backward(dyn)
    # the consumer's WEIGHT gradient from the activation the backward apply wrote (its operand in the model: Encoder._cbr_bwd)
    dw = torch.zeros(Cout, k * k, Cin, device='cuda')
    ops.conv2d_wgrad(act, to_pxc(dyn), dw, N, H, W, Ho, Wo, k, k, s, p, d)
    assert relerr(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)) < 1e-2, 'wgrad from the side output'
    w = w.detach()
    wt = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).to(BF).cuda().contiguous()
    da = torch.zeros(N * H * W, Cin, dtype=BF, device='cuda')
    fsums = ops.new_stats(G, 8, 2, Cin)
    try:
        ops.conv2d_bnbwd(to_pxc(dyn), wt, da, N, Ho, Wo, H, W, k, k, s, p, d, 1, None, fsums, G, None, cg, mi, 2,
                         bn_gamma=gam, bn_beta=bet)
    except ValueError:
        return          # row groups do not tile for this geometry: the model falls back to rgda_bn_bwd_reduce
    assert relerr(from_pxc(da, N, H, W), ar.grad) < 1.5e-2
    rsums = ops.new_stats(G, 8, 2, Cin)
    ops.bn_bwd_reduce(da, None, cg, mi, rsums, N * H * W, Cin, 2, groups=G, gamma=gam, beta=bet)
    a_, b_ = ops.stats_value(fsums, True).sum(1), ops.stats_value(rsums, True).sum(1)
    torch.testing.assert_close(a_.float().cpu(), b_.float().cpu(), rtol=2e-3, atol=2e-2)


def test_maxpool_with_batchnorm_relu_on_the_operand_path(ops):
    """rgda_maxpool_fwd_bnin against F.max_pool2d(F.relu(bn(c))) on the stem geometry (64 channels, 2 groups), and the
    argmax taps against the materialised route (rgda_maxpool_fwd over the same bf16 activation)."""
    N, C, H, W, G = 4, 64, 64, 128, 2
    gen = torch.Generator().manual_seed(5)
    c = rbf(torch.randn(N, C, H, W, generator=gen) * 1.5 + 0.2)
    gamma = 0.5 + torch.rand(C, generator=gen)
    gamma[::5] *= -1.0
    beta = 0.2 * torch.randn(C, generator=gen)
    bnr, mean_ref, var_ref = _bn_ref(c, gamma, beta, G)
    a_ref = rbf(F.relu(bnr))
    ref = F.max_pool2d(a_ref, 3, 2, 1)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    cg = to_pxc(c)
    pst = ops.new_stats(G, 8, 2, C)
    rows = N * H * W // G
    for g_ in range(G):
        ops.bn_stats(cg[g_ * rows:(g_ + 1) * rows], pst[g_], rows, C)
    mi = torch.zeros(G, 2, C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    gam, bet = gamma.cuda(), beta.cuda()
    y = torch.zeros(N * Ho * Wo, C, dtype=BF, device='cuda')
    idx = torch.zeros(N * Ho * Wo, C, dtype=torch.uint8, device='cuda')
    ops.maxpool_fwd_bnin(ops.bn_operand(pst, gam, bet, mi, rm, rv, nbt, G, True), cg, y, idx, N, H, W, C, Ho, Wo)
    assert relerr(from_pxc(y, N, Ho, Wo), ref) < 1e-2
    torch.testing.assert_close(mi[:, 0].cpu(), mean_ref.float(), rtol=1e-4, atol=1e-5)
    assert int(nbt.item()) == G
    # the activation rgda_bn_bwd_apply(relu = 2) writes is the operand the pooling saw: pooling it reproduces y and idx
    sums = ops.new_stats(G, 8, 2, C)
    gz = torch.zeros(N * H * W, C, dtype=BF, device='cuda')
    dcx, act = torch.zeros_like(gz), torch.zeros_like(gz)
    ops.bn_bwd_apply(gz, None, cg, mi, gam, sums, dcx, N * H * W, C, 2, groups=G, beta=bet, act_out=act)
    y2, idx2 = torch.zeros_like(y), torch.zeros_like(idx)
    ops.maxpool_fwd(act, y2, idx2, N, H, W, C, Ho, Wo)
    assert torch.equal(y2, y) and torch.equal(idx2, idx)


def test_grouped_small_convolutions_equal_the_single_launches(ops):
    """rgda_conv2d_grouped: the PPM branches' problems -- 2048 -> 512 on s x s maps per statistics group, 512 -> 4608,
    4608 -> 512 and the 512 -> 2048 data gradients with a residual -- in shared launches: bit-identical outputs AND
    accumulators to one rgda_conv2d each (same kernel, same tiles); a problem that another kernel serves rides along as its
    own launch; sixteen small problems make two launches."""
    g = torch.Generator().manual_seed(91)
    N = 16

    def mk(M, C):
        return (torch.randn(M, C, generator=g) * 0.5).to(BF).cuda()

    def wt(Co, Ci):
        return (torch.randn(Co, 1, Ci, generator=g) * 0.05).to(BF).cuda()
    items, refs = [], []
    for s_ in (1, 2, 3, 6):                                # branch convolutions: one problem per scale and statistics group
        for grp in range(2):
            Ng = N // 2
            x, w = mk(Ng * s_ * s_, 2048), wt(512, 2048)
            items.append([x, w, None, Ng, s_, s_, s_, s_, 1, 1, 1, 0, 1, 0, None, 'stats', 1])
    for s_ in (1, 2, 3, 6):                                # Z = q W^T
        items.append([mk(N * s_ * s_, 512), wt(4608, 512), None, N, s_, s_, s_, s_, 1, 1, 1, 0, 1])
    for s_ in (1, 6):                                      # dq = dZ W (K = 4608) and a data gradient with a residual
        items.append([mk(N * s_ * s_, 4608), wt(512, 4608), None, N, s_, s_, s_, s_, 1, 1, 1, 0, 1])
        items.append([mk(N * s_ * s_, 512), wt(2048, 512), None, N, s_, s_, s_, s_, 1, 1, 1, 0, 1, 1, mk(N * s_ * s_, 2048)])
    items.append([mk(4 * 32 * 32, 256), wt(1024, 256), None, 4, 32, 32, 32, 32, 1, 1, 1, 0, 1])      # a large one: own launch
    outs_a, outs_b, st_a, st_b = [], [], [], []
    a_items, b_items = [], []
    for it in items:
        M, Co = it[3] * it[6] * it[7], it[1].shape[0]
        ya, yb = torch.zeros(M, Co, dtype=BF, device='cuda'), torch.zeros(M, Co, dtype=BF, device='cuda')
        ia, ib = list(it), list(it)
        ia[2], ib[2] = ya, yb
        if len(it) > 15 and it[15] == 'stats':
            sa, sb = ops.new_stats(1, 8, 2, Co), ops.new_stats(1, 8, 2, Co)
            ia[15], ib[15] = sa, sb
            st_a.append(sa); st_b.append(sb)
        outs_a.append(ya); outs_b.append(yb)
        a_items.append(tuple(ia)); b_items.append(tuple(ib))
    assert ops.conv2d_grouped_launches(a_items) == 1 + 2      # 16 small problems in two launches + the large one on its own
    ops.conv2d_grouped(a_items)
    for it in b_items:
        kw = {}
        ops.conv2d(*it[:13], *(it[13:17] if len(it) > 13 else ()))
    torch.cuda.synchronize()
    for i, (ya, yb) in enumerate(zip(outs_a, outs_b)):
        assert torch.equal(ya, yb), i
        assert float(ya.float().abs().sum()) > 0
    for sa, sb in zip(st_a, st_b):       # (a workgroup's replica is its XCD: the partials land in other replicas, the totals agree)
        assert torch.equal(sa.sum(1), sb.sum(1)) and int(sa.sum(1).abs().sum()) > 0
    # against fp32 on the same operands (first and last small problem)
    for it, y in ((a_items[0], outs_a[0]), (a_items[15], outs_a[15])):
        ref = it[0].float() @ it[1].float().view(it[1].shape[0], -1).t()
        if len(it) > 14 and it[14] is not None:
            ref = ref + it[14].float()
        assert relerr(y.float().cpu(), ref.cpu()) < 6e-3
    # a wrong descriptor: nothing is launched
    bad = list(a_items[0]); bad[3] = 0
    y0 = outs_a[1].clone()
    outs_a[1].zero_()
    with pytest.raises(ValueError):
        ops.conv2d_grouped([a_items[1], tuple(bad)])
    torch.cuda.synchronize()
    assert float(outs_a[1].float().abs().sum()) == 0.0 and float(y0.float().abs().sum()) > 0
    ops.conv2d_grouped([])


def test_small_map_batchnorm_several_layers_per_launch(ops):
    """rgda_bn_train_small / rgda_bn_bwd_small: the PPM branches' BatchNorm + ReLU on s x s maps (8 - 288 rows per
    statistics group), four scales per launch -- against torch autograd per group in fp32 on the same bf16 inputs, and
    against the general kernels (statistics from the accumulators, reduce + apply); C = 72 leaves a partial channel block,
    ten layers make two launches, strided views on both sides."""
    g = torch.Generator().manual_seed(44)
    N, G = 16, 2
    cases = [(s_, 512) for s_ in (1, 2, 3, 6)] + [(2, 72), (3, 1024)] + [(1, 512)] * 4
    fwd, bwd, keep = [], [], []
    for li, (s_, C) in enumerate(cases):
        M = N * s_ * s_
        xw = (torch.randn(M, C + 8, generator=g) * 1.5 + 0.3).to(BF).cuda()
        x = xw[:, :C] if li % 2 else xw[:, :C].contiguous()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
        rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
        y = torch.zeros(M, C, dtype=BF, device='cuda')
        mi = torch.zeros(G, 2, C, device='cuda')
        use_mask = li % 3 != 0
        mask = torch.zeros(M, C // 8, dtype=torch.uint8, device='cuda') if use_mask else None
        go = (torch.randn(M, C, generator=g)).to(BF).cuda()
        dx = torch.zeros(M, C, dtype=BF, device='cuda')
        dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        fwd.append((x, y, mi, rm, rv, nbt, gamma, beta, M, C, True, G, mask))
        bwd.append((go, None if use_mask else y, x, mi, gamma, dx, dgam, dbet, M, C, True, G, mask))
        keep.append((xw, x, gamma, beta, rm, rv, nbt, y, mi, mask, go, dx, dgam, dbet, M, C))
    ops.bn_train_small(fwd)
    ops.bn_bwd_small(bwd)
    torch.cuda.synchronize()
    for li, (xw, x, gamma, beta, rm, rv, nbt, y, mi, mask, go, dx, dgam, dbet, M, C) in enumerate(keep):
        Mg = M // G
        xr = x.float().cpu().view(G, Mg, C).clone().requires_grad_(True)
        gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
        rm2, rv2 = torch.zeros(C), torch.ones(C)
        outs = [F.relu(F.batch_norm(xr[gi].t().reshape(1, C, Mg), rm2, rv2, gr, br, True, 0.1, 1e-5)) for gi in range(G)]
        yref = torch.cat([o.reshape(C, Mg).t() for o in outs], 0)
        yref.backward(go.float().cpu())
        assert relerr(y.float().cpu(), yref.detach()) < 1e-2, li
        torch.testing.assert_close(rm.cpu(), rm2, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(rv.cpu(), rv2, rtol=2e-4, atol=2e-5)
        assert int(nbt) == G
        torch.testing.assert_close(mi[:, 0].cpu(), xr.detach().mean(1), rtol=1e-4, atol=1e-5)
        if mask is not None:
            assert torch.equal(unpack_mask(mask, C), (y > 0).cpu())
        # the gradient: tolerance of the general kernels' test (bf16 output, fp32 sums in another order)
        assert relerr(dx.float().cpu(), xr.grad.reshape(M, C)) < 2e-2, li
        assert relerr(dgam.cpu(), gr.grad) < 1e-2 and relerr(dbet.cpu(), br.grad) < 1e-2
        # ... and against the general kernels on the same tensors
        st = ops.new_stats(G, 8, 2, C)
        xc = x.contiguous()
        ops.bn_stats(xc[:Mg], st[0], Mg, C)
        ops.bn_stats(xc[Mg:], st[1], Mg, C)
        mi2 = torch.empty(G, 2, C, device='cuda')
        y2 = torch.empty(M, C, dtype=BF, device='cuda')
        rm3, rv3, nbt3 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
        ops.bn_train_apply(xc, st, mi2, rm3, rv3, nbt3, gamma, beta, y2, M, C, True, None, None, 1, groups=G)
        torch.testing.assert_close(mi, mi2, rtol=1e-5, atol=1e-6)
        assert (y.float() - y2.float()).abs().max().item() <= 2 * 2.0 ** -8 * max(1.0, y2.float().abs().max().item())
        torch.testing.assert_close(rm, rm3, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, rv3, rtol=1e-5, atol=1e-6)
    # two runs: identical bits (fixed summation order, no atomics but the parameter-gradient adds of ONE workgroup each)
    y_first = [k[7].clone() for k in keep]
    dx_first = [k[11].clone() for k in keep]
    for k in keep:
        k[4].zero_(); k[5].fill_(1.0); k[6].zero_(); k[12].zero_(); k[13].zero_()
    ops.bn_train_small(fwd)
    ops.bn_bwd_small(bwd)
    torch.cuda.synchronize()
    for k, y0, d0 in zip(keep, y_first, dx_first):
        assert torch.equal(k[7], y0) and torch.equal(k[11], d0)
    # argument checks: one row per group, too many rows
    x1 = torch.zeros(2, 64, dtype=BF, device='cuda')
    with pytest.raises(ValueError):
        ops.bn_train_small([(x1, x1.clone(), torch.zeros(2, 2, 64, device='cuda'), None, None, None, torch.ones(64, device='cuda'),
                             torch.zeros(64, device='cuda'), 2, 64, True, 2, None)])
    big = torch.zeros(2 * 328, 64, dtype=BF, device='cuda')
    with pytest.raises(ValueError):
        ops.bn_train_small([(big, big.clone(), torch.zeros(2, 2, 64, device='cuda'), None, None, None, torch.ones(64, device='cuda'),
                             torch.zeros(64, device='cuda'), 2 * 328, 64, True, 2, None)])


def test_wide_maps_select_the_banded_halo_kernel():
    """The library's own dispatch (rgda_conv2d_kernel reports it, no launch): long-K 3x3 convolutions on maps whose width is a
    multiple of 32 beyond 32 take conv3x3_halo_wide_kernel (the CASES above with 64 / 96 columns check its results); 32-wide
    maps keep conv3x3_halo_kernel; widths that are no multiple of 32 the implicit-GEMM kernel."""
    from regda_amd._lib import lib
    kname = lib().raw('rgda_conv2d_kernel')
    kname.restype = __import__('ctypes').c_char_p

    def name(N, H, W, Cin, Cout, dil=1, mode=0):
        r = kname(0, N, H, W, Cin, H, W, Cout, 3, 3, 1, dil, dil, mode, 1, 1)
        return r.decode() if r else None
    assert name(4, 32, 64, 512, 512) == 'conv3x3_halo_wide_kernel<1, 8, 4, true, false>'
    assert name(4, 32, 64, 512, 512, dil=2, mode=1) == 'conv3x3_halo_wide_kernel<2, 8, 3, true, false>'
    assert name(4, 32, 64, 256, 256) == 'conv3x3_halo_wide_kernel<1, 4, 4, true, false>'
    assert name(2, 40, 96, 512, 512) == 'conv3x3_halo_wide_kernel<1, 8, 4, true, false>'
    assert name(16, 32, 32, 512, 512) == 'conv3x3_halo_kernel<1, 8, false, 4, true>'
    assert name(14, 24, 24, 512, 512).startswith('conv_igemm_kernel<')
