"""dev: the tail of a bottleneck block on the next conv1's operand path (rgda_conv1x1_block) against the two launches it
replaces (rgda_bn_train_apply with residual + ReLU + sign mask, then the plain 1x1 convolution with statistics): results
and isolated timings on rotating buffers, layer-3 / layer-4 / layer-2 geometry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
G = 2


def t_of(fn, reps=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, H, W, Cin, Cout, name) in [(16, 32, 32, 1024, 256, 'layer3'), (16, 64, 64, 512, 128, 'layer2'), (16, 128, 128, 256, 128, 'layer1->2.0')]:
    M = N * H * W
    NB = 3
    gen = torch.Generator(device='cuda').manual_seed(1)
    w0 = (torch.randn(Cin, 1, Cin // 4, device='cuda', generator=gen) * 0.05).to(BF)           # the producing conv3
    a2 = [torch.randn(M, Cin // 4, device='cuda', generator=gen).relu().to(BF) for _ in range(NB)]
    c3 = [torch.empty(M, Cin, dtype=BF, device='cuda') for _ in range(NB)]
    st3 = [ops.new_stats(G, 8, 2, Cin) for _ in range(NB)]
    for i in range(NB):
        ops.conv2d(a2[i], w0, c3[i], N, H, W, H, W, 1, 1, 1, 0, 1, 0, None, st3[i], G)
    res = [torch.randn(M, Cin, device='cuda', generator=gen).relu().to(BF) for _ in range(NB)]
    gamma = (torch.rand(Cin, device='cuda', generator=gen) * 0.2 + 0.05)
    beta = torch.randn(Cin, device='cuda', generator=gen) * 0.1
    w1 = (torch.randn(Cout, 1, Cin, device='cuda', generator=gen) * 0.03).to(BF)
    mi = torch.empty(G, 2, Cin, device='cuda')
    rm, rv, nbt = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    # route A: apply pass + plain convolution
    yA = [torch.empty(M, Cin, dtype=BF, device='cuda') for _ in range(NB)]
    mA = [torch.empty(M, Cin // 8, dtype=torch.uint8, device='cuda') for _ in range(NB)]
    cA = [torch.empty(M, Cout, dtype=BF, device='cuda') for _ in range(NB)]
    sA = [ops.new_stats(G, 8, 2, Cout) for _ in range(NB)]
    it = [0]
    def routeA():
        i = it[0] % NB; it[0] += 1
        ops.bn_train_apply(c3[i], st3[i], mi, rm, rv, nbt, gamma, beta, yA[i], M, Cin, True, res=res[i], groups=G, relu_mask=mA[i])
        ops.conv2d(yA[i], w1, cA[i], N, H, W, H, W, 1, 1, 1, 0, 1, 0, None, sA[i], G)
    yB = [torch.empty(M, Cin, dtype=BF, device='cuda') for _ in range(NB)]
    mB = [torch.empty(M, Cin // 8, dtype=torch.uint8, device='cuda') for _ in range(NB)]
    cB = [torch.empty(M, Cout, dtype=BF, device='cuda') for _ in range(NB)]
    sB = [ops.new_stats(G, 8, 2, Cout) for _ in range(NB)]
    mi2 = torch.empty(G, 2, Cin, device='cuda')
    rm2, rv2, nbt2 = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    assert ops.conv1x1_block_supported(M, Cout, Cin, G), (M, Cout, Cin)
    def routeB():
        i = it[0] % NB; it[0] += 1
        bo = ops.bn_operand(st3[i], gamma, beta, mi2, rm2, rv2, nbt2, groups=G, relu=True)
        ops.conv1x1_block(bo, c3[i], res[i], yB[i], w1, cB[i], side_mask=mB[i], stats=sB[i], stat_groups=G)
    it[0] = 0; routeA(); it[0] = 0; routeB(); torch.cuda.synchronize()
    fa, fb = yA[0].float(), yB[0].float()
    print('%-12s side output: max abs diff %.3g, differing elements %.4f %%, mask equal %.6f, conv out rel L2 %.3g; mi diff %.2g' % (
        name, float((fa - fb).abs().max()), 100.0 * float((yA[0] != yB[0]).float().mean()), float((mA[0] == mB[0]).float().mean()),
        float((cA[0].float() - cB[0].float()).norm() / cA[0].float().norm()), float((mi - mi2).abs().max())), flush=True)
    # the plain convolution over the fused kernel's OWN side output reproduces its convolution output bit for bit
    cC, sC = torch.empty(M, Cout, dtype=BF, device='cuda'), ops.new_stats(G, 8, 2, Cout)
    ops.conv2d(yB[0], w1, cC, N, H, W, H, W, 1, 1, 1, 0, 1, 0, None, sC, G)
    torch.cuda.synchronize()
    print('             plain conv over the side output == fused conv output: %s; statistics totals equal: %s' % (
        torch.equal(cC, cB[0]), torch.equal(sC.sum(1), sB[0].sum(1))), flush=True)
    ta, tb = t_of(routeA), t_of(routeB)
    ta2, tb2 = t_of(routeA), t_of(routeB)
    print('             apply + conv %.1f / %.1f us   fused %.1f / %.1f us' % (ta, ta2, tb, tb2), flush=True)
