import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
G = 2
for (N, H, W, Cin, Cout) in [(16, 64, 64, 512, 128), (16, 32, 32, 1024, 256)]:
    M = N * H * W
    gen = torch.Generator(device='cuda').manual_seed(1)
    w0 = (torch.randn(Cin, 1, Cin // 4, device='cuda', generator=gen) * 0.05).to(BF)
    a2 = torch.randn(M, Cin // 4, device='cuda', generator=gen).relu().to(BF)
    c3 = torch.empty(M, Cin, dtype=BF, device='cuda')
    st3 = ops.new_stats(G, 8, 2, Cin)
    ops.conv2d(a2, w0, c3, N, H, W, H, W, 1, 1, 1, 0, 1, 0, None, st3, G)
    res = torch.randn(M, Cin, device='cuda', generator=gen).relu().to(BF)
    gamma = (torch.rand(Cin, device='cuda', generator=gen) * 0.2 + 0.05)
    beta = torch.randn(Cin, device='cuda', generator=gen) * 0.1
    w1 = (torch.randn(Cout, 1, Cin, device='cuda', generator=gen) * 0.03).to(BF)
    mi = torch.empty(G, 2, Cin, device='cuda')
    for rep in range(3):
        yB = torch.full((M, Cin), 7.0, dtype=BF, device='cuda')
        mB = torch.empty(M, Cin // 8, dtype=torch.uint8, device='cuda')
        cB = torch.empty(M, Cout, dtype=BF, device='cuda')
        sB = ops.new_stats(G, 8, 2, Cout)
        bo = ops.bn_operand(st3, gamma, beta, mi, None, None, None, groups=G, relu=True)
        ops.conv1x1_block(bo, c3, res, yB, w1, cB, side_mask=mB, stats=sB, stat_groups=G)
        torch.cuda.synchronize()
        x = c3.float().view(G, M // G, Cin)
        mean = mi[:, 0].view(G, 1, Cin); istd = mi[:, 1].view(G, 1, Cin)
        ref = (((x - mean) * istd * gamma + beta) + res.float().view(G, M // G, Cin)).relu().view(M, Cin)
        d = (yB.float() - ref).abs()
        bad = (d > 0.05).nonzero()
        print(Cin, Cout, 'rep', rep, 'bad elements', bad.shape[0], 'max', float(d.max()))
        if bad.shape[0]:
            rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
            print('   rows', rows[:12].tolist(), '... n', rows.numel(), ' row%64', (rows % 64).unique()[:16].tolist(),
                  ' tiles', (rows // 64).unique()[:10].tolist(), ' col//64', (cols // 64).unique().tolist(), 'vals', yB[bad[0, 0], bad[0, 1]].item(), ref[bad[0, 0], bad[0, 1]].item())
        cref = (yB.float() @ w1.float().view(Cout, Cin).t())
        print('   conv rel L2 vs (side @ w)', float((cB.float() - cref).norm() / cref.norm()))
