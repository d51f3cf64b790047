"""dev: BN apply / backward-apply bandwidth on rotating (cache-cold) buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
def bench(fn, n=48):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
R = 12
CFG = [dict()]
for (M, C, res) in [(16384, 256, 0), (16384, 1024, 1), (65536, 128, 0), (65536, 512, 1), (262144, 64, 0), (262144, 256, 1), (16384, 512, 0), (16384, 2048, 1)]:
    G = 2
    x = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    y = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    r = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    g = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    dx = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    gm = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    mk = [torch.zeros(M, C // 8, dtype=torch.uint8, device='cuda') for _ in range(R)]
    stats = torch.rand(G, 8, 2, C, device='cuda')
    stats[:, :, 1] += 4.0 * M / G / 8
    mi = torch.zeros(G, 2, C, device='cuda')
    rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    sums = torch.zeros(G, 8, 2, C, device='cuda')
    for cfg in CFG:
        os.environ.update(cfg)
        t1 = bench(lambda i: ops.bn_train_apply(x[i % R], stats, mi, rm, rv, nbt, gamma, beta, y[i % R], M, C, True, r[i % R] if res else None, None, 0, groups=G, relu_mask=mk[i % R]))
        t2 = bench(lambda i: ops.bn_bwd_apply(g[i % R], None, x[i % R], mi, gamma, sums, dx[i % R], M, C, True, gm[i % R] if res else None, dgam, dbet, None, 0, groups=G, relu_mask=mk[i % R]))
        for k in cfg: os.environ.pop(k)
        print('M=%-7d C=%-5d res=%d %-14s| train_apply %6.1fus | bwd_apply %6.1fus' % (M, C, res, ','.join(cfg.values()) or 'default', t1, t2))
