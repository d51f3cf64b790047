"""dev: what a plain device copy / elementwise pass achieves on the BatchNorm passes' tensor sizes (the practical floor
for bn_apply-like kernels), producer-hot (tensor just written by another kernel) and rotating buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
def bench(fn, n=40):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, C) in [(16384, 256), (16384, 1024), (65536, 512), (262144, 256)]:
    R = 8
    x = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    y = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    r = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    mb = M * C * 2 / 1e6
    t_copy = bench(lambda i: y[i % R].copy_(x[i % R]))
    t_hot = bench(lambda i: y[0].copy_(x[0]))
    t_relu = bench(lambda i: torch.clamp(x[i % R], min=0, out=y[i % R]))
    t_add = bench(lambda i: torch.add(x[i % R], r[i % R], out=y[i % R]))
    G = 2
    stats = torch.rand(G, 8, 2, C, device='cuda'); stats[:, :, 1] += 4.0 * M / G / 8
    mi = torch.zeros(G, 2, C, device='cuda')
    rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    t_bn = bench(lambda i: ops.bn_train_apply(x[i % R], stats, mi, rm, rv, nbt, gamma, beta, y[i % R], M, C, True, None, None, 0, groups=G))
    t_bnr = bench(lambda i: ops.bn_train_apply(x[i % R], stats, mi, rm, rv, nbt, gamma, beta, y[i % R], M, C, True, r[i % R], None, 0, groups=G))
    t_bnh = bench(lambda i: ops.bn_train_apply(x[0], stats, mi, rm, rv, nbt, gamma, beta, y[0], M, C, True, None, None, 0, groups=G))
    print('M=%-7d C=%-5d %6.1f MB | torch copy %6.1f us (%4.1f TB/s) hot %6.1f | relu %6.1f | add(3 streams) %6.1f (%4.1f TB/s) | bn_train_apply %6.1f hot %6.1f | +res %6.1f (%4.1f TB/s)' % (
        M, C, mb, t_copy, 2 * mb / t_copy, t_hot, t_relu, t_add, 3 * mb / t_add, t_bn, t_bnh, t_bnr, 3 * mb / t_bnr))
