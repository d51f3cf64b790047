"""Isolated cost of the BN-statistics epilogue on the HBM-bound small-K convolutions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
shapes = [(16, 128, 128, 64, 256), (16, 128, 128, 256, 64), (16, 64, 64, 128, 512), (16, 64, 64, 512, 128), (16, 64, 64, 1024, 256), (16, 64, 64, 256, 1024)]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Ci, Co) in shapes:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, 1, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    res = torch.randn(M, Co, device='cuda').to(BF)
    st = torch.zeros(2, 8, 2, Co, device='cuda')
    byt = (M * Ci + M * Co) * 2
    t0 = bench(lambda: ops.conv2d(x, w, y, N, H, W, H, W, 1, 1, 1, 0, 1, 0))
    t1 = bench(lambda: ops.conv2d(x, w, y, N, H, W, H, W, 1, 1, 1, 0, 1, 0, stats=st, stat_groups=2))
    t2 = bench(lambda: ops.conv2d(x, w, y, N, H, W, H, W, 1, 1, 1, 0, 1, 0, res=res))
    print((M, Ci, Co), 'plain %.1f us (%.2f TB/s) | +stats %.1f us | +res %.1f us (%.2f TB/s)' % (t0, byt / t0 / 1e6, t1, t2, (byt + M * Co * 2) / t2 / 1e6))
