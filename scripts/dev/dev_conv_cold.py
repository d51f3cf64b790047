"""dev: isolated conv timings with (a) BN statistics in the epilogue, (b) rotating buffers (cold L2 / MALL)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
SHAPES = [(16, 32, 32, 256, 256, 3, 1, 1), (16, 32, 32, 1024, 256, 1, 0, 1), (16, 32, 32, 256, 1024, 1, 0, 1)]
def bench(fn, n=40):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Ci, Co, k, p, d) in SHAPES:
    M = N * H * W
    R = 24
    xs = [torch.randn(M, Ci, device='cuda').to(BF) for _ in range(R)]
    ws = [(torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF) for _ in range(R)]
    ys = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(R)]
    st = torch.zeros(R, 2 * 8 * 2 * Co, device='cuda')
    line = '%-28s' % str((M, Ci, Co, k))
    for name, rot, stats in (('warm', 0, 0), ('warm+stats', 0, 1), ('cold', 1, 0), ('cold+stats', 1, 1), ('cold+stats g2', 1, 2)):
        def fn(i):
            j = (i % R) if rot else 0
            ops.conv2d(xs[j], ws[j], ys[j], N, H, W, H, W, k, k, 1, p, d, 0, None, st[j] if stats else None, max(stats, 1))
        line += ' | %s %5.1fus' % (name, bench(fn))
    print(line)
