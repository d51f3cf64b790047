"""dev: kernel time with the K loop's DMA and/or compute removed (RGDA_CONV_SKIP), no timestamps inside."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
def bench(fn, n=20):
    for i in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Ci, Co, k, p, d) in [(16, 32, 32, 256, 1024, 1, 0, 1), (16, 128, 128, 64, 256, 1, 0, 1), (16, 64, 64, 128, 512, 1, 0, 1)]:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    KT = k * k * Ci // 64
    for tile in sys.argv[1:]:
        os.environ['RGDA_TILE'] = tile
        line = '%-26s %-11s KT=%-4d' % ((M, Ci, Co, k), tile, KT)
        ts = []
        for skip in ('0', '1', '2', '3', '3'):
            os.environ['RGDA_CONV_SKIP'] = skip
            ts.append(bench(lambda: ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)))
        os.environ.pop('RGDA_CONV_SKIP')
        line += ' full %7.1f | no-DMA %7.1f | no-compute %7.1f | neither %7.1f | neither %7.1f us  -> per K tile (ns): full %5.0f compute %5.0f dma %5.0f loop %5.0f' % (
            ts[0], ts[1], ts[2], ts[3], ts[4], 0, 0, 0, 0)
        print(line)
