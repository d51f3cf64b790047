import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
SHAPES = [  # N,H,W,Cin,Cout,k,s,p,d
    (8, 32, 32, 512, 2048, 1, 1, 0, 1),
    (8, 32, 32, 2048, 512, 1, 1, 0, 1),
    (16, 32, 32, 512, 2048, 1, 1, 0, 1),
    (16, 64, 64, 128, 512, 1, 1, 0, 1),
    (16, 128, 128, 64, 256, 1, 1, 0, 1),
    (16, 128, 128, 64, 64, 3, 1, 1, 1),
    (16, 32, 32, 4096, 512, 3, 1, 1, 1),
    (16, 32, 32, 256, 256, 3, 1, 1, 1),
    (16, 32, 32, 1024, 256, 1, 1, 0, 1),
    (16, 32, 32, 256, 1024, 1, 1, 0, 1),
    (8, 32, 32, 256, 256, 3, 1, 1, 1),
    (8, 32, 32, 1024, 256, 1, 1, 0, 1),
    (8, 32, 32, 256, 1024, 1, 1, 0, 1),
    (8, 32, 32, 4096, 512, 3, 1, 1, 1),
    (8, 128, 128, 64, 256, 1, 1, 0, 1),
]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, H, W, Ci, Co, k, s, p, d) in SHAPES:
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    M = N * Ho * Ho
    x = torch.randn(N * H * W, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    wt = (torch.randn(Ci, k * k, Co, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    dy = torch.randn(M, Co, device='cuda').to(BF)
    dx = torch.empty(N * H * W, Ci, dtype=BF, device='cuda')
    dw = torch.zeros(Co, k * k, Ci, device='cuda')
    fl = 2.0 * M * Co * Ci * k * k
    line = '%-34s' % str((M, Ci, Co, k))
    for tile in sys.argv[1:] or ['']:
        if tile: os.environ['RGDA_TILE'] = tile
        t = bench(lambda: ops.conv2d(x, w, y, N, H, W, Ho, Ho, k, k, s, p, d, 0))
        line += ' | %s fwd %6.1fus %6.0fTF' % (tile, t * 1e3, fl / t / 1e9)
    os.environ.pop('RGDA_TILE', None)
    t = bench(lambda: ops.conv2d_wgrad(x, dy, dw, N, H, W, Ho, Ho, k, k, s, p, d))
    line += ' | wgrad %6.1fus %6.0fTF' % (t * 1e3, fl / t / 1e9)
    print(line)
