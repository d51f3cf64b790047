"""dev: generic wgrad timings under tuning envs (RGDA_WGRAD_EPI / _MINKT / _SPLITS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
SHAPES = [(16, 32, 32, 256, 1024, 1, 0, 1), (16, 32, 32, 1024, 256, 1, 0, 1), (16, 32, 32, 256, 256, 3, 1, 1), (16, 64, 64, 128, 512, 1, 0, 1),
          (16, 128, 128, 64, 256, 1, 0, 1), (16, 32, 32, 512, 2048, 1, 0, 1)]
def bench(fn, n=30):
    for i in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cfgs = [dict(), dict(RGDA_WGRAD_STAGES='4'), dict(RGDA_WGRAD_STAGES='5'), dict(RGDA_WGRAD_STAGES='5', RGDA_WGRAD_MINKT='32')]
for (N, H, W, Ci, Co, k, p, d) in SHAPES:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    dy = torch.randn(M, Co, device='cuda').to(BF)
    dw = torch.zeros(Co, k * k, Ci, device='cuda')
    fl = 2.0 * M * Co * Ci * k * k
    line = '%-28s' % str((M, Ci, Co, k))
    for c in cfgs:
        os.environ.update(c)
        t = bench(lambda: ops.conv2d_wgrad(x, dy, dw, N, H, W, H, W, k, k, 1, p, d))
        for kk in c: os.environ.pop(kk)
        line += ' | %s %5.1fus %4.0fTF' % (','.join('%s=%s' % (a[11:], b) for a, b in c.items()) or 'base', t, fl / t / 1e6)
    print(line)
