"""dev: steady-state rate of the wgrad kernels when tiles are plentiful (what a grouped launch would see)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
def bench(fn, n=10):
    for i in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Ci, Co, k, p, d) in [(16, 32, 32, 2048, 4096, 1, 0, 1), (16, 32, 32, 1024, 1024, 3, 1, 1), (16, 32, 32, 1024, 2048, 3, 1, 1)]:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    dy = torch.randn(M, Co, device='cuda').to(BF)
    dw = torch.zeros(Co, k * k, Ci, device='cuda')
    fl = 2.0 * M * Co * Ci * k * k
    line = '%-30s' % str((M, Ci, Co, k))
    for sp in ('1', '2', '4', '16'):
        os.environ['RGDA_WGRAD_SPLITS'] = sp
        t = bench(lambda: ops.conv2d_wgrad(x, dy, dw, N, H, W, H, W, k, k, 1, p, d))
        line += ' | splits %s %7.1fus %4.0fTF' % (sp, t, fl / t / 1e6)
    os.environ.pop('RGDA_WGRAD_SPLITS')
    print(line)
