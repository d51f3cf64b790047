"""dev: per-workgroup phase timestamps of the generic wgrad kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
for (N, H, W, Ci, Co, k, p, d) in [(16, 32, 32, 256, 1024, 1, 0, 1), (16, 32, 32, 512, 2048, 1, 0, 1)]:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    dy = torch.randn(M, Co, device='cuda').to(BF)
    dw = torch.zeros(Co, k * k, Ci, device='cuda')
    for epi in ('8w', '4w'):
        os.environ.pop('RGDA_WGRAD_W4', None)
        if epi == '4w': os.environ['RGDA_WGRAD_W4'] = '1'
        for _ in range(3): ops.conv2d_wgrad(x, dy, dw, N, H, W, H, W, k, k, 1, p, d)
        torch.cuda.synchronize()
        dbg = torch.zeros(4096 * 4, dtype=torch.int64, device='cuda')
        os.environ['RGDA_CONV_DBG'] = str(dbg.data_ptr())
        ops.conv2d_wgrad(x, dy, dw, N, H, W, H, W, k, k, 1, p, d)
        torch.cuda.synchronize()
        os.environ.pop('RGDA_CONV_DBG')
        tt = dbg.view(-1, 4).cpu()
        tt = tt[tt[:, 3] > 0]
        t0 = tt[:, 0].min()
        f = lambda v: '%.0f/%.0f/%.0f' % (v.float().min(), v.float().median(), v.float().max())
        print((M, Ci, Co, k), 'epi', epi, 'wgs', len(tt), 'start skew', f(tt[:, 0] - t0), ' first', f(tt[:, 1] - tt[:, 0]), ' loop', f(tt[:, 2] - tt[:, 1]),
              ' epilogue', f(tt[:, 3] - tt[:, 2]), ' span', int(tt[:, 3].max() - t0))
