import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
shapes = [(16, 32, 32, 256, 256, 3, 1, 1), (16, 32, 32, 4096, 512, 3, 1, 1)]
for (N, H, W, Ci, Co, k, p, d) in shapes:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    for tile in (sys.argv[1:] or ['128,64,3']):
      for skip in ('0', '1', '2', '3'):
        os.environ['RGDA_TILE'] = tile
        os.environ['RGDA_CONV_SKIP'] = skip
        dbg = torch.zeros(8192 * 8, dtype=torch.int64, device='cuda')
        for _ in range(3):
            ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)
        torch.cuda.synchronize()
        os.environ['RGDA_CONV_DBG'] = str(dbg.data_ptr())
        ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)
        torch.cuda.synchronize()
        os.environ.pop('RGDA_CONV_DBG')
        allv = dbg.view(-1, 4).cpu()
        nw = int((allv[:, 2] > 0).sum())
        t = allv[:nw]
        tw = allv[nw:2 * nw]
        d1, d2, d3 = (t[:, 1] - t[:, 0]).float(), (t[:, 2] - t[:, 1]).float(), (t[:, 3] - t[:, 2]).float()
        span = (t[:, 3].max() - t[:, 0].min()).item()
        print((M, Ci, Co, k), tile, 'skip', skip, 'wgs', len(t), 'first-tile wait %.0f  kloop %.0f  epilogue %.0f  total/WG %.0f | kernel span %d (counter ticks)' % (d1.median(), d2.median(), d3.median(), (t[:, 3] - t[:, 0]).float().median(), span), ' vmcnt-wait %.0f barrier %.0f' % (tw[:, 0].float().median(), tw[:, 1].float().median()))
