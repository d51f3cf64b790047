"""Per-workgroup phase timestamps of convolutions (tuning build only: make TUNING=1).
argv: nothing = the HBM-bound small-K 1x1 shapes; `l3` = the 32 x 32-map shapes of layer 3 / the head;
`one N H W Ci Co k` = that shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
shapes = [(16, 128, 128, 64, 256, 1), (16, 128, 128, 256, 64, 1), (16, 64, 64, 128, 512, 1), (16, 64, 64, 512, 128, 1), (16, 64, 64, 1024, 256, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'l3':
    shapes = [(16, 32, 32, 256, 256, 3), (16, 32, 32, 256, 1024, 1), (16, 32, 32, 1024, 256, 1), (8, 32, 32, 256, 256, 3),
              (8, 32, 32, 256, 1024, 1), (16, 32, 32, 2048, 512, 3)]
if len(sys.argv) > 1 and sys.argv[1] == 'one':        # one N H W Ci Co k
    shapes = [tuple(int(v) for v in sys.argv[2:8])]
for (N, H, W, Ci, Co, k) in shapes:
    M = N * H * W
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    st = ops.new_stats(8, 2, Co)
    def go(): ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, k // 2, 1, 0, None, st)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    dbg = torch.zeros(8192 * 16, dtype=torch.int64, device='cuda')
    os.environ['RGDA_CONV_DBG'] = str(dbg.data_ptr())
    go(); torch.cuda.synchronize()
    os.environ.pop('RGDA_CONV_DBG')
    allv = dbg.view(-1, 4).cpu()
    nw = int((allv[:8192, 2] > 0).sum())
    t = allv[:nw]
    d1, d2, d3 = (t[:, 1] - t[:, 0]).float(), (t[:, 2] - t[:, 1]).float(), (t[:, 3] - t[:, 2]).float()
    ep = allv[16384:16384 + nw]
    if (ep[:, 0] > 0).any():
        e1, e2, e3 = (ep[:, 0] - t[:, 2]).float().median(), (ep[:, 1] - ep[:, 0]).float().median(), (ep[:, 2] - ep[:, 1]).float().median()
        print('   epilogue split: acc->LDS+sync %.0f | rows->memory %.0f | stats reduce+sync %.0f | atomics issue %.0f' %
              (e1, e2, e3, (t[:, 3] - ep[:, 2]).float().median()))
    it = allv[24576:24576 + nw]
    if (it[:, 0] > 0).any():
        print('   row-loop iterations (end of each, from the loop start):', [int((it[:, i] - ep[:, 0]).float().median()) for i in range(4)])
    span = (t[:, 3].max() - t[:, 0].min()).item()
    byt = (M * Ci + M * Co) * 2
    print((M, Ci, Co, k), 'wgs', nw, '%.1f us  %.2f TB/s | first-tile wait %.0f  kloop %.0f  epilogue %.0f  total/WG %.0f | span %d ticks; WG-lifetime sum / span = %.2f concurrent WGs' % (
        us, byt / us / 1e6, d1.median(), d2.median(), d3.median(), (t[:, 3] - t[:, 0]).float().median(), span, (t[:, 3] - t[:, 0]).float().sum().item() / span))
