import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
N, HW, C = 16, 1024, 512
hid = torch.randn(N * HW, C, device='cuda').to(BF)
w = torch.randn(6, C, device='cuda') * 0.05
gl = torch.randn(N, 6, 32, 32, device='cuda')
dh = torch.empty(N * HW, C, dtype=BF, device='cuda')
dw, db = torch.zeros(6, C, device='cuda'), torch.zeros(6, device='cuda')
for rows in ('64', '128', '256', '512', '1024'):
    os.environ['RGDA_CLS_ROWS'] = rows
    for _ in range(3): ops.classifier_bwd(hid, w, gl, dh, dw, db, N, HW, C, 6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.classifier_bwd(hid, w, gl, dh, dw, db, N, HW, C, 6)
    e1.record(); torch.cuda.synchronize()
    print('rows/block', rows, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
