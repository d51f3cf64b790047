"""dev: a few launches of one wgrad / conv shape (for PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
N, H, W, Ci, Co, k, p, d = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else (16, 32, 32, 256, 1024, 1, 0, 1))]
M = N * H * W
x = torch.randn(M, Ci, device='cuda').to(BF)
dy = torch.randn(M, Co, device='cuda').to(BF)
dw = torch.zeros(Co, k * k, Ci, device='cuda')
w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
y = torch.empty(M, Co, dtype=BF, device='cuda')
for _ in range(5):
    ops.conv2d_wgrad(x, dy, dw, N, H, W, H, W, k, k, 1, p, d)
    ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)
torch.cuda.synchronize()
