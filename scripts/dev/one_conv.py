"""dev: one conv geometry launched repeatedly (for PMC passes).  args: N H W Cin Cout k pad dil [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
N, H, W, Ci, Co, k, p, d = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
M = N * H * W
x = torch.randn(M, Ci, device='cuda').to(BF)
w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
y = torch.empty(M, Co, dtype=BF, device='cuda')
for _ in range(3):
    ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps
print('conv M=%d Ci=%d Co=%d k=%d: %.1f us  %.0f TF/s' % (M, Ci, Co, k, t * 1e3, 2.0 * M * Co * Ci * k * k / t / 1e9))
