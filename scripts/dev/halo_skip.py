"""dev: K-loop ablation of the pipelined halo kernel (tuning library; RGDA_CONV_SKIP bits: 1 no DMA in the loop, 2 no vmcnt
wait, 4 no barrier -- results are garbage, timings only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
GEOMS = [(16, 32, 32, 2048, 512, 1, 'head 2048->512'), (16, 32, 32, 256, 256, 1, 'layer3 256->256')]
reps = 30
for N, H, W, Ci, Co, d, name in GEOMS:
    M = N * H * W
    xs = [torch.randn(M, Ci, device='cuda').to(BF) for _ in range(4)]
    w = (torch.randn(Co, 9, Ci, device='cuda') * 0.05).to(BF)
    ys = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(4)]
    for rnd in range(2):
      for skip in (0, 1, 2, 4, 3, 7):
        os.environ['RGDA_CONV_SKIP'] = str(skip)
        for i in range(3):
            ops.conv2d(xs[i % 4], w, ys[i % 4], N, H, W, H, W, 3, 3, 1, d, d, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ops.conv2d(xs[i % 4], w, ys[i % 4], N, H, W, H, W, 3, 3, 1, d, d, 0)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print('%-18s skip %d: %.1f us %.0f TF/s' % (name, skip, t * 1e3, 2.0 * M * Co * Ci * 9 / t / 1e9), flush=True)
