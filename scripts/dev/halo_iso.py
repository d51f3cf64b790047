"""dev: isolated timings of the halo-kernel geometries with the library that is in place (no environment switches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
GEOMS = [(16, 32, 32, 2048, 512, 1, 'head 2048->512'), (16, 32, 32, 512, 512, 1, 'layer4 512->512'),
         (16, 32, 32, 256, 256, 1, 'layer3 256->256'), (8, 32, 32, 2048, 512, 1, 'teacher head'), (16, 32, 32, 512, 512, 2, 'layer4 dil2')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sparse = len(sys.argv) > 2
for N, H, W, Ci, Co, d, name in GEOMS:
    M = N * H * W
    xs = [torch.randn(M, Ci, device='cuda') for _ in range(4)]
    if sparse:
        xs = [x.relu() for x in xs]
    xs = [x.to(BF) for x in xs]
    w = (torch.randn(Co, 9, Ci, device='cuda') * 0.05).to(BF)
    ys = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(4)]
    for mode in (0, 1):
        for i in range(3):
            ops.conv2d(xs[i % 4], w, ys[i % 4], N, H, W, H, W, 3, 3, 1, d, d, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ops.conv2d(xs[i % 4], w, ys[i % 4], N, H, W, H, W, 3, 3, 1, d, d, mode)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print('%-18s mode %d %s %.1f us %.0f TF/s' % (name, mode, 'relu-x' if sparse else 'randn', t * 1e3, 2.0 * M * Co * Ci * 9 / t / 1e9), flush=True)
