"""dev: weight-gradient launches of the step (grouped as the step groups them) next to the vendor GEMM
(torch.matmul -> hipBLASLt) on dense problems of the same size: dW[Cout, K] = dY^T[Cout, M] x X[M, K]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16


def t_of(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (count in one grouped launch, N, H, W, Cin, Cout, k, pad, dil)
CASES = [(7, 16, 32, 32, 256, 256, 3, 1, 1), (14, 16, 32, 32, 1024, 256, 1, 0, 1), (14, 16, 32, 32, 256, 1024, 1, 0, 1),
         (1, 16, 32, 32, 2048, 512, 3, 1, 1), (2, 16, 32, 32, 512, 512, 3, 2, 2), (3, 16, 32, 32, 512, 2048, 1, 0, 1),
         (4, 16, 128, 128, 64, 64, 3, 1, 1), (3, 16, 128, 128, 64, 256, 1, 0, 1)]
for (cnt, N, H, W, Ci, Co, k, p, d) in CASES:
    M, K = N * H * W, Ci * k * k
    items, mats = [], []
    for _ in range(cnt):
        x = torch.randn(M, Ci, device='cuda').to(BF)
        dy = torch.randn(M, Co, device='cuda').to(BF)
        dw = torch.zeros(Co, k * k, Ci, device='cuda')
        items.append((x, dy, dw, N, H, W, H, W, k, k, 1, p, d))
        mats.append((dy, torch.randn(M, K, device='cuda').to(BF) if cnt * M * K * 2 < 3e9 else None))
    tc = t_of(lambda: ops.conv2d_wgrad_grouped(items))
    fl = 2.0 * M * Co * K * cnt
    outs = [torch.empty(Co, K, dtype=BF, device='cuda') for _ in range(cnt)]
    if mats[0][1] is not None:
        def mm():
            for (dy, a), o in zip(mats, outs):
                torch.matmul(dy.t(), a, out=o)
        tg = t_of(mm)
        print('x%-2d M=%6d K=%5d Cout=%4d  wgrad %.0f us %.0f TF/s | matmul %.0f us %.0f TF/s' %
              (cnt, M, K, Co, tc * 1e3, fl / tc / 1e9, tg * 1e3, fl / tg / 1e9), flush=True)
    else:
        print('x%-2d M=%6d K=%5d Cout=%4d  wgrad %.0f us %.0f TF/s' % (cnt, M, K, Co, tc * 1e3, fl / tc / 1e9), flush=True)
    del items, mats, outs
