"""dev: what the vendor GEMM (torch.matmul -> hipBLASLt, bf16) reaches on the GEMM shapes of the step's convolutions,
next to this library's implicit-GEMM kernel on the same shape (calibration of what the chip gives, not a product path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16


def t_of(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (N, H, W, Cin, Cout, k, pad, dil)
CASES = [(16, 32, 32, 256, 256, 3, 1, 1), (16, 32, 32, 256, 1024, 1, 0, 1), (16, 32, 32, 1024, 256, 1, 0, 1),
         (16, 32, 32, 2048, 512, 3, 1, 1), (16, 32, 32, 512, 512, 3, 2, 2), (16, 32, 32, 512, 2048, 1, 0, 1),
         (16, 64, 64, 128, 512, 1, 0, 1), (16, 128, 128, 64, 256, 1, 0, 1)]
for (N, H, W, Ci, Co, k, p, d) in CASES:
    M, K = N * H * W, Ci * k * k
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    tc = t_of(lambda: ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, p, d, 0))
    a = torch.randn(M, K, device='cuda').to(BF)
    b = w.reshape(Co, K)
    o = torch.empty(M, Co, dtype=BF, device='cuda')
    tg = t_of(lambda: torch.matmul(a, b.t(), out=o))
    fl = 2.0 * M * Co * K
    print('M=%6d K=%5d N=%4d  conv %.1f us %.0f TF/s | matmul %.1f us %.0f TF/s' % (M, K, Co, tc * 1e3, fl / tc / 1e9, tg * 1e3, fl / tg / 1e9), flush=True)
