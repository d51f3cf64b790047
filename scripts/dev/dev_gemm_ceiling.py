"""dev calibration only: what the vendor GEMM (hipBLASLt through torch.matmul) reaches on the GEMM shapes our
implicit-GEMM convs correspond to.  Not used by the product."""
import torch
BF = torch.bfloat16
def bench(fn, n=20):
    for i in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(16384, 256, 2304), (16384, 512, 36864), (16384, 256, 1024), (16384, 1024, 256), (16384, 512, 4608), (65536, 128, 1152), (8192, 256, 2304), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device='cuda').to(BF)
    b = torch.randn(N, K, device='cuda').to(BF)
    c = torch.empty(M, N, device='cuda', dtype=BF)
    t = bench(lambda: torch.matmul(a, b.t(), out=c))
    print('M=%-6d N=%-5d K=%-6d %8.1f us %7.0f TF/s' % (M, N, K, t, 2.0 * M * N * K / t / 1e6))
