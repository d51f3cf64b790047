"""dev: isolated timings of the kernels that touch the per-channel statistic accumulators, in whichever tree this file
is run from (old fp32 or new fixed-point accumulators): BN train apply / backward reduce / backward apply and a 1x1
convolution with the statistics epilogue, on rotating (cache-cold) buffers.  usage: python scripts/dev/kbench.py"""
import sys, os
ROOT = os.environ.get('KB_ROOT') or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from regda_amd import ops
BF = torch.bfloat16
new = hasattr(ops, 'new_stats')
mk_stats = (lambda *s: ops.new_stats(*s)) if new else (lambda *s: torch.zeros(*s, device='cuda'))


def bench(fn, n=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R = 10
print('tree:', ROOT, 'fixed-point' if new else 'fp32', flush=True)
for (M, C, res) in [(16384, 256, 0), (16384, 1024, 1), (65536, 512, 1), (262144, 64, 0), (16384, 2048, 1)]:
    G = 2
    x = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    y = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    r = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    mk = [torch.zeros(M, C // 8, dtype=torch.uint8, device='cuda') for _ in range(R)]
    stats = mk_stats(G, 8, 2, C)
    ops.bn_stats(x[0][:M // 2], stats[0], M // 2, C); ops.bn_stats(x[0][M // 2:], stats[1], M // 2, C)
    mi = torch.zeros(G, 2, C, device='cuda')
    rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    sums = mk_stats(G, 8, 2, C)
    t1 = bench(lambda i: ops.bn_train_apply(x[i % R], stats, mi, rm, rv, nbt, gamma, beta, y[i % R], M, C, True, r[i % R] if res else None, None, 0, groups=G, relu_mask=mk[i % R]))
    t2 = bench(lambda i: ops.bn_bwd_apply(r[i % R], None, x[i % R], mi, gamma, sums, y[i % R], M, C, True, None, dgam, dbet, None, 0, groups=G, relu_mask=mk[i % R]))
    t3 = bench(lambda i: ops.bn_bwd_reduce(r[i % R], None, x[i % R], mi, sums, M, C, True, None, 0, groups=G, relu_mask=mk[i % R]))
    print('M=%-7d C=%-5d res=%d | train_apply %6.1fus | bwd_apply %6.1fus | bwd_reduce %6.1fus' % (M, C, res, t1, t2, t3), flush=True)
for (M, Ci, Co) in [(16384, 256, 1024), (16384, 1024, 256), (65536, 128, 512), (262144, 64, 256)]:
    N, H = 16, int((M // 16) ** 0.5)
    x = [torch.randn(M, Ci, device='cuda').to(BF) for _ in range(R)]
    w = (torch.randn(Co, 1, Ci, device='cuda') * 0.05).to(BF)
    y = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(R)]
    st = mk_stats(2, 8, 2, Co)
    t0 = bench(lambda i: ops.conv2d(x[i % R], w, y[i % R], N, H, H, H, H, 1, 1, 1, 0, 1, 0, None, None, 1))
    t1 = bench(lambda i: ops.conv2d(x[i % R], w, y[i % R], N, H, H, H, H, 1, 1, 1, 0, 1, 0, None, st, 2))
    print('conv1x1 M=%-7d %4d -> %-4d | plain %6.1fus | with statistics %6.1fus' % (M, Ci, Co, t0, t1), flush=True)
