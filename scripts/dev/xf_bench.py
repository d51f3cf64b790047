"""dev: BatchNorm + ReLU on the operand path against the materialised route, isolated, rotating (cache-cold) buffers:
   bn_train_apply + conv2d(stats)   vs   conv2d_bnin(stats)      for the bottleneck geometries of the step.
With the tuning library (bash scripts/tune.sh dev xf_bench.py) RGDA_CONV_SKIP=256 drops the statistics -> table prologue,
512 the in-loop transform (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16


def bench(fn, n=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R, G = 8, 2
print('RGDA_CONV_SKIP =', os.environ.get('RGDA_CONV_SKIP'), flush=True)
for (N, H, Ci, Co, k, s, p) in [(16, 32, 256, 256, 3, 1, 1), (16, 32, 256, 1024, 1, 1, 0), (16, 32, 512, 512, 3, 1, 1), (16, 32, 512, 2048, 1, 1, 0),
                                (16, 64, 128, 128, 3, 1, 1), (16, 64, 128, 512, 1, 1, 0), (16, 128, 64, 256, 1, 1, 0)]:
    Ho = (H + 2 * p - (k - 1) - 1) // s + 1
    Min, M = N * H * H, N * Ho * Ho
    c = [torch.randn(Min, Ci, device='cuda').to(BF) for _ in range(R)]
    a = [torch.empty(Min, Ci, dtype=BF, device='cuda') for _ in range(R)]
    mk = [torch.empty(Min, Ci // 8, dtype=torch.uint8, device='cuda') for _ in range(R)]
    y = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(R)]
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    pst = ops.new_stats(G, 8, 2, Ci)
    for g in range(G):
        ops.bn_stats(c[0][g * Min // G:(g + 1) * Min // G], pst[g], Min // G, Ci)
    mi = torch.zeros(G, 2, Ci, device='cuda')
    rm, rv, nbt = torch.zeros(Ci, device='cuda'), torch.ones(Ci, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    gamma, beta = torch.ones(Ci, device='cuda'), torch.zeros(Ci, device='cuda')
    st = ops.new_stats(G, 8, 2, Co)
    bnop = ops.bn_operand(pst, gamma, beta, mi, rm, rv, nbt, G, True)

    def mat(i):
        ops.bn_train_apply(c[i % R], pst, mi, rm, rv, nbt, gamma, beta, a[i % R], Min, Ci, True, None, None, 0, groups=G, relu_mask=mk[i % R])
        ops.conv2d(a[i % R], w, y[i % R], N, H, H, Ho, Ho, k, k, s, p, 1, 0, None, st, G)
    t_apply = bench(lambda i: ops.bn_train_apply(c[i % R], pst, mi, rm, rv, nbt, gamma, beta, a[i % R], Min, Ci, True, None, None, 0, groups=G, relu_mask=mk[i % R]))
    t_conv = bench(lambda i: ops.conv2d(a[i % R], w, y[i % R], N, H, H, Ho, Ho, k, k, s, p, 1, 0, None, st, G))
    t_mat = bench(mat)
    t_xf = bench(lambda i: ops.conv2d_bnin(bnop, c[i % R], w, y[i % R], N, H, H, Ho, Ho, k, k, s, p, 1, None, st, G))
    print('%2d x %3d^2 %4d -> %-4d k%d | apply %5.1f + conv %5.1f = %5.1f (back to back %5.1f) | operand path %5.1f us' %
          (N, H, Ci, Co, k, t_apply, t_conv, t_apply + t_conv, t_mat, t_xf), flush=True)
