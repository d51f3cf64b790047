"""dev: what bounds the large BatchNorm element-wise passes (bn3 + residual + ReLU forward, its backward apply)?
Isolated timings on rotating buffers of  rgda_bn_train_apply  (statistics rebuilt per workgroup),  rgda_bn_apply
(mean / invstd given: no rebuild) and a plain torch element-wise add of the same shape (2 reads + 1 write), per layer
geometry of the 8 + 8 step; with the tuning library the grid hooks RGDA_BN_ROWS / RGDA_BN_VPB are swept.
usage: bash scripts/tune.sh dev bn_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from regda_amd import ops
BF = torch.bfloat16


def bench(fn, n=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R = 12
for (M, C) in [(16384, 1024), (16384, 2048), (65536, 512), (262144, 256)]:
    G = 2
    x = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    y = [torch.empty(M, C, dtype=BF, device='cuda') for _ in range(R)]
    r = [torch.randn(M, C, device='cuda').to(BF) for _ in range(R)]
    stats = ops.new_stats(G, 8, 2, C)
    ops.bn_stats(x[0][:M // 2], stats[0], M // 2, C); ops.bn_stats(x[0][M // 2:], stats[1], M // 2, C)
    mi = torch.zeros(G, 2, C, device='cuda')
    rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(1, dtype=torch.int64, device='cuda')
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    sums = ops.new_stats(G, 8, 2, C)
    mb = M * C * 2 * 3 / 1e6
    t_add = bench(lambda i: torch.add(x[i % R], r[i % R], out=y[i % R]))
    t_cp = bench(lambda i: y[i % R].copy_(x[i % R]))
    print('M=%-7d C=%-5d %.0f MB | torch add %6.1f us (%.2f TB/s) | torch copy %6.1f us (%.2f TB/s)' %
          (M, C, mb, t_add, mb / t_add, t_cp, mb * 2 / 3 / t_cp), flush=True)
    wide = [{'RGDA_BN_VPB': str(v), 'RGDA_BN_ROWS': str(r)} for v in (32, 64, 128, 256) if v * 8 <= C for r in (4, 8, 16)]
    for env in [{}, {'RGDA_BN_ROWS': '4'}, {'RGDA_BN_ROWS': '16'}, {'RGDA_BN_ROWS': '32'}] + wide:
        for k in ('RGDA_BN_ROWS', 'RGDA_BN_VPB'): os.environ.pop(k, None)
        os.environ.update(env)
        if int(env.get('RGDA_BN_VPB', 16)) > 16:       # no statistics prologue above 128 channels per workgroup
            t0 = bench(lambda i: ops.bn_apply(x[i % R], mi, gamma, beta, y[i % R], M, C, True, r[i % R], None, 0, groups=G))
            print('   %-48s apply(mi) %6.1f us (%.2f TB/s)' % (env, t0, mb / t0), flush=True)
            continue
        t1 = bench(lambda i: ops.bn_train_apply(x[i % R], stats, mi, rm, rv, nbt, gamma, beta, y[i % R], M, C, True, r[i % R], None, 0, groups=G))
        t0 = bench(lambda i: ops.bn_apply(x[i % R], mi, gamma, beta, y[i % R], M, C, True, r[i % R], None, 0, groups=G))
        t2 = bench(lambda i: ops.bn_bwd_apply(r[i % R], None, x[i % R], mi, gamma, sums, y[i % R], M, C, False, None, dgam, dbet, None, 0, groups=G))
        print('   %-40s train_apply %6.1f us (%.2f TB/s) | apply(mi) %6.1f us (%.2f TB/s) | bwd_apply<false> %6.1f us' %
              (env, t1, mb / t1, t0, mb / t0, t2), flush=True)
