"""dev: do a main-chain convolution and a weight-gradient kernel share the chip when they run on two streams?  Times a
layer-3 3x3 forward convolution (128 x 128 tiles, 96 KB of LDS) and a grouped weight-gradient launch alone, back to back
and on two streams at once.  usage (GPU box): bash scripts/tune.sh dev coresidency.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
N, H, W = 16, 32, 32
M = N * H * W


def conv_case(Ci, Co, k):
    x = torch.randn(M, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    y = torch.empty(M, Co, dtype=BF, device='cuda')
    st = ops.new_stats(2, 8, 2, Co)
    return lambda: ops.conv2d(x, w, y, N, H, W, H, W, k, k, 1, k // 2, 1, 0, None, st, 2)


def wgrad_case(Ci, Co, k, layers):
    items = []
    for _ in range(layers):
        x = torch.randn(M, Ci, device='cuda').to(BF)
        dy = torch.randn(M, Co, device='cuda').to(BF)
        dw = torch.zeros(Co, k * k, Ci, device='cuda')
        items.append((x, dy, dw, N, H, W, H, W, k, k, 1, k // 2, 1))
    return lambda: ops.conv2d_wgrad_grouped(items)


def timed(fa, fb, na, nb, mode):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run():
        if mode == 'serial':
            for _ in range(na): fa()
            for _ in range(nb): fb()
        else:
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with ops.use_stream(s1):
                for _ in range(na): fa()
            with ops.use_stream(s2):
                for _ in range(nb): fb()
            cur.wait_stream(s1); cur.wait_stream(s2)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3


def bn_case(C):
    x = torch.randn(M, C, device='cuda').to(BF); g = torch.randn(M, C, device='cuda').to(BF)
    dx = torch.empty(M, C, dtype=BF, device='cuda')
    mk = torch.zeros(M, C // 8, dtype=torch.uint8, device='cuda')
    mi = torch.zeros(2, 2, C, device='cuda'); mi[:, 1] = 1.0
    gamma = torch.ones(C, device='cuda'); sums = ops.new_stats(2, 8, 2, C)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    return lambda: ops.bn_bwd_apply(g, None, x, mi, gamma, sums, dx, M, C, True, None, dg, db, None, 0, groups=2, relu_mask=mk)


# HBM-bound BatchNorm-backward passes (114 registers per wave) next to kernels that leave / do not leave them registers
for bname, bn, nb_ in (('bn_bwd_apply 1024 ch', bn_case(1024), 20), ('bn_bwd_apply 256 ch', bn_case(256), 40)):
    for wname, wg, nw in (('conv 3x3 256->256 (2 waves x 96 regs / SIMD)', conv_case(256, 256, 3), 20),
                          ('wgrad 1x1 x14 (2 waves x 216 regs / SIMD)', wgrad_case(1024, 256, 1, 14), 2),
                          ('wgrad 3x3 x7 (2 waves x 256 regs / SIMD)', wgrad_case(256, 256, 3, 7), 2)):
        ta = timed(bn, lambda: None, nb_, 0, 'serial')
        tb = timed(lambda: None, wg, 0, nw, 'serial')
        tc = timed(bn, wg, nb_, nw, 'concurrent')
        print('%-22s x%d %7.1f us | %-46s x%d %7.1f us | sum %7.1f | two streams %7.1f us (%.0f %% of the sum, %.0f %% of the longer)' % (
            bname, nb_, ta, wname, nw, tb, ta + tb, tc, 100 * tc / (ta + tb), 100 * tc / max(ta, tb)), flush=True)

for cname, conv in (('3x3 256->256 (128x128 PIPE, 96 KB)', conv_case(256, 256, 3)), ('1x1 256->1024 (128x128 2-stage, 64 KB)', conv_case(256, 1024, 1)),
                    ('1x1 1024->256 (PIPE, 96 KB)', conv_case(1024, 256, 1))):
    for wname, wg, nb in (('tap-fused 3x3 x7 (56 KB, 2/CU)', wgrad_case(256, 256, 3, 7), 2), ('1x1 1024->256 x14 (256x128 tiles, 144 KB)', wgrad_case(1024, 256, 1, 14), 2)):
        na = 20
        ta = timed(conv, lambda: None, na, 0, 'serial')
        tb = timed(lambda: None, wg, 0, nb, 'serial')
        tc = timed(conv, wg, na, nb, 'concurrent')
        print('%-42s x%d %7.1f us | %-44s x%d %7.1f us | sum %7.1f | two streams %7.1f us (%.0f %% of the sum)' % (
            cname, na, ta, wname, nb, tb, ta + tb, tc, 100 * tc / (ta + tb)), flush=True)
