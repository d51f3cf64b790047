"""dev: does an L2/MFMA-bound convolution overlap with an HBM-bound BatchNorm pass on another stream?
Times N launches of each alone and both concurrently (the launches are long enough for the host to keep up)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
dev = 'cuda'
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, H, W = 16 * scale, 32, 32
M = N * H * W
reps = 60
x = torch.randn(M, 256, device=dev).to(BF); w = (torch.randn(256, 9, 256, device=dev) * .05).to(BF)
y = torch.empty(M, 256, dtype=BF, device=dev)
x2 = torch.randn(M // 2, 256, device=dev).to(BF); y2 = torch.empty(M // 2, 256, dtype=BF, device=dev)
c = torch.randn(M, 1024, device=dev).to(BF); r = torch.randn(M, 1024, device=dev).to(BF); o = torch.empty_like(c)
mi = torch.cat([torch.zeros(1024), torch.ones(1024)]).to(dev); ga = torch.ones(1024, device=dev); be = torch.zeros(1024, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def conv(): ops.conv2d(x, w, y, N, H, W, H, W, 3, 3, 1, 1, 1, 0)
def convh(): ops.conv2d(x2, w, y2, N // 2, H, W, H, W, 3, 3, 1, 1, 1, 0)
def bn(): ops.bn_apply(c, mi, ga, be, o, M, 1024, True, r)


def run(fa, fb, na=reps, nb=reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    for i in range(max(na, nb)):
        if fa is not None and i < na:
            with torch.cuda.stream(sa): fa()
        if fb is not None and i < nb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(2):
    run(conv, bn, 5, 5)
ta, tb, tab = run(conv, None), run(None, bn), run(conv, bn)
print('M=%d  conv alone %.0f us/launch, bn alone %.0f, both concurrently %.0f per pair (sum %.0f, max %.0f)' %
      (M, ta / reps, tb / reps, tab / reps, (ta + tb) / reps, max(ta, tb) / reps))
tcc = run(conv, conv)
print('conv || conv: %.0f per pair (2 x alone = %.0f)' % (tcc / reps, 2 * ta / reps))
th = run(convh, None)
thh = run(convh, convh)
print('half-M conv alone %.0f; two half convs concurrently %.0f per pair; full conv %.0f' % (th / reps, thh / reps, ta / reps))
tbb = run(bn, bn)
print('bn || bn: %.0f per pair (2 x alone = %.0f)' % (tbb / reps, 2 * tb / reps))
