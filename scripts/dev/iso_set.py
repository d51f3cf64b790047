"""dev: isolated timings of a fixed set of convolution / weight-gradient launches with the library that is in place
(for A/B of two builds on one box: run it once per library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16


def t_of(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# forward convolutions: (N, H, W, Cin, Cout, k, pad, dil, mode, name)
CONVS = [(16, 32, 32, 1024, 256, 1, 0, 1, 0, '1x1 1024->256'), (16, 32, 32, 2048, 512, 1, 0, 1, 0, '1x1 2048->512'),
         (16, 32, 32, 256, 1024, 1, 0, 1, 0, '1x1 256->1024'), (8, 32, 32, 1024, 256, 1, 0, 1, 0, 'teacher 1x1 1024->256'),
         (16, 32, 32, 2048, 512, 3, 1, 1, 0, 'head 3x3'), (16, 32, 32, 256, 256, 3, 1, 1, 0, 'layer3 3x3'),
         (16, 32, 32, 512, 512, 3, 2, 2, 0, 'layer4 3x3 d2'), (16, 64, 64, 128, 128, 3, 1, 1, 0, 'layer2 3x3')]
for N, H, W, Ci, Co, k, p, d, mode, name in CONVS:
    M = N * H * W
    xs = [torch.randn(M, Ci, device='cuda').relu().to(BF) for _ in range(4)]
    w = (torch.randn(Co, k * k, Ci, device='cuda') * 0.05).to(BF)
    ys = [torch.empty(M, Co, dtype=BF, device='cuda') for _ in range(4)]
    it = [0]
    def f():
        i = it[0] % 4; it[0] += 1
        ops.conv2d(xs[i], w, ys[i], N, H, W, H, W, k, k, 1, p, d, mode)
    t = t_of(f)
    print('conv  %-24s %7.1f us %6.0f TF/s' % (name, t, 2.0 * M * Co * Ci * k * k / t / 1e6), flush=True)
WG = [(7, 16, 32, 32, 256, 256, 3, 1, 1, 'x7 3x3 256'), (1, 16, 32, 32, 2048, 512, 3, 1, 1, 'head 3x3'),
      (14, 16, 32, 32, 1024, 256, 1, 0, 1, 'x14 1x1 1024->256'), (14, 16, 32, 32, 256, 1024, 1, 0, 1, 'x14 1x1 256->1024'),
      (2, 16, 32, 32, 512, 512, 3, 2, 2, 'x2 3x3 512 d2'), (3, 16, 64, 64, 128, 128, 3, 1, 1, 'x3 3x3 128 (64 maps)')]
for cnt, N, H, W, Ci, Co, k, p, d, name in WG:
    M = N * H * W
    items = []
    for _ in range(cnt):
        x = torch.randn(M, Ci, device='cuda').relu().to(BF)
        dy = torch.randn(M, Co, device='cuda').to(BF)
        dw = torch.zeros(Co, k * k, Ci, device='cuda')
        items.append((x, dy, dw, N, H, W, H, W, k, k, 1, p, d))
    t = t_of(lambda: ops.conv2d_wgrad_grouped(items), 15)
    print('wgrad %-24s %7.1f us %6.0f TF/s' % (name, t, 2.0 * M * Co * Ci * k * k * cnt / t / 1e6), flush=True)
    del items
