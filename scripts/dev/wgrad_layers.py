"""dev: single-layer and grouped timings of the layer-2 weight gradients that share conv_wgrad_kernel<128, 128> (and the
small-channel tap-fused 3x3 layers), isolated, back to back."""
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


def item(N, H, W, Ci, Co, k, s, p, d):
    Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    x = torch.randn(N * H * W, Ci, device='cuda').to(BF)
    dy = torch.randn(N * Ho * Wo, Co, device='cuda').to(BF)
    dw = torch.zeros(Co, k * k, Ci, device='cuda')
    return (x, dy, dw, N, H, W, Ho, Wo, k, k, s, p, d), 2.0 * N * Ho * Wo * Co * k * k * Ci


LAYERS = {'l2 conv1 256->128': (16, 64, 64, 256, 128, 1, 1, 0, 1), 'l2 conv1 512->128': (16, 64, 64, 512, 128, 1, 1, 0, 1),
          'l2.0 conv2 3x3 s2 128->128': (16, 128, 128, 128, 128, 3, 2, 1, 1), 'l2 conv2 3x3 128->128': (16, 64, 64, 128, 128, 3, 1, 1, 1),
          'l1 conv2 3x3 64->64': (16, 128, 128, 64, 64, 3, 1, 1, 1), 'l1 conv1 256->64': (16, 128, 128, 256, 64, 1, 1, 0, 1),
          'l1 conv3 64->256': (16, 128, 128, 64, 256, 1, 1, 0, 1), 'l2 conv3 128->512': (16, 64, 64, 128, 512, 1, 1, 0, 1)}
for name, sh in LAYERS.items():
    it, fl = item(*sh)
    t = t_of(lambda: ops.conv2d_wgrad_grouped([it]))
    print('%-30s %7.1f us  %6.0f TF/s' % (name, t * 1e3, fl / t / 1e9), flush=True)
grp = [item(*LAYERS['l2 conv1 256->128']), item(*LAYERS['l2 conv1 512->128']), item(*LAYERS['l2 conv1 512->128']),
       item(*LAYERS['l2 conv1 512->128']), item(*LAYERS['l2.0 conv2 3x3 s2 128->128'])]
t = t_of(lambda: ops.conv2d_wgrad_grouped([g[0] for g in grp]))
print('grouped <128,128> family as in the step: %.1f us  %.0f TF/s' % (t * 1e3, sum(g[1] for g in grp) / t / 1e9))
grp = [item(*LAYERS['l2 conv1 256->128'])] + [item(*LAYERS['l2 conv1 512->128']) for _ in range(3)]
t = t_of(lambda: ops.conv2d_wgrad_grouped([g[0] for g in grp]))
print('  ... the four 1x1 layers only: %.1f us  %.0f TF/s' % (t * 1e3, sum(g[1] for g in grp) / t / 1e9))
