import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from regda_amd import ops
BF = torch.bfloat16
def to_pxc(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(-1, c).to(BF).cuda().contiguous()
def from_pxc(t, n, h, w):
    return t.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2)
def rbf(x): return x.to(BF).float()
def l2(a, b): return ((a - b).norm() / b.norm()).item()
for case in [(4, 8, 8, 4096, 512, 3, 1, 1, 1), (4, 8, 8, 512, 512, 3, 1, 2, 2), (4, 8, 8, 2048, 512, 1, 1, 0, 1), (2, 16, 16, 256, 256, 3, 2, 1, 1)]:
    N, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(1)
    x = rbf(torch.randn(N, Cin, H, W, generator=g))
    w = rbf(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, p, d)
    dy = rbf(torch.randn(N, Cout, Ho, Wo, generator=g))
    ref.backward(dy)
    xg, dyg = to_pxc(x), to_pxc(dy)
    wg = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin).to(BF).cuda().contiguous()
    wt = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).to(BF).cuda().contiguous()
    y = torch.zeros(N * Ho * Wo, Cout, dtype=BF, device='cuda')
    ops.conv2d(xg, wg, y, N, H, W, Ho, Wo, k, k, s, p, d, 0)
    dx = torch.zeros(N * H * W, Cin, dtype=BF, device='cuda')
    ops.conv2d(dyg, wt, dx, N, Ho, Wo, H, W, k, k, s, p, d, 1)
    dw = torch.zeros(Cout, k * k, Cin, device='cuda')
    ops.conv2d_wgrad(xg, dyg, dw, N, H, W, Ho, Wo, k, k, s, p, d)
    print(case, 'fwd', l2(from_pxc(y, N, Ho, Wo), ref.detach()), 'dgrad', l2(from_pxc(dx, N, H, W), xr.grad),
          'wgrad', l2(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)))
