"""dev: is the max-norm error of rgda_bn_bwd_apply(relu = 2) on the 16 x 128 x 128 x 128 case a ReLU-sign tie?  Per seed:
the raw max-norm error, the error with the near-zero pre-activations masked, and the pre-activation at the worst element.
usage: python scripts/dev/bnin_tie_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from regda_amd import ops
from test_conv_gpu import to_pxc, from_pxc, rbf, relerr, _bn_ref
N, H, W, Cin, G = 16, 128, 128, 128, 2
for seed in range(12):
    gen = torch.Generator().manual_seed(seed)
    c = rbf(torch.randn(N, Cin, H, W, generator=gen) * (0.5 + torch.rand(1, Cin, 1, 1, generator=gen)) + torch.randn(1, Cin, 1, 1, generator=gen))
    gamma = 0.5 + torch.rand(Cin, generator=gen); gamma[::7] *= -1.0
    beta = 0.3 * torch.randn(Cin, generator=gen)
    bnr, mean_ref, var_ref = _bn_ref(c, gamma, beta, G)
    cg = to_pxc(c)
    mi = torch.stack([mean_ref.float(), (1.0 / torch.sqrt(var_ref + 1e-5)).float()], 1).cuda().contiguous()
    gam, bet = gamma.cuda(), beta.cuda()
    gact = rbf(torch.randn(N, Cin, H, W, generator=gen))
    cr = c.clone().requires_grad_(True)
    outs = [F.relu(F.batch_norm(cgrp, None, None, gamma, beta, True, 0.1, 1e-5)) for cgrp in cr.chunk(G, 0)]
    torch.cat(outs, 0).backward(gact)
    gg = to_pxc(gact)
    sums = ops.new_stats(G, 8, 2, Cin)
    ops.bn_bwd_reduce(gg, None, cg, mi, sums, N * H * W, Cin, 2, groups=G, gamma=gam, beta=bet)
    dcx = torch.zeros(N * H * W, Cin, dtype=torch.bfloat16, device='cuda')
    dgam, dbet = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    ops.bn_bwd_apply(gg, None, cg, mi, gam, sums, dcx, N * H * W, Cin, 2, None, dgam, dbet, groups=G, beta=bet)
    d = from_pxc(dcx, N, H, W)
    err = (d - cr.grad).abs()
    i = err.argmax()
    tie = bnr.abs() < 2e-6 * (1.0 + c.abs() * gamma.abs().view(1, -1, 1, 1))
    print('seed %2d raw %.4f masked %.4f | worst element: pre-activation %.3e g %.3f | ties %d' %
          (seed, relerr(d, cr.grad), relerr(torch.where(tie, 0., d), torch.where(tie, 0., cr.grad)),
           bnr.flatten()[i].item(), gact.flatten()[i].item(), int(tie.sum())), flush=True)
