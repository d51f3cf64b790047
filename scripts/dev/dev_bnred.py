"""dev: bn_bwd_reduce / bn_bwd_apply / bn_train_apply at the large geometries of the step (GB/s of what they move)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
dev = 'cuda'


def t_of(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, C) in [(262144, 256), (65536, 512), (16384, 1024), (16384, 2048), (1048576, 64), (16384, 512)]:
    # rotate over several buffers so the 256 MB Infinity Cache does not serve the re-reads
    nb = max(1, int(600e6 // (M * C * 4)))
    gs = [torch.randn(M, C, device=dev).to(BF) for _ in range(nb)]
    xs = [torch.randn(M, C, device=dev).to(BF) for _ in range(nb)]
    mi = torch.cat([torch.zeros(2, C), torch.ones(2, C)], 0).reshape(2, 2, C)[:, :, :].contiguous().to(dev)
    mi = torch.stack([torch.stack([torch.zeros(C), torch.ones(C)]) for _ in range(2)]).to(dev).contiguous()
    sums = torch.zeros(2 * 8 * 2 * C, device=dev)
    i = [0]
    def red():
        k = i[0] % nb; i[0] += 1
        ops.bn_bwd_reduce(gs[k], None, xs[k], mi, sums, M, C, False, groups=2)
    t = t_of(red)
    print('reduce M=%7d C=%4d: %6.1f us  %5.2f TB/s' % (M, C, t, 2 * M * C * 2 / t / 1e6), flush=True)
