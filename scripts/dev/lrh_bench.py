"""dev: the SSL step's pseudo_selection + LRH chain (rgda_pseudo_lrh) at 8 x 512 x 512, tuning library:
RGDA_LRH_LDS = regions whose histogram rows live in LDS, RGDA_LRH_WG = least workgroups of the histogram kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
from regda_amd.synthetic import make_batch
b = make_batch(b=8, size=512, seed=2333)
g = torch.Generator().manual_seed(11)
blocks = torch.randn(8, 6, 32, 32, generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
softs = [torch.softmax(c * blocks + torch.randn(8, 6, 512, 512, generator=g), 1).contiguous().cuda() for c in (3.0, 0.5)]
regs = b['regs_t'].squeeze(1).contiguous()
def t_of(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ref = {}
for lds in ('2048',):
    for wg in ('64', '128', '256', '512'):
        os.environ['RGDA_LRH_LDS'] = lds; os.environ['RGDA_LRH_WG'] = wg
        ts = []
        for si, soft in enumerate(softs):
            cmax = soft.amax((2, 3)).contiguous()
            out, ws = ops.pseudo_lrh(soft, cmax, regs, 0.8, 0.6, 0.5, 6, -1, max_regions=4096)
            if si not in ref: ref[si] = out.clone()
            assert torch.equal(out, ref[si])
            ts.append(t_of(lambda: ops.pseudo_lrh(soft, cmax, regs, 0.8, 0.6, 0.5, 6, -1, max_regions=4096, ws=ws)))
        print('lds regions %5s  min workgroups %5s : %.1f us (confident labels) %.1f us (noisy labels)  [clear + pick_hist + gather]' % (lds, wg, ts[0], ts[1]), flush=True)
