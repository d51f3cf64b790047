import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model as omodel
from tests.test_model_gpu import build, l2
rt = 'resnet17t'
m = build(rt)
sd = omodel.init_state_dict(rt, 6, seed=4)
gen = torch.Generator().manual_seed(9)
xs, xt = torch.randn(2, 3, 64, 64, generator=gen).cuda(), (torch.randn(2, 3, 64, 64, generator=gen) * 2 + 1).cuda()
ones = torch.ones(2, 512)
def sep():
    m.load_state_dict(sd, strict=True); m.train(); m.set_drop_masks(ones, ones)
    m._debug_taps = {}
    with torch.no_grad():
        a = m._forward_plan(xs, m.new_tape())
        ta = dict(m._debug_taps); m._debug_taps = {}
        b = m._forward_plan(xt, m.new_tape())
        tb = dict(m._debug_taps)
    return a, b, ta, tb
def grp():
    m.load_state_dict(sd, strict=True); m.train(); m.set_drop_masks(ones, ones)
    m._debug_taps = {}
    with torch.no_grad():
        c = m._forward_plan([xs, xt], m.new_tape(groups=2))
    return c, dict(m._debug_taps)
a, b, ta, tb = sep()
a2, b2, ta2, tb2 = sep()
c, tc = grp()
print('run-to-run separate tgt x1', l2(b2[0], b[0]))
for k in tb:
    print(k, 'grp-vs-sep tgt', l2(tc[k][2:], tb[k]), ' src', l2(tc[k][:2], ta[k]), ' sep run2run', l2(tb2[k], tb[k]))
