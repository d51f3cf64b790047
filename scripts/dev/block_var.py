import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops
BF = torch.bfloat16
G = 2
N, H, W, Cin, Cout = 16, 32, 32, 1024, 256
M = N * H * W
NB = 3
gen = torch.Generator(device='cuda').manual_seed(1)
c3 = [torch.randn(M, Cin, device='cuda', generator=gen).to(BF) for _ in range(NB)]
st3 = [ops.new_stats(G, 8, 2, Cin) for _ in range(NB)]
for s in st3: s.random_(0, 1 << 30)
res = [torch.randn(M, Cin, device='cuda', generator=gen).relu().to(BF) for _ in range(NB)]
gamma = torch.rand(Cin, device='cuda') * 0.2 + 0.05; beta = torch.randn(Cin, device='cuda') * 0.1
w1 = (torch.randn(Cout, 1, Cin, device='cuda') * 0.03).to(BF)
mi = torch.empty(G, 2, Cin, device='cuda')
yB = [torch.empty(M, Cin, dtype=BF, device='cuda') for _ in range(NB)]
mB = [torch.empty(M, Cin // 8, dtype=torch.uint8, device='cuda') for _ in range(NB)]
cB = [torch.empty(M, Cout, dtype=BF, device='cuda') for _ in range(NB)]
sB = [ops.new_stats(G, 8, 2, Cout) for _ in range(NB)]
it = [0]
def routeB():
    i = it[0] % NB; it[0] += 1
    bo = ops.bn_operand(st3[i], gamma, beta, mi, None, None, None, groups=G, relu=True)
    ops.conv1x1_block(bo, c3[i], res[i], yB[i], w1, cB[i], side_mask=mB[i], stats=sB[i], stat_groups=G)
def t_of(fn, reps=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rnd in range(2):
    for v, name in ((0, 'D=4'), (1, 'D=4 no side stores'), (2, 'D=8'), (3, 'D=8 no side stores')):
        os.environ['RGDA_BLK'] = str(v)
        print(name, '%.1f us' % t_of(routeB), flush=True)
