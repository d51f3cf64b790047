import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model as omodel
from regda_amd.models.Encoder import Deeplabv2
rt = 'resnet17t'
m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                   cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                   inchannels=2048, num_classes=6, is_ins_norm=True))
sd = omodel.init_state_dict(rt, 6, seed=6)
gen = torch.Generator().manual_seed(13)
x = [torch.randn(2, 3, 64, 64, generator=gen).cuda(), torch.randn(2, 3, 64, 64, generator=gen).cuda()]
outs = []
g1, g2 = torch.randn(4, 6, 4, 4, generator=gen).cuda(), torch.randn(4, 6, 4, 4, generator=gen).cuda()
for it in range(4):
    m.relu_sign_mask = [False, True, False, True][it]
    m.load_state_dict(sd, strict=True)
    m.train()
    m.set_drop_masks(torch.ones(2, 512), torch.ones(2, 512))
    m._debug_taps = {}
    T = m.new_tape(groups=2)
    with torch.no_grad():
        c1, c2, f = m._forward_plan(x, T)
        taps = dict(m._debug_taps)
        m._debug_taps = None
        m._backward_plan(T, g1, g2)
    outs.append((c1.clone(), taps))
for it in range(1, 4):
    print('run', it, 'logits equal run0:', torch.equal(outs[0][0], outs[it][0]))
    for k in outs[0][1]:
        if not torch.equal(outs[0][1][k], outs[it][1][k]):
            print('   differing tap:', k, (outs[0][1][k] - outs[it][1][k]).abs().max().item())
