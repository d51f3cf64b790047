"""Developer script (not a test): prints error metrics of the HIP model vs the CPU oracle on the small case."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import model as omodel, labelpath as opath
from regda_amd.models.Encoder import Deeplabv2


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item(), (a - b).abs().max().item(), b.abs().max().item()


def main(rt='resnet101'):
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'model_small.npz'))
    sd = omodel.init_state_dict(rt, 6, seed=1)
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                       cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                       inchannels=2048, num_classes=6, is_ins_norm=True))
    print('keys', len(m.state_dict()), 'params', sum(p.numel() for p in m.parameters()))
    m.load_state_dict(sd, strict=True)
    m.train()
    xs = torch.from_numpy(g['xs'])
    m5, m6 = torch.from_numpy(g['m5'][0]), torch.from_numpy(g['m6'][0])
    big = '--big' in sys.argv
    if big:
        gen = torch.Generator().manual_seed(3)
        xs = torch.randn(4, 3, 128, 128, generator=gen)
        m5 = (torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8)
        m6 = (torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8)
    m.set_drop_masks(m5, m6)
    taps = {}
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    r1, r2, rf = omodel.forward(sdr, xs, True, (m5, m6), rt, {}, taps, emulate_bf16=('--emu' in sys.argv))
    m._debug_taps = {}
    x1, x2, feat = m(xs.cuda())
    torch.cuda.synchronize()
    for k, v in m._debug_taps.items():
        print('tap', k, rel(v, taps[k]))
    m._debug_taps = None
    print('x1   rel/maxabs/refmax', rel(x1, r1))
    print('x2  ', rel(x2, r2))
    print('feat', rel(feat, rf))
    lab = torch.from_numpy(g['lab_s'].astype(np.int64))
    if big:
        lab = torch.from_numpy(np.kron(np.random.default_rng(0).integers(-1, 6, size=(4, 8, 8)), np.ones((16, 16), np.int64)))
    for k, v in taps.items():
        if v.requires_grad:
            v.retain_grad()
    lref = opath.loss_calc([r1, r2], lab, -1)
    names = omodel.param_names(sd)
    gref = torch.autograd.grad(lref, [sdr[k] for k in names])
    from regda_amd.gast.balance import CrossEntropy
    from regda_amd.utils.tools import loss_calc
    loss = loss_calc([x1, x2], lab.cuda(), CrossEntropy(-1), multi=True)
    m._debug_grads = {}
    loss.backward()
    torch.cuda.synchronize()
    for k, v in m._debug_grads.items():
        if taps[k].grad is not None:
            print('gtap', k, rel(v, taps[k].grad))
    print('loss', loss.item(), lref.item())
    named = dict(m.named_parameters())
    worst = []
    for k, gr in zip(names, gref):
        e = rel(named[k].grad, gr)
        worst.append((e[0], k, e[1], e[2]))
    worst.sort(reverse=True)
    for w in worst[:12]:
        print('grad', w)
    for w in worst:
        if w[1].startswith(('layer5', 'encoder.resnet.layer4.1', 'encoder.resnet.layer1.0', 'encoder.resnet.conv1', 'encoder.resnet.bn1')):
            print('  g', w[1], round(w[0], 4), w[3])
    print('median rel', sorted(w[0] for w in worst)[len(worst) // 2])
    a = torch.cat([named[k].grad.float().cpu().reshape(-1) for k in names if 'ppm.0' not in k]); b = torch.cat([x.reshape(-1) for k, x in zip(names, gref) if 'ppm.0' not in k])
    print('global grad: cos', (a @ b / (a.norm() * b.norm())).item(), 'rel', ((a - b).norm() / b.norm()).item())
    tot = torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in names)).item()
    print('grad norm', tot, torch.sqrt(sum((x.double() ** 2).sum() for x in gref)).item())


if __name__ == '__main__':
    main('resnet17t' if '--tiny' in sys.argv else 'resnet101')
