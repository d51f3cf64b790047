"""dev: which weight gradients end up in which grouped launch for a given wgrad_group_gflop (the flush points decide
where ~4.4 ms of weight-gradient kernels run next to the backward main chain)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
thr = [float(v) for v in sys.argv[1:]] or [250.0, 500.0]
model = Deeplabv2(dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False,
                       use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True))
step = SSLStep(model, torch.randn(6, 2048), ema_decay=0.999)
b = make_batch(b=8, size=512, seed=1, with_soft=False)
orig = Deeplabv2._flush_wgrads
log = []


def spy(self, T):
    pend = T['wgrad_pending']
    if pend:
        fl = sum(2.0 * it[3] * it[6] * it[7] * it[2].numel() for it in pend) / 1e9     # 2 * N*Ho*Wo * Cout*taps*Cin
        log.append((len(pend), fl, [tuple(it[2].shape) for it in pend][:3]))
    return orig(self, T)


Deeplabv2._flush_wgrads = spy
for t in thr:
    model.wgrad_group_gflop = t
    del log[:]
    step.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    torch.cuda.synchronize()
    print('threshold %.0f GFLOP: %d grouped launches' % (t, len(log)))
    for n, fl, shp in log:
        print('   %3d layers %7.0f GFLOP  first: %s' % (n, fl, shp[0]))
