"""dev: per-output-channel cosine of the head convolution's weight gradient (HIP vs the fp32 oracle, and the bf16-emulating
oracle vs the fp32 oracle) on the model_mid.npz inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import model as omodel
from oracle.step import CpuStep
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
g = np.load('tests/golden/model_mid.npz')
sd = omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=0.02)
t = lambda k: torch.from_numpy(g[k])
ms, mt = (t('m5')[0], t('m6')[0]), (t('m5')[1], t('m6')[1])
torch.set_num_threads(16)
res = {}
for emu in (False, 'grad'):
    cpu = CpuStep(sd, t('protos'), resnet_type='resnet101', lr=1e-2, emulate_bf16=emu)
    res[emu] = cpu.step(t('xs'), t('lab_s').long(), t('xt'), t('soft_t'), t('regs').long(), ms, mt)['grads']
m = Deeplabv2(dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
                   ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True))
m.load_state_dict(sd, strict=True)
m.set_drop_masks(torch.from_numpy(np.concatenate([g['m5'][0], g['m5'][1]])), torch.from_numpy(np.concatenate([g['m6'][0], g['m6'][1]])))
st = SSLStep(m, t('protos'))
c = lambda k, dt=None: torch.from_numpy(g[k] if dt is None else g[k].astype(dt)).cuda()
st.step(c('xs'), c('lab_s', np.int64), c('xt'), c('soft_t'), c('regs', np.int64), lr=1e-2)
torch.cuda.synchronize()
def cosv(a, b):
    a, b = a.flatten(1).double(), b.flatten(1).double()
    return (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1) + 1e-300)
for k in ('layer5.conv_last.0.weight', 'layer6.conv_last.0.weight', 'encoder.resnet.layer4.2.conv3.weight'):
    ref, emu, hip = res[False][k], res['grad'][k], m._gviews[k].detach().float().cpu()
    for name, x in (('emu', emu), ('hip', hip)):
        full = float(cosv(x.reshape(1, -1), ref.reshape(1, -1)))
        pc = cosv(x, ref)
        print('%-40s %s: whole %.4f  per-channel min %.4f median %.4f  ch0 %.4f  dropped-both? masks ch0 %s' % (
            k, name, full, float(pc.min()), float(pc.median()), float(pc[0]), (g['m5'][:, :, 0].tolist() if 'layer5' in k else '')))
        if k.startswith('layer5'):
            print('      feature half whole %.4f, PPM half whole %.4f' % (float(cosv(x[:, :2048].reshape(1, -1), ref[:, :2048].reshape(1, -1))),
                                                                          float(cosv(x[:, 2048:].reshape(1, -1), ref[:, 2048:].reshape(1, -1)))))
