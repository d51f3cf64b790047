"""dev: eager vs plan-replayed SSL steps on the shallow topology: per-step outputs and host enqueue times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model as omodel
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
rt = sys.argv[1] if len(sys.argv) > 1 else 'resnet17t'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
sd = omodel.init_state_dict(rt, 6, seed=12)
ones = torch.ones(2 * nb, 512)
b1, b2 = make_batch(b=nb, size=size, seed=21), make_batch(b=nb, size=size, seed=22)
seq = [b1, b1, b2, b1, b2, b2]
for mode in ('eager', 'plan', 'eager', 'plan'):
    m = Deeplabv2(cfg); m.load_state_dict(sd, strict=True); m.set_drop_masks(ones, ones)
    st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(5)), ema_decay=0.9)
    st.keep_debug = True
    for i, b in enumerate(seq):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == 'plan' and i == 1:
            print('   record:', st.record_plan(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t']))
            o = st._out
        else:
            o = st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
        d = st.debug
        print('%-5s step %d: loss_s %.5f loss_t %.5f gn %.4e | soft_in %.5f soft %.5f hard>=0 %.4f feat_t %.4f | host %.2f ms total %.2f ms' % (
            mode, i, o[0].item(), o[1].item(), o[2].item(), d['soft_in'].float().square().mean().item(), d['soft'].square().mean().item(),
            (st.last_hard >= 0).float().mean().item(), d['feat_t'].abs().mean().item(), th * 1e3, tt * 1e3))
