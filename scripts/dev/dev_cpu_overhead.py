import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
model = Deeplabv2(dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True))
step = SSLStep(model, torch.randn(6, 2048), ema_decay=0.999)
b = make_batch(b=8, size=512, seed=1, with_soft=False)
for _ in range(3):
    step.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step ; total %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
