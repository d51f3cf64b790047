"""dev: which part of torch.distributed / RCCL initialisation slows the single-GPU step down?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
mode = sys.argv[1]
kw = dict(init_method='tcp://127.0.0.1:29545', rank=0, world_size=1)
if mode == 'gloo':
    dist.init_process_group('gloo', **kw)
elif mode == 'nccl-lazy':
    dist.init_process_group('nccl', **kw)
elif mode in ('nccl-eager', 'nccl-destroy'):
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0), **kw)
elif mode == 'nccl-lazy-used':
    dist.init_process_group('nccl', **kw)
    t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
if mode == 'nccl-destroy':
    dist.destroy_process_group()
cfg = dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
m = Deeplabv2(cfg)
b = make_batch(b=8, size=512, seed=21, with_soft=False)
st = SSLStep(m, torch.randn(6, 2048), ema_decay=0.999)
st.reducer.force = False
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3)
print('%-16s ms/step %.3f %.3f' % (mode, run(10), run(10)), {k: v for k, v in os.environ.items() if 'NCCL' in k or 'RCCL' in k or 'HSA' in k})
