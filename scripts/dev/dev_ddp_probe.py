"""dev: what does initialising torch.distributed("nccl") at world size 1 cost the step, and the all-reduces themselves?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
mode = sys.argv[1]      # none | init | force
if mode != 'none':
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29544', rank=0, world_size=1, device_id=torch.device('cuda', 0))
if mode == 'force':
    os.environ['RGDA_FORCE_DDP'] = '1'
cfg = dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
m = Deeplabv2(cfg)
b = make_batch(b=8, size=512, seed=21, with_soft=False)
st = SSLStep(m, torch.randn(6, 2048), ema_decay=0.999)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3)
print(mode, 'eager  ms/step %.3f %.3f' % (run(10), run(10)))
if mode != 'none':
    # the bare collectives: 354 MB in the step's buckets, nothing else on the GPU
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        st.reducer.force = True
        st.reducer.reset(); st.reducer.finish()
    torch.cuda.synchronize(); print(mode, 'bare all-reduce of the flat gradient: %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3), len(st.reducer.buckets), 'buckets')
    st.reducer.force = (mode == 'force')
    dist.destroy_process_group()
