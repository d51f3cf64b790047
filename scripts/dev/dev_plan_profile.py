"""dev: where the host time of a replayed plan goes (per item), full-size step."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import _lib
from regda_amd.models.Encoder import Deeplabv2
from regda_amd.ssl import SSLStep
from regda_amd.synthetic import make_batch
cfg = dict(backbone=dict(resnet_type='resnet101', output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=True,
           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6, is_ins_norm=True)
m = Deeplabv2(cfg)
b = make_batch(b=8, size=512, seed=21, with_soft=False)
st = SSLStep(m, torch.randn(6, 2048), ema_decay=0.999)
for _ in range(2):
    st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
print(st.record_plan(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t']))
p = st._plan
L = _lib.lib(); run = L.raw('rgda_plan_run'); failed = ctypes.c_int(-1)
for trial in range(2):
    torch.cuda.synchronize()
    rows, t_seg, t_host, log = 0, 0.0, 0.0, []
    t00 = time.perf_counter()
    for it in p.items:
        t0 = time.perf_counter()
        if it[0] == 'host':
            with torch.cuda.stream(it[2]):
                it[1]()
            dt = time.perf_counter() - t0; t_host += dt
            log.append(('host', 1, dt, time.perf_counter() - t00))
        else:
            run(it[1], it[2], ctypes.byref(failed))
            dt = time.perf_counter() - t0; t_seg += dt; rows += it[2]
            log.append(('seg', it[2], dt, time.perf_counter() - t00))
    tot = time.perf_counter() - t00
    torch.cuda.synchronize()
    print('trial %d: total host %.2f ms: segments %.2f ms (%d rows), host actions %.2f ms (%d); GPU done at %.2f ms' % (
        trial, tot * 1e3, t_seg * 1e3, rows, t_host * 1e3, sum(1 for l in log if l[0] == 'host'), (time.perf_counter() - t00) * 1e3))
    acc = 0
    for kind, n, dt, at in log:
        if kind == 'seg':
            acc += n
            print('   seg %4d rows  %7.1f us  (%5.1f us/row)  at %6.2f ms, rows so far %d' % (n, dt * 1e6, dt * 1e6 / n, at * 1e3, acc))
        elif dt > 30e-6:
            print('   host action %7.1f us at %6.2f ms' % (dt * 1e6, at * 1e3))

print('--- st.step() in pieces')
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for dst, src in zip(st._static, (b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'])):
        if dst is not None and src is not dst:
            dst.copy_(src, non_blocking=True)
    t1 = time.perf_counter()
    st.lr_dev.fill_(1e-3)
    t2 = time.perf_counter()
    chk = m.flat_p._version != m._synced_version
    t3 = time.perf_counter()
    st._plan.replay()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    print('copies %.2f ms, lr fill %.2f ms, version check %.3f ms (%s), replay %.2f ms, GPU done %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, chk, (t4 - t3) * 1e3, (time.perf_counter() - t0) * 1e3))
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], 1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('st.step host %.2f ms, GPU done %.2f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
