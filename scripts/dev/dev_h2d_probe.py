"""dev: raw pinned host -> device copy rate of one 8 + 8 batch (84 MB in 4 tensors) on the copy stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd.synthetic import make_batch
from regda_amd.utils.prefetch import DevicePrefetcher
b = make_batch(b=8, size=512, seed=1, with_soft=False, device='cpu')
pf = DevicePrefetcher([b, b])
torch.cuda.synchronize()
for trial in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(pf.copy_stream):
        e0.record()
        for k, dst in pf.slots[0].items():
            dst.copy_(pf.host[0][k], non_blocking=True)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print('H2D %.1f MB in %.2f ms = %.1f GB/s' % (pf.bytes_per_batch / 1e6, ms, pf.bytes_per_batch / ms / 1e6))
big = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); dev = torch.empty_like(big, device='cuda')
torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(big, non_blocking=True); torch.cuda.synchronize()
print('one 256 MB pinned copy: %.1f GB/s' % (0.268 / (time.perf_counter() - t0)))
