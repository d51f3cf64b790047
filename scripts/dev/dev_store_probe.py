import ctypes, os, torch
dll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes', 'libprobe.so'))
dll.probe_run_store.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (32, 128, 512, 2048):
    n = mb << 20
    dst = torch.empty(n, dtype=torch.uint8, device='cuda'); src = torch.empty(n, dtype=torch.uint8, device='cuda')
    row = '%5d MB:' % mb
    for mode, name in ((0, 'store'), (1, 'store.nt'), (2, 'copy'), (3, 'copy.nt')):
        best = 0
        for blocks in (1024, 2048, 8192, 65536):
            ms = t(lambda: dll.probe_run_store(mode, blocks, 256, dst.data_ptr(), src.data_ptr(), n // 16, st))
            best = max(best, n * (2 if mode >= 2 else 1) / ms / 1e6)
        row += '  %s %6.0f GB/s' % (name, best)
    ms = t(lambda: dst.fill_(1)); row += '  torch.fill %6.0f' % (n / ms / 1e6)
    ms = t(lambda: dst.copy_(src)); row += '  torch.copy %6.0f' % (2 * n / ms / 1e6)
    print(row)
