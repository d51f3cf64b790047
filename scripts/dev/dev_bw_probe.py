import ctypes, os, sys, torch
dll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes', 'libprobe.so'))
src = torch.randn(64 * 1024 * 1024 // 4, device='cuda')
sink = torch.zeros(4, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(mode, depth, blocks, threads, tile, iters=200):
    def go():
        r = dll.probe_run_bw(mode, depth, blocks, threads, ctypes.c_void_p(src.data_ptr()), tile, iters, ctypes.c_void_p(sink.data_ptr()), st)
        assert r == 0
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tot = blocks * tile * iters
    return tot / ms / 1e6   # GB/s
for blocks in (1, 32, 256, 512):
    for threads in (64, 128, 256):
        row = 'blocks %4d threads %4d :' % (blocks, threads)
        for (mode, depth) in ((0, 1), (0, 2), (0, 4), (0, 8), (1, 1)):
            tile = 16384 if threads < 1024 else 16384
            g = run(mode, depth, blocks, threads, tile)
            row += ' %s%d %7.1f GB/s (%5.1f/blk)' % ('dma' if mode == 0 else 'reg', depth, g, g / blocks)
        print(row)
