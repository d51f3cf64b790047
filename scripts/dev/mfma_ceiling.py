"""dev: what the matrix pipes SUSTAIN on this chip with nothing else going on -- 256 CUs x 8 waves issuing independent
v_mfma_f32_32x32x16_bf16 back to back from registers (tests/probes/probe.hip: probe_mfma_rate), for launches of
different lengths and for random vs all-zero operands.  The nominal 2.5 PFLOP/s is 8 passes per instruction at 2.4 GHz."""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dll = ctypes.CDLL(os.path.join(root, 'tests', 'probes', 'libprobe.so'))
dev = 'cuda'
out = torch.zeros(1 << 16, device=dev)
rnd = torch.randint(0, 2 ** 31 - 1, (4096,), dtype=torch.int32, device=dev)
# bf16 bit patterns of N(0,1) values in both halves of every word
v = torch.randn(8192, device=dev).to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xffff
rnd = (v[:4096] | (v[4096:] << 16)).to(torch.int32)
zero = torch.zeros(4096, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(seed, blocks, threads, iters, reps):
    f = lambda: dll.probe_run_mfma_rate(ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, threads,
                                        iters, ctypes.c_void_p(st))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * 32 * 32 * 16 * 8 * iters * blocks * (threads // 64)
    return ms * 1e3, flop / ms / 1e9


for name, seed in (('random', rnd), ('zeros', zero)):
    for wpc in (4, 8, 16):
        for iters in (100, 500, 2500, 12500, 50000):
            us, tf = run(seed, 256, wpc * 64, iters, 20 if iters <= 2500 else 5)
            print('%-6s %2d waves/CU  %6d x 8 MFMA per wave: %9.1f us  %6.0f TFLOP/s' % (name, wpc, iters, us, tf), flush=True)
