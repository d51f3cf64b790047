"""dev: cost of a software grid barrier over N co-resident workgroups (tests/probes/probe.hip: probe_grid_barrier)."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
dll = ctypes.CDLL(os.path.join(root, 'tests', 'probes', 'libprobe.so'))
for blocks in (64, 128, 256, 512):
    for rounds in (1, 100):
        cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
        out = torch.zeros(blocks, dtype=torch.int64, device='cuda')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = dll.probe_run_grid_barrier(ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, rounds,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        cyc = out.float()
        print('%4d workgroups, %3d rounds: kernel %.1f us, %.2f us per barrier, cycles per barrier median %.0f max %.0f' % (
            blocks, rounds, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / rounds, cyc.median().item() / rounds, cyc.max().item() / rounds))

print('hierarchical (per-XCD counters and flags):')
for blocks in (64, 128, 256, 512):
    for rounds in (1, 100):
        st = torch.zeros(17 * 32, dtype=torch.int32, device='cuda')
        out = torch.zeros(blocks, dtype=torch.int64, device='cuda')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = dll.probe_run_grid_barrier_hier(ctypes.c_void_p(st.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, rounds,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        cyc = out.float()
        print('%4d workgroups, %3d rounds: kernel %.1f us, %.2f us per barrier, cycles per barrier median %.0f max %.0f' % (
            blocks, rounds, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / rounds, cyc.median().item() / rounds, cyc.max().item() / rounds))
