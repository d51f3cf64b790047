"""GPU: pins the gfx950 instruction semantics the conv kernels are built on (tests/probes/probe.hip)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SO = os.path.join(os.path.dirname(__file__), 'probes', 'libprobe.so')


def test_mfma_32x32x16_layout():
    dll = ctypes.CDLL(SO)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(32, 16, generator=g).to(torch.bfloat16)
    B = torch.randn(16, 32, generator=g).to(torch.bfloat16)       # asymmetric on purpose
    Ad, Bd = A.cuda(), B.cuda()
    D = torch.zeros(32, 32, device='cuda')
    st = dll.probe_run_mfma(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()),
                            ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    ref = A.float() @ B.float()
    torch.testing.assert_close(D.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('stride', [160, 64])
def test_ds_read_tr16_b64_semantics(stride):
    dll = ctypes.CDLL(SO)
    T = torch.arange(16 * stride, dtype=torch.int16).reshape(16, stride)
    Td = T.cuda()
    out = torch.zeros(64, 2, 4, dtype=torch.int16, device='cuda')
    st = dll.probe_run_tr(ctypes.c_void_p(Td.data_ptr()), ctypes.c_int(stride), ctypes.c_void_p(out.data_ptr()),
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    out = out.cpu().numpy()
    exp = np.zeros((64, 2, 4), np.int16)
    Tn = T.numpy()
    for l in range(64):
        g, a = l >> 4, l & 15
        for s in range(2):
            for j in range(4):
                # lane gets T[k = (g>>1)*8 + 4*s + j][i = (g&1)*16 + a]  == A[i = l&31][k = (l>>5)*8 + 4*s + j]
                exp[l, s, j] = Tn[(g >> 1) * 8 + 4 * s + j, (g & 1) * 16 + a]
    if not np.array_equal(out, exp):
        os.makedirs('gpurun_out', exist_ok=True)
        np.save(f'gpurun_out/tr_probe_{stride}.npy', out)
    assert np.array_equal(out, exp)


def test_buffer_load_lds_out_of_range_writes_zeros():
    """conv kernels feed padded / out-of-range operand rows by giving the LDS-DMA an out-of-range
    buffer offset: the hardware must deposit zeros in the LDS slot (not skip it)."""
    dll = ctypes.CDLL(SO)
    src = torch.arange(256, dtype=torch.float32, device='cuda') + 1
    out = torch.zeros(64, 4, device='cuda')
    st = dll.probe_run_buflds(ctypes.c_void_p(src.data_ptr()), ctypes.c_int(1024), ctypes.c_void_p(out.data_ptr()),
                              ctypes.c_int(5), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    exp = src.reshape(64, 4).clone()
    exp[5] = 0
    assert torch.equal(out.cpu(), exp.cpu()), out[:8]
