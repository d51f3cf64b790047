"""The drop-in boundary from a host that is neither Python nor torch: examples/host_lrh.c (plain C, include/rgda_hip.h +
the HIP runtime) is compiled against the in-tree library -- here, without a GPU -- and run on the GPU box, where it checks
Homogenizer.forward on two hand-worked label maps (the 2 : 2 tie just below `percent` among them) and the status codes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'examples', 'host_lrh.c')
LIBDIR = os.path.join(ROOT, 'regda_amd', 'csrc')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(out):
    # gcc, not hipcc: the host side of the boundary is plain C
    cmd = ['gcc', '-std=c11', '-O1', '-D__HIP_PLATFORM_AMD__', '-o', out, SRC, '-I' + os.path.join(ROOT, 'include'),
           '-I' + os.path.join(ROCM, 'include'), '-L' + LIBDIR, '-lrgda_hip', '-L' + os.path.join(ROCM, 'lib'), '-lamdhip64',
           '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_c_host_compiles_against_the_header_and_library(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, 'librgda_hip.so')), 'build the library first (__graft_entry__.build())'
    _build(str(tmp_path / 'host_lrh'))


@pytest.mark.gpu
def test_c_host_runs_lrh_through_the_c_abi(tmp_path):
    exe = str(tmp_path / 'host_lrh')
    _build(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'host_lrh ok' in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
