"""CPU: the C-ABI library loads and exports every symbol include/rgda_hip.h declares (no compute)."""
import pytest

from regda_amd import _lib


def test_library_loads_and_exports_all_symbols():
    L = _lib.lib()
    assert L.missing == [], f'declared but not exported: {L.missing}'
    assert L.raw('rgda_abi_version')() == 2
    assert L.raw('rgda_strerror')(0) == b'ok'
    assert b'workspace' in L.raw('rgda_strerror')(-2)
    assert len(L.protos) >= 30


def test_workspace_queries_and_argument_errors():
    L = _lib.lib()
    assert L.size('rgda_lrh_workspace', 8, 256, 6) == (8 * 256 * 6 + 8 * 256) * 4 + 16
    assert L.size('rgda_pseudo_select_workspace', 8, 6) == 8 * 6 * 4 + 16
    # argument validation happens before anything touches the GPU
    with pytest.raises(ValueError):
        L.call('rgda_lrh', None, None, None, 1, 16, 6, -1, 0.5, 16, None, 0, None)
    with pytest.raises(ValueError):
        L.call('rgda_pseudo_select', None, None, 1, 6, 16, 0.8, 0.6, -1, 0, None, 0, None)
