"""CPU: the C-ABI library loads and exports every symbol include/rgda_hip.h declares (no compute)."""
import pytest

from regda_amd import _lib


def test_library_loads_and_exports_all_symbols():
    L = _lib.lib()
    assert L.missing == [], f'declared but not exported: {L.missing}'
    assert L.raw('rgda_abi_version')() == 10
    # the fixed-point formats of the per-channel accumulators (rgda_stat_t) as the Python side scales them
    import re
    from regda_amd import ops
    hdr = open(_lib.HEADER_PATH).read()
    assert int(re.search(r'#define\s+RGDA_STAT_FRAC_FWD\s+(\d+)', hdr).group(1)) == ops.STAT_FRAC_FWD
    assert int(re.search(r'#define\s+RGDA_STAT_FRAC_BWD\s+(\d+)', hdr).group(1)) == ops.STAT_FRAC_BWD
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


def test_rccl_wrappers_validate_their_arguments():
    """rgda_comm_*: argument errors are reported before librccl.so is even looked for (no GPU, no RCCL needed here)."""
    import ctypes
    L = _lib.lib()
    buf = (ctypes.c_char * 128)()
    h = ctypes.c_void_p()
    with pytest.raises(ValueError):
        L.call('rgda_comm_unique_id', None)
    for args in ((None, 0, 1, ctypes.byref(h)), (buf, 0, 1, None), (buf, 1, 1, ctypes.byref(h)), (buf, 0, 0, ctypes.byref(h))):
        with pytest.raises(ValueError):
            L.call('rgda_comm_init', *args)
    fake = ctypes.c_void_p(16)
    with pytest.raises(ValueError):
        L.call('rgda_comm_destroy', None)
    with pytest.raises(ValueError):
        L.call('rgda_comm_all_reduce', None, fake, 8, 0, None)
    with pytest.raises(ValueError):
        L.call('rgda_comm_all_reduce', fake, fake, 8, 7, None)          # unknown dtype code
    with pytest.raises(ValueError):
        L.call('rgda_comm_all_gather', fake, fake, None, 8, 0, None)
    with pytest.raises(ValueError):
        L.call('rgda_comm_all_to_all', fake, fake, fake, 8, 1, None)    # send == recv


def test_plan_replay_dispatch_table_and_error_rows():
    """rgda_plan_run: every replayable entry point is in the dispatch table with the argument count of its prototype;
    rows are called in order and the first failing row stops the walk (argument errors need no GPU)."""
    import ctypes
    from regda_amd.plan import MAX_ARGS, PlanEntry
    L = _lib.lib()
    n = L.raw('rgda_plan_fn_count')()
    replayable = [k for k, (res, _) in L.protos.items() if res is ctypes.c_int and k not in
                  ('rgda_abi_version', 'rgda_plan_run', 'rgda_plan_fn_id', 'rgda_plan_fn_count')]
    assert n == len(replayable)
    ids = {k: L.raw('rgda_plan_fn_id')(k.encode()) for k in replayable}
    assert sorted(ids.values()) == list(range(n))
    assert L.raw('rgda_plan_fn_id')(b'rgda_lrh_workspace') == -1 and L.raw('rgda_plan_fn_id')(b'nope') == -1
    assert max(len(a) for _, a in L.protos.values()) <= MAX_ARGS
    rows = (PlanEntry * 3)()
    rows[0].fn, rows[0].nargs = ids['rgda_conv2d_wgrad_grouped'], 5          # (NULL, 0, NULL, 0, stream): a valid empty list
    rows[1].fn, rows[1].nargs = ids['rgda_lrh'], len(L.protos['rgda_lrh'][1])    # null pointers -> RGDA_ERR_ARG
    rows[2].fn, rows[2].nargs = ids['rgda_conv2d_wgrad_grouped'], 5
    failed = ctypes.c_int(-1)
    assert L.raw('rgda_plan_run')(rows, 1, ctypes.byref(failed)) == 0 and failed.value == -1
    assert L.raw('rgda_plan_run')(rows, 3, ctypes.byref(failed)) == -1 and failed.value == 1
    rows[0].nargs = 2                                                          # wrong argument count for the entry point
    assert L.raw('rgda_plan_run')(rows, 1, ctypes.byref(failed)) == -1 and failed.value == 0
    rows[0].fn = n
    assert L.raw('rgda_plan_run')(rows, 1, None) == -1
    assert L.raw('rgda_plan_run')(None, 0, None) == 0
