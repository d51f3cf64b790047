"""Record a static launch sequence once, replay it below the ABI (include/rgda_hip.h "plan replay").

A recorded step is a list of items in issue order:
  * segments  -- consecutive entry-point calls, packed as `rgda_plan_entry` rows and replayed by ONE call of
                 `rgda_plan_run` (a C loop: no Python, no ctypes marshalling per launch);
  * host actions -- everything between them that is not an entry point: torch ops on the step's static tensors
                 (zero_, cat, rand, add_, ...), stream / event waits, collectives.  They are closures, re-run at replay.

Recording runs the step eagerly (results are real) inside a private `torch.cuda.MemPool`: every tensor the step
allocates lives in that pool, so the addresses in the table stay valid and are never handed to anyone else while the
plan exists -- replay allocates nothing.  The code being recorded marks its host actions with `host(fn)`; entry-point
calls are captured by `_lib.call`.  Nothing here is global state of the library: the table and the pool belong to the
Plan object.
"""
import ctypes
import struct

import torch

from . import _lib

MAX_ARGS = 36


class PlanEntry(ctypes.Structure):
    _fields_ = [('fn', ctypes.c_int32), ('nargs', ctypes.c_int32), ('args', ctypes.c_uint64 * MAX_ARGS)]


_ACTIVE = None      # the Plan being recorded (one at a time)


def recording():
    return _ACTIVE is not None


def host(fn):
    """Run `fn()` now; when a plan is being recorded also store it as a host action of the plan (re-run at replay on
    the torch stream that is current now)."""
    global _ACTIVE
    act, _ACTIVE = _ACTIVE, None        # entry points called BY the action belong to it (re-run with it), not to the table
    try:
        out = fn()
    finally:
        _ACTIVE = act
    if act is not None:
        act._host(fn)
    return out


class EventBox:
    """An event recorded by one host action and waited for by a later one (fresh event at every replay)."""
    __slots__ = ('ev',)

    def __init__(self):
        self.ev = None


def record_event(stream, box=None):
    box = EventBox() if box is None else box
    host(lambda: setattr(box, 'ev', stream.record_event()))
    return box


def wait_event(stream, box):
    host(lambda: stream.wait_event(box.ev))


def wait_stream(waiter, other):
    """`waiter.wait_stream(other)` as a host action."""
    host(lambda: waiter.wait_stream(other))


def _pack(value, ctype, keep):
    if ctype is ctypes.c_float:
        return struct.unpack('<I', struct.pack('<f', float(value)))[0]
    if ctype is ctypes.c_double:
        return struct.unpack('<Q', struct.pack('<d', float(value)))[0]
    if ctype is ctypes.c_void_p:
        if value is None:
            return 0
        if isinstance(value, int):
            return value & 0xFFFFFFFFFFFFFFFF
        keep.append(value)                      # a host array / pointer object: must outlive the plan
        return (ctypes.cast(value, ctypes.c_void_p).value or 0) & 0xFFFFFFFFFFFFFFFF
    return int(value) & 0xFFFFFFFFFFFFFFFF


class Plan:
    def __init__(self):
        self.items = []             # ('seg', PlanEntry array, n) | ('host', fn)
        self._rows = []
        self._keep = []
        self._ids = {}
        self.pool = None
        self.n_calls = 0

    # ---- recording
    def _call(self, name, args):
        L = _lib.lib()
        fid = self._ids.get(name)
        if fid is None:
            fid = L.raw('rgda_plan_fn_id')(name.encode())
            if fid < 0:
                raise _lib.RgdaError(f'{name} is not a replayable entry point')
            self._ids[name] = fid
        types = L.protos[name][1]
        assert len(types) == len(args) <= MAX_ARGS, name
        self._rows.append((fid, [_pack(v, t, self._keep) for v, t in zip(args, types)]))
        self.n_calls += 1

    def _flush(self):
        if self._rows:
            arr = (PlanEntry * len(self._rows))()
            for e, (fid, vals) in zip(arr, self._rows):
                e.fn, e.nargs = fid, len(vals)
                for i, v in enumerate(vals):
                    e.args[i] = v
            self.items.append(('seg', arr, len(self._rows)))
            self._rows = []

    def _host(self, fn):
        self._flush()
        self.items.append(('host', fn, torch.cuda.current_stream()))

    def record(self, fn):
        """Run `fn()` eagerly under a private memory pool, recording its entry-point calls and host actions."""
        global _ACTIVE
        assert _ACTIVE is None, 'one plan is recorded at a time'
        # no cyclic garbage collection while the private pool is the active allocator: a collection that happens to run
        # here may finalise ANOTHER plan's MemPool (or a captured graph) from inside this pool's context, which aborts the
        # process in the allocator (seen once the step allocated enough Python objects to trigger a collection mid-record)
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        self.pool = torch.cuda.MemPool()
        _ACTIVE = self
        try:
            with torch.cuda.use_mem_pool(self.pool):
                out = fn()
        finally:
            _ACTIVE = None
            if gc_was_on:
                gc.enable()
        self._flush()
        return out

    # ---- replay
    def replay(self):
        L = _lib.lib()
        run = L.raw('rgda_plan_run')
        failed = ctypes.c_int(-1)
        cur = torch.cuda.current_stream()
        for it in self.items:
            if it[0] == 'host':
                if it[2] == cur:
                    it[1]()
                else:
                    with torch.cuda.stream(it[2]):
                        it[1]()
            else:
                rc = run(it[1], it[2], ctypes.byref(failed))
                if rc != 0:
                    msg = L.raw('rgda_strerror')(rc).decode()
                    raise _lib.RgdaError(f'plan row {failed.value} of a {it[2]}-row segment failed: {msg} (status {rc})')

    def stats(self):
        segs = [it for it in self.items if it[0] == 'seg']
        return dict(calls=self.n_calls, segments=len(segs), host_actions=len(self.items) - len(segs))


import sys as _sys
_lib._plan = _sys.modules[__name__]
