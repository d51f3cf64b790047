"""Data-parallel gradient exchange for the flat fp32 gradient buffer (new capability: the reference is
single-GPU).  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI)
on the GPU box, "gloo" in the CPU tests.

The gradient is ONE flat tensor laid out in forward order, so backward finalises it from the END towards
the start: buckets are contiguous ranges issued in reverse order as soon as the backward pass has moved
below them (the last one, which nothing can overlap, is kept short), each overlapping the remaining dgrad/wgrad kernels; the
1/world scale is folded into the fused clip+SGD kernel (`gscale`), not applied as a separate pass.

Two payloads (`payload=`):
  'fp32' (default until an 8-GPU run says otherwise): one async all-reduce (sum) of the fp32 bucket -- 2 (N-1)/N x 4 bytes
         per element over every rank's links.
  'bf16': the bucket is cast to bf16 and exchanged in two phases -- all-to-all of the N shards (each rank receives ITS
         shard from every rank), fp32 accumulation of the N copies on arrival in rank order, the bf16-rounded sums
         all-gathered and widened back into the fp32 buffer: 2 (N-1)/N x 2 bytes per element, HALF the fp32 all-reduce at
         every N, no bf16 addition anywhere (a bf16 all-reduce would round after each of its N-1 adds).  Every rank ends
         with the same bits (its own shard goes through the same bf16 rounding as everyone else's copy of it).  The sum
         differs from the fp32 all-reduce by at most half a bf16 ulp per contribution plus half an ulp of the result.
"""
import torch
import torch.distributed as dist


def make_buckets(boundaries, total, bucket_elems, tail_elems=None):
    """boundaries: ascending element offsets where a bucket may be cut (layer starts).  Returns
    [(start, end)] in ISSUE order (last range first), each at least `bucket_elems` long except the last.

    The LAST bucket (the start of the buffer = the first layers of the net) is the one all-reduce that nothing can
    hide: its gradients are final only when backward ends.  `tail_elems` (default bucket_elems / 8) keeps it short:
    it ends at the largest cut <= tail_elems, and the layers above it -- final while the early layers' backward is
    still running -- go into the bucket before."""
    cuts = sorted(set(b for b in boundaries if 0 < b < total))
    if tail_elems is None:
        tail_elems = bucket_elems // 8
    tail = max([c for c in cuts if c <= tail_elems], default=0)
    buckets, end = [], total
    for c in reversed(cuts):
        if c <= tail:
            break
        if end - c >= bucket_elems:
            buckets.append((c, end))
            end = c
    if tail and end > tail:
        buckets.append((tail, end))
        end = tail
    buckets.append((0, end))
    return buckets


def all_reduce_prototype_statistics(stats, class_num, feat_channels, group=None, comm=None, world=None):
    """Cross-rank state of `Aligner.update_prototype` (regda/gast/alignment.py:300-327).  `stats` is the flat
    float32 buffer rgda_proto_stats leaves: sums[c][k] (sum of the source features over the pixels of downscaled class c)
    followed by cnt[c].  Both ADD over batches, so one all-reduce (sum) of these class_num * (feat_channels + 1) floats
    gives every rank the statistics of the concatenated global batch; rgda_proto_apply then forms
    sums / (cnt + 1e-7), keeps the old prototype where the GLOBAL count is < 1, and does the EMA -- what the reference
    computes on the whole batch.  (Averaging per-rank prototypes is a different number whenever class counts differ
    across ranks, and lets a rank without a class vote for the old prototype.)

    world: what the CALLER believes the number of ranks is.  With more than one rank there must be something to exchange
    through -- an RcclComm or an initialised process group; silently applying the local statistics would let the
    prototypes of the ranks drift apart."""
    n = class_num * feat_channels + class_num
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= n
    if comm is not None:                    # an RcclComm: enqueued on the current stream
        comm.all_reduce(stats[:n])
    elif dist.is_initialized():
        dist.all_reduce(stats[:n], op=dist.ReduceOp.SUM, group=group)
    elif world is not None and world > 1:
        raise RuntimeError('prototype statistics of %d ranks cannot be exchanged: no RcclComm was given and '
                           'torch.distributed is not initialised' % world)
    return stats


class RcclComm:
    """One RCCL communicator behind the C ABI (include/rgda_hip.h: rgda_comm_*): the collectives of the data-parallel step
    for a host WITHOUT torch.distributed -- the wrappers SURVEY.md 8b lists in the boundary's op set.  Rank 0 draws the
    128-byte id (`RcclComm.unique_id()`) and the host ships it to the other ranks (a file, a socket, MPI; in a torch
    process `RcclComm.from_torch_store()` uses the default process group's store); every rank constructs the
    communicator on its current device.  Calls enqueue on the given (default: the current) stream; tensors are
    contiguous device tensors, reduced / exchanged in place as FlatGradReducer does with torch.distributed.

    Threading and ordering contract (RCCL's, restated because the step leans on it): ONE host thread drives a
    communicator, and every rank issues the SAME collectives in the SAME host order.  The step issues them from more than
    one stream -- gradient buckets on the reducer's communication stream, the 48 KB prototype statistics on the second
    (weight-gradient) stream, ClassBalance's counts on the current stream: RCCL serialises the collectives of one
    communicator in host issue order whatever stream each was enqueued on (a later one waits for the earlier one's
    kernel), so a small collective issued between two buckets delays the second bucket by its latency (~20-50 us) and no
    more; what must never happen is two ranks issuing in different orders, which the step's static launch sequence rules
    out.  `tests/test_ddp_gpu.py::test_two_rank_rccl_step` runs buckets + statistics at world 2 where two GPUs exist."""
    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.float64: 3}       # RGDA_COMM_F32 / BF16 / I64 / F64

    @staticmethod
    def unique_id():
        import ctypes
        from ._lib import lib
        buf = (ctypes.c_char * 128)()
        lib().call('rgda_comm_unique_id', buf)
        return bytes(buf)

    def __init__(self, unique_id, rank, world):
        import ctypes
        from ._lib import lib
        self._h = None
        assert len(unique_id) == 128
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        lib().call('rgda_comm_init', ctypes.c_char_p(unique_id), self.rank, self.world, ctypes.byref(h))
        self._h = h

    _store_serial = 0       # communicators made through the store so far, in this process (identical on every rank)

    @classmethod
    def from_torch_store(cls, key='rgda_comm_id'):
        """Inside a torch.distributed job: the id travels through the default group's store, nothing else of torch is used.
        Every call uses a fresh store key (`key` + a per-process serial that advances identically on all ranks, since
        every rank makes the same communicators in the same order): a second communicator can never pick up the first
        one's id while rank 0 has not yet published the new one."""
        rank, world = dist.get_rank(), dist.get_world_size()
        store = dist.distributed_c10d._get_default_store()
        key = '%s/%d' % (key, cls._store_serial)
        cls._store_serial += 1
        if rank == 0:
            store.set(key, cls.unique_id())
        return cls(bytes(store.get(key)), rank, world)

    def _args(self, t, stream):
        assert t.is_cuda and t.is_contiguous() and t.dtype in self._DT, (t.device, t.dtype)
        return self._DT[t.dtype], (stream if stream is not None else torch.cuda.current_stream()).cuda_stream

    def all_reduce(self, t, stream=None):
        from ._lib import lib
        dt, st = self._args(t, stream)
        lib().call('rgda_comm_all_reduce', self._h, t.data_ptr(), t.numel(), dt, st)

    def all_gather(self, send, recv, stream=None):
        from ._lib import lib
        dt, st = self._args(send, stream)
        assert recv.dtype == send.dtype and recv.is_contiguous() and recv.numel() == send.numel() * self.world
        lib().call('rgda_comm_all_gather', self._h, send.data_ptr(), recv.data_ptr(), send.numel(), dt, st)

    def all_to_all(self, send, recv, stream=None):
        from ._lib import lib
        dt, st = self._args(send, stream)
        assert recv.dtype == send.dtype and recv.is_contiguous() and recv.numel() == send.numel() and send.numel() % self.world == 0
        lib().call('rgda_comm_all_to_all', self._h, send.data_ptr(), recv.data_ptr(), send.numel() // self.world, dt, st)

    def destroy(self):
        if self._h is not None:
            from ._lib import lib
            lib().call('rgda_comm_destroy', self._h)
            self._h = None

    def __del__(self):
        # a communicator nobody destroyed: best effort (never raise out of a finalizer; at interpreter exit the library
        # or the device may already be gone)
        try:
            self.destroy()
        except Exception:
            pass


class FlatGradReducer:
    def __init__(self, flat_g, boundaries, bucket_elems=12 << 20, group=None, payload='fp32', comm=None):
        """comm: an `RcclComm` -- the buckets then go through the library's own RCCL entry points (on a stream of the
        reducer's) instead of torch.distributed; None (default): torch.distributed on `group`."""
        assert payload in ('fp32', 'bf16')
        self.flat_g = flat_g
        self.group = group
        self.payload = payload
        self.comm = comm
        import os
        if comm is not None:
            assert flat_g.is_cuda
            self.world = comm.world
            self.force = bool(os.environ.get('RGDA_FORCE_DDP'))
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.force = dist.is_initialized() and bool(os.environ.get('RGDA_FORCE_DDP'))   # single-GPU exercise of the path
        self.buckets = make_buckets(boundaries, flat_g.numel(), bucket_elems)
        self._next = 0
        self._works = []
        self.measure = False        # bench: a timing event on the issuing stream per bucket (`issue_events`, cleared by reset())
        self.issue_events = []
        self._stage = {}            # bf16 payload: per-bucket staging buffers, allocated once
        self._comm_stream = None    # bf16 payload on a GPU: the two collectives and the kernels between them run here
        if comm is not None and (self.world > 1 or self.force):
            self._comm_stream = torch.cuda.Stream(device=flat_g.device)
        if payload == 'bf16' and (self.world > 1 or self.force):
            # every staging buffer and the stream exist before the first exchange: nothing is allocated inside a recorded
            # plan's private pool or on the communication stream
            for a, b in self.buckets:
                self._staging(a, b)
            if flat_g.is_cuda and self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat_g.device)

    def reset(self):
        self._next = 0
        self._works = []
        self.issue_events = []

    @property
    def active(self):
        return self.world > 1 or self.force

    # ---- bf16 payload
    def _staging(self, a, b):
        st = self._stage.get((a, b))
        if st is None:
            n, W = b - a, self.world
            shard = -(-n // (W * 8)) * 8                    # elements per rank, a multiple of 8 (16-byte vectors)
            dev = self.flat_g.device
            z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
            st = dict(shard=shard, send=z(W * shard), recv=z(W * shard), red=z(shard), gath=z(W * shard))
            self._stage[(a, b)] = st
        return st

    def _exchange_bf16(self, a, b):
        st, W, g = self._staging(a, b), self.world, self.flat_g[a:b]
        n = b - a
        if g.is_cuda:
            from . import ops
            ops.cast_bf16(g, st['send'])                    # (the tail of `send` beyond n stays zero)
            if self.comm is not None:
                self.comm.all_to_all(st['send'], st['recv'])
            else:
                dist.all_to_all_single(st['recv'], st['send'], group=self.group)
            ops.ddp_accumulate_bf16(st['recv'], W, st['red'])
            if self.comm is not None:
                self.comm.all_gather(st['red'], st['gath'])
            else:
                dist.all_gather_into_tensor(st['gath'], st['red'], group=self.group)
            ops.cast_f32(st['gath'], g)
        else:       # gloo on CPU tensors (tests): the same arithmetic in torch
            st['send'][:n].copy_(g)
            dist.all_to_all_single(st['recv'], st['send'], group=self.group)
            acc = torch.zeros(st['shard'])
            for r in range(W):                              # rank order
                acc += st['recv'][r * st['shard']:(r + 1) * st['shard']].float()
            st['red'].copy_(acc)
            dist.all_gather_into_tensor(st['gath'], st['red'], group=self.group)
            g.copy_(st['gath'][:n])

    def ready_down_to(self, offset):
        """Backward has finished every gradient at element offset >= `offset`: launch the buckets that
        lie entirely above it."""
        if not self.active:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= offset:
            a, b = self.buckets[self._next]
            if self.measure and self.flat_g.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()             # on the stream the bucket is issued from: when its gradients were final
                self.issue_events.append((self._next, ev))
            if self.payload == 'fp32' and self.comm is not None:
                # (the same shape as the bf16 route: a stream of the reducer's own behind the issuing stream)
                self._comm_stream.wait_stream(torch.cuda.current_stream())
                self.comm.all_reduce(self.flat_g[a:b], stream=self._comm_stream)
            elif self.payload == 'fp32':
                self._works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True))
            elif self.flat_g.is_cuda:
                # a stream of its own: the collectives and the small kernels between them are ordered there, behind what
                # the caller's stream has produced so far, and the caller's stream (weight gradients) runs on
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=self.flat_g.device)
                self._comm_stream.wait_stream(torch.cuda.current_stream())
                from . import ops
                with ops.use_stream(self._comm_stream):
                    self._exchange_bf16(a, b)
            else:
                self._exchange_bf16(a, b)
            self._next += 1

    def finish(self):
        """Launch what is left and make the current stream wait for every bucket."""
        self.ready_down_to(0)
        for w in self._works:
            w.wait()
        self._works = []
        if self._comm_stream is not None and self.active:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    @property
    def gscale(self):
        return 1.0 / self.world
