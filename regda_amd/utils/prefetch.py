"""Host -> device input staging for the training loops.

The reference moves every tensor of a batch with `.cuda()` inside the iteration (tools/train_ssl_reg.py:200-206:
images_s, label_s, images_t, label_t_soft, regs_t), a synchronous pageable copy of ~84 MB per 8 + 8 batch of 512 x 512
tiles with int64 labels / region maps.  Here the loader's batches sit in PINNED host memory and are copied by a
dedicated HIP stream into one of `depth` device-resident slots while the previous step computes; the consumer only
waits for an event.  The tensors keep the reference's dtypes and shapes (the int64 API is the boundary)."""
import torch


class DevicePrefetcher:
    def __init__(self, host_batches, device=None, depth=2, into=None):
        """host_batches: list of {name: CPU tensor or None} with identical shapes; cycled through in order.
        into: {name: device tensor} -- stage every batch straight into THESE tensors (one slot: the static input
        buffers of a recorded step, SSLStep.static_inputs()); `release()` must then be given the event after which the
        step no longer reads its inputs (SSLStep.inputs_consumed)."""
        assert host_batches and (into is not None or depth >= 2)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.host = [{k: (None if v is None else v.contiguous().pin_memory()) for k, v in b.items()} for b in host_batches]
        if into is not None:
            depth = 1
            self.slots = [{k: into.get(k) for k in self.host[0]}]
            for k, v in self.host[0].items():
                assert (v is None) == (self.slots[0][k] is None) and (v is None or (v.shape == self.slots[0][k].shape and v.dtype == self.slots[0][k].dtype)), k
        else:
            self.slots = [{k: (None if v is None else torch.empty_like(v, device=self.device)) for k, v in self.host[0].items()}
                          for _ in range(depth)]
        self.single = into is not None
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.copied = [None] * depth        # event: the slot holds its batch
        self.consumed = [None] * depth      # event: the step that read the slot has been enqueued and finished with it
        self.i = 0                          # index of the next batch handed out
        self.bytes_per_batch = sum(v.numel() * v.element_size() for v in self.host[0].values() if v is not None)
        if self.single:
            # `into` are the live input buffers of a recorded step: whatever has been enqueued on the current stream so
            # far (a step still reading them) must be done before the first batch lands there
            self.consumed[0] = torch.cuda.current_stream().record_event()
        self._stage(0)

    def _stage(self, i):
        slot = i % len(self.slots)
        if self.consumed[slot] is not None:
            self.copy_stream.wait_event(self.consumed[slot])
        src = self.host[i % len(self.host)]
        with torch.cuda.stream(self.copy_stream):
            for k, dst in self.slots[slot].items():
                if dst is not None:
                    dst.copy_(src[k], non_blocking=True)
            self.copied[slot] = self.copy_stream.record_event()

    def next(self):
        """-> the device batch for this step (valid until `release()` + `depth - 1` further `next()` calls); the copy of
        the following batch is started on the copy stream."""
        slot = self.i % len(self.slots)
        torch.cuda.current_stream().wait_event(self.copied[slot])
        self._cur = slot
        if not self.single:
            self._stage(self.i + 1)
        self.i += 1
        return self.slots[slot]

    def release(self, consumed=None):
        """The step consuming the current batch has been enqueued on the current stream.  consumed: an event recorded
        where the step is done READING its inputs (default: the end of everything enqueued so far).  With a single slot
        the copy of the next batch is started here, behind that event."""
        self.consumed[self._cur] = consumed if consumed is not None else torch.cuda.current_stream().record_event()
        if self.single:
            self._stage(self.i)
