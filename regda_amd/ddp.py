"""Data-parallel gradient exchange for the flat fp32 gradient buffer (new capability: the reference is
single-GPU).  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI)
on the GPU box, "gloo" in the CPU tests.

The gradient is ONE flat tensor laid out in forward order, so backward finalises it from the END towards
the start: buckets are contiguous ranges issued in reverse order as soon as the backward pass has moved
below them (the last one, which nothing can overlap, is kept short), each as an async all-reduce (sum) that overlaps the remaining dgrad/wgrad kernels; the
1/world scale is folded into the fused clip+SGD kernel (`gscale`), not applied as a separate pass.
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


class FlatGradReducer:
    def __init__(self, flat_g, boundaries, bucket_elems=12 << 20, group=None):
        self.flat_g = flat_g
        self.group = group
        import os
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = dist.is_initialized() and bool(os.environ.get('RGDA_FORCE_DDP'))   # single-GPU exercise of the path
        self.buckets = make_buckets(boundaries, flat_g.numel(), bucket_elems)
        self._next = 0
        self._works = []

    def reset(self):
        self._next = 0
        self._works = []

    def ready_down_to(self, offset):
        """Backward has finished every gradient at element offset >= `offset`: launch the buckets that
        lie entirely above it."""
        if self.world == 1 and not self.force:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= offset:
            a, b = self.buckets[self._next]
            self._works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self._next += 1

    def finish(self):
        """Launch what is left and make the current stream wait for every bucket."""
        self.ready_down_to(0)
        for w in self._works:
            w.wait()
        self._works = []

    @property
    def gscale(self):
        return 1.0 / self.world
