"""Data-parallel gradient averaging over the UNet's flat gradient buffer.

The reference has no distributed code; multi-GPU exists only through Lightning's DDP wrapper
(SURVEY.md 2d): one process per GPU, per-rank batch fixed, gradients all-reduced (SUM) and
divided by world size every step.  Here the gradients already live in ONE flat fp32 buffer in
layer order, and backward finalises it from the back (final conv first, time MLP last), so the
reducer simply all-reduces contiguous slices of that buffer as they become final -- RCCL
(`nccl` backend) runs them on its own stream while the remaining backward kernels execute.
xGMI is point-to-point (7 links x ~153 GB/s per GPU), so buckets are large (default 32 MB,
~4 per step for the 118 MB cfg-2 gradient) to stay bandwidth- rather than latency-bound.
The division by world size is folded into the fused Adam kernel (gscale).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, flat_grads: torch.Tensor, bucket_bytes: int = 32 << 20, group=None, tail_bytes: Optional[int] = None):
        self.flat = flat_grads
        self.bucket_elems = max(1, bucket_bytes // 4)
        # The LAST all-reduce of a step cannot overlap anything (backward is over), so it should be small: once the part of the
        # buffer that is still to come is below `tail_bytes` (default min(4 MB, a quarter bucket)), whatever is pending goes out at once instead
        # of waiting to be merged with it.  cfg 2: the final flush shrinks from the 22 MB remainder to the 2.4 MB time-MLP block.
        self.tail_elems = max(1, (min(4 << 20, bucket_bytes // 4) if tail_bytes is None else tail_bytes) // 4)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._works: List = []
        self._lo: Optional[int] = None
        self._hi: Optional[int] = None
        self._covered = 0
        self._tail_sent = False
        self.launched: List[tuple] = []          # (lo, hi) of every all-reduce of the current step
        # stream capture of a step (src/runtime/graphed.py): a bucket that is ready is handed to this callable instead of being
        # all-reduced -- the capture is cut there and the collective is issued between the replays of the two segments
        self.capture_sink = None
        # measurement switch (bench.py): the same buckets, but every collective is issued by finish() -- i.e. AFTER backward, nothing
        # overlapped -- so that (this step time) - (the overlapped step time) is the communication the overlap hides
        self.defer = False
        self._deferred: List[tuple] = []

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def begin(self):
        self._works.clear(); self.launched.clear(); self._deferred.clear()
        self._lo = self._hi = None
        self._covered = 0
        self._tail_sent = False

    def _launch(self):
        lo, hi = self._lo, self._hi
        self._lo = self._hi = None
        if lo is None or hi <= lo:
            return
        if self.launched and hi < self.launched[-1][0]:
            hi = self.launched[-1][0]             # alignment padding between two ranges: keep the launches gap-free
        self.launched.append((lo, hi))
        if self.capture_sink is not None:
            self.capture_sink(lo, hi)
        elif self.defer:
            self._deferred.append((lo, hi))
        elif self.world > 1:
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def all_reduce_async(self, lo: int, hi: int):
        """The collective of one recorded bucket (replay of a segmented graph step)."""
        if self.world > 1:
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait_all(self):
        for w in self._works:
            w.wait()
        self._works.clear()

    def range_ready(self, lo: int, hi: int):
        """Gradients in flat[lo:hi) are final.  Called in descending address order by backward."""
        if self._lo is None:
            self._lo, self._hi = lo, hi
        else:
            if hi > self._lo and lo < self._lo:       # overlapping/adjacent below: extend downwards
                pass
            self._lo = min(self._lo, lo)
            self._hi = max(self._hi, hi)
        if self._hi - self._lo >= self.bucket_elems:
            self._launch()
        elif self._lo <= self.tail_elems and not self._tail_sent:
            self._tail_sent = True                    # ... once: what follows (< tail_bytes) goes out with finish(), not piece by piece
            self._launch()

    def finish(self):
        """Flush the last bucket (always extended to offset 0) and make the current stream wait."""
        if self._lo is not None:
            self._lo = 0
            self._launch()
        for lo, hi in self._deferred:
            self.all_reduce_async(lo, hi)
        self._deferred.clear()
        for w in self._works:
            w.wait()
        self._works.clear()


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None):
    """DDP construction-time parameter broadcast: one collective over the flat buffer."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)   # in-place torch op: bumps flat._version for the bf16 shadows


def allreduce_flat_grads(nets, group=None):
    """Sum the flat gradient buffers of `nets` over the ranks (models that step their own optimizers call this between
    backward and step; the fused Adam divides by the world size through its grad_scale)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    works = [dist.all_reduce(n.flat_grads, op=dist.ReduceOp.SUM, group=group, async_op=True) for n in nets]
    for w in works:
        w.wait()
