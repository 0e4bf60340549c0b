"""Data-parallel gradient reduction: one process per GPU, torch.distributed.

The reference has no multi-GPU path (its run_1.sh starts unrelated runs on
different --gpu ids); data parallelism over the batch is exact here because
every op of the step treats samples independently and the loss is a mean.
Each rank computes gradients of its local-mean loss into the model's flat
gradient buffer; the buckets of that buffer (classifier + fusion head, attention
small terms, image projection -- the order backward completes them) are
sum-all-reduced as soon as they are complete, overlapping the rest of
backward, and the 1/world factor is folded into the Adam kernel's gradient
scale.  With the "nccl" backend (RCCL on ROCm) an async all-reduce runs on the
process group's own stream after the kernels already enqueued on the compute
stream; ``finish`` makes the compute stream wait for it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, process_group: Optional["dist.ProcessGroup"] = None, defer_last: bool = True):
        self.group = process_group
        # the last bucket (the image projection) only completes at the very end of backward: its
        # all-reduce is left running under the next step's parameter-independent prologue
        self.defer_last = defer_last
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self._handles: List = []
        self._flat_grad: Optional[torch.Tensor] = None
        self._buckets: Sequence[Tuple[int, int]] = ()
        self._fired: List[bool] = []

    def begin(self, model) -> None:
        """Bind to the model's flat gradient buffer for one backward pass."""
        self._flat_grad = model._flat_grad
        self._buckets = model.buckets
        self._fired = [False] * len(self._buckets)
        self._handles = []

    def bucket_ready(self, index: int) -> None:
        """Gradient bucket `index` is complete: start its all-reduce."""
        if self._fired[index]:
            return
        self._fired[index] = True
        if not self.enabled:
            return
        s, e = self._buckets[index]
        self._handles.append(dist.all_reduce(self._flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))

    def finish(self, defer_last: bool = False) -> float:
        """Wait for every bucket (reducing any that was never signalled) -- all but the last one
        when `defer_last` -- and return the factor that turns the summed gradients into the
        global-batch mean."""
        for i, done in enumerate(self._fired):
            if not done:
                self.bucket_ready(i)
        keep = []
        if defer_last and self.enabled and len(self._handles) == len(self._buckets):
            keep = [self._handles.pop()]           # handles are in bucket order
        for h in self._handles:
            h.wait()
        self._handles = keep
        return 1.0 / self.world

    def finish_last(self) -> None:
        """Wait for the bucket left in flight by finish(defer_last=True)."""
        for h in self._handles:
            h.wait()
        self._handles = []


def reduce_counters(counters: torch.Tensor, process_group=None) -> torch.Tensor:
    """Sum the integer per-question-type count / right counters over ranks; OA / AA
    computed from the result do not depend on the number of ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=process_group)
    return counters
