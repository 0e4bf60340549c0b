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


def reserve_compute_units_for_resident_kernels(max_channels: int = 48) -> None:
    """Call BEFORE ``init_process_group``.  The question encoder's weight-stationary kernel (vqa_lstm_resident) needs
    200 of the 256 compute units to itself for the length of its launch -- each of its workgroups takes a whole CU's
    registers and all of them meet at grid barriers -- and it runs while the previous step's gradient all-reduce is
    still in flight.  An RCCL kernel occupies one CU per channel, so the channel count is capped (unless the user set
    NCCL_MAX_NCHANNELS) to keep 200 CUs free; beyond the cap the encoder would simply start after the collective."""
    import os
    os.environ.setdefault('NCCL_MAX_NCHANNELS', str(int(max_channels)))


def connect_collectives(device, process_group=None, nbytes: int = 32 << 20) -> None:
    """Call once AFTER ``init_process_group`` and before the first train step.  RCCL connects its peers (rings, channels,
    the protocol for large messages) inside the first collective of each kind, which can take seconds and holds compute
    units meanwhile; doing that here, with one bucket-sized and one small all-reduce followed by a device
    synchronise, keeps it out of the step -- where the weight-stationary encoder would otherwise wait at its first
    grid barrier for the compute units the connecting collective holds."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for n in (max(int(nbytes) // 4, 1), 1):
        t = torch.zeros(n, device=device, dtype=torch.float32)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    if torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)


class GradReducer:
    def __init__(self, process_group: Optional["dist.ProcessGroup"] = None, defer_last: bool = True,
                 force: bool = False, bucket_bytes: int = 32 << 20, defer_all: bool = True):
        """`bucket_bytes`: completed buckets are held back until this much gradient is ready and then reduced
        in one collective (every collective costs the compute stream two cross-stream dependencies, about
        20 us of GPU time on MI355X, whatever its size); the last bucket always goes on its own so that it
        can be deferred.  `force` issues the collectives even in a group of one rank (where they are the
        identity): a single-GPU machine can then exercise the RCCL stream ordering of the bucketed path."""
        self.group = process_group
        # compute units the collectives' kernels may hold while backward is still running (one per RCCL channel: the
        # cap of reserve_compute_units_for_resident_kernels or the user's own NCCL_MAX_NCHANNELS, plus a margin):
        # kernels whose workgroups all have to be resident at once (the big weight gradient, the weight-stationary
        # encoder) are sized to fit beside them
        import os
        channels = os.environ.get('NCCL_MAX_NCHANNELS', '')
        self.collective_cus = min((int(channels) if channels.isdigit() else 48) + 8, 128)
        # the last bucket (the image projection) only completes at the very end of backward: its
        # all-reduce is left running under the next step's parameter-independent prologue
        self.defer_last = defer_last
        # ... and with `defer_all` no bucket is waited for inside the step at all: every collective keeps
        # running under the rest of backward and the next step's prologue, and the whole Adam update is
        # applied there after a single cross-stream wait (same arithmetic, same results)
        self.defer_all = defer_all and defer_last
        self.bucket_bytes = bucket_bytes
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or force)
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self._in_order = self.enabled and dist.get_backend(process_group) == 'nccl'
        self._handles: List = []
        self._last_handle = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._buckets: Sequence[Tuple[int, int]] = ()
        self._fired: List[bool] = []
        self._hold_all = False
        self._next = 0                      # first bucket not yet handed to a collective or to the pending run
        self._run: Optional[Tuple[int, int]] = None   # element range of the completed buckets being held back
        self.launched: List[Tuple[int, int]] = []     # element ranges of the collectives of this pass, in order
        # timing (switched on by setting `timing = {}`): per collective the bytes and an event on the compute stream
        # at its launch; per wait an event pair on the compute stream around the cross-stream wait.  The wait pair is
        # the communication the compute stream could not hide; launch -> end of the wait bounds the collective's span.
        # An event record costs the stream about 10 us of idle time, so only every `timing_stride`-th pass is timed
        # (its launches, and the wait that completes them, which may sit in the next step).
        self.timing: Optional[dict] = None
        self.timing_stride = 1
        self._pass = -1
        self._timed = False

    def begin(self, model) -> None:
        """Bind to the model's flat gradient buffer for one backward pass."""
        self._flat_grad = model._flat_grad
        self._buckets = model.buckets
        self._fired = [False] * len(self._buckets)
        self._handles = []
        self._last_handle = None
        self._next, self._run = 0, None
        self.launched = []
        self._pass += 1
        self._timed = self.timing is not None and self._pass % max(int(self.timing_stride), 1) == 0
        # With several attention hops backward still has whole-CU kernels with more workgroups than compute units to
        # launch (the earlier hops' attention backward: one workgroup per sample) after the first buckets complete, and
        # beside a collective's kernel they would need an extra round: everything but the last bucket then goes out as
        # one collective once the bucket before the last is complete (the last hop's small terms).
        self._hold_all = getattr(model, 'hops', 1) > 1

    def _launch(self, s: int, e: int):
        self.launched.append((s, e))
        if self._timed and self.timing is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.timing.setdefault('launch', []).append((4 * (e - s), ev))
        return dist.all_reduce(self._flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def bucket_ready(self, index: int) -> None:
        """Gradient bucket `index` is complete.  Buckets are reduced in index order (the order backward
        completes them); a bucket signalled early waits for its predecessors."""
        if self._fired[index]:
            return
        self._fired[index] = True
        if not self.enabled:
            return
        n = len(self._buckets)
        while self._next < n and self._fired[self._next]:
            i = self._next
            self._next += 1
            s, e = self._buckets[i]
            if i == n - 1 and n > 1:        # the last bucket: on its own, after whatever was held back
                if self._run is not None:
                    self._handles.append(self._launch(*self._run))
                    self._run = None
                self._last_handle = self._launch(s, e)
                continue
            self._run = (self._run[0], e) if self._run is not None else (s, e)
            if (not self._hold_all and 4 * (self._run[1] - self._run[0]) >= self.bucket_bytes) or i >= n - 2:
                self._handles.append(self._launch(*self._run))
                self._run = None

    def finish(self, defer_last: bool = False) -> float:
        """Wait for every bucket (reducing any that was never signalled) -- all but the last one
        when `defer_last` -- and return the factor that turns the summed gradients into the
        global-batch mean."""
        for i, done in enumerate(self._fired):
            if not done:
                self.bucket_ready(i)
        self._wait(self._handles)
        self._handles = []
        if not (defer_last and self.enabled):
            self.finish_last()
        return 1.0 / self.world

    def defer_everything(self) -> float:
        """Start the collectives of any bucket not yet signalled and wait for none of them;
        finish_last() completes them all.  Returns the same factor as finish()."""
        for i, done in enumerate(self._fired):
            if not done:
                self.bucket_ready(i)
        return 1.0 / self.world

    def _wait(self, handles) -> None:
        # RCCL runs a group's collectives in order on one stream: making the compute stream wait for the
        # youngest one covers the older ones, and each wait that is saved is one cross-stream dependency
        # less on the GPU.  Other backends complete their work items independently.
        if not handles:
            return
        pair = None
        if self._timed and self.timing is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        if self._in_order:
            handles[-1].wait()
        else:
            for h in handles:
                h.wait()
        if pair is not None:
            pair[1].record()
            self.timing.setdefault('wait', []).append(pair)

    def finish_last(self) -> None:
        """Wait for whatever finish(defer_last=True) or defer_everything() left in flight."""
        pending = self._handles + ([self._last_handle] if self._last_handle is not None else [])
        self._wait(pending)
        self._handles, self._last_handle = [], None


def reduce_counters(counters: torch.Tensor, process_group=None) -> torch.Tensor:
    """Sum the integer per-question-type count / right counters over ranks; OA / AA
    computed from the result do not depend on the number of ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=process_group)
    return counters
