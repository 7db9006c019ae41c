"""Overlapped gradient exchange for the data-parallel step.

The reference's DDP all-reduces on every micro-step backward (aldi/dropin.py:53).  Here the student gradient is ONE flat
fp32 buffer reduced once per step; with the fused schedule there is a single backward, so the exchange can start while
that backward is still running: the engine reports each group of layers as soon as its weight-gradient kernels are
enqueued (box head first, then RPN/FPN, then res5, res4, res3) and this class launches an asynchronous all-reduce
(RCCL over xGMI through torch.distributed: the collective waits for the work enqueued so far on the compute stream,
later kernels run beside it).  `finish()` reduces whatever was never reported and joins all collectives before the
optimizer reads the buffer.  Sums are linear: same result as one reduction at the end.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def merge_ranges(ranges: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    out: List[Tuple[int, int]] = []
    for lo, hi in sorted(r for r in ranges if r[1] > r[0]):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def complement(ranges: Sequence[Tuple[int, int]], n: int) -> List[Tuple[int, int]]:
    out, pos = [], 0
    for lo, hi in merge_ranges(ranges):
        if lo > pos:
            out.append((pos, lo))
        pos = max(pos, hi)
    if pos < n:
        out.append((pos, n))
    return out


class BucketedReducer:
    """all-reduce(SUM) of a flat gradient tensor in the order its pieces become final."""

    MIN_ELEMS = 1 << 18          # pieces smaller than 1 MB wait for the next report / finish(): latency-bound collectives

    _LAUNCH_STREAMS = {}

    def __init__(self, grad: torch.Tensor, group=None):
        self.grad, self.group = grad, group
        self.done: List[Tuple[int, int]] = []
        self.pending: List[Tuple[int, int]] = []
        self.works = []
        self.events = []                 # producer events of the pieces not launched yet
        # Collectives are issued from a stream of their own that waits only on the PRODUCERS' events (the weight-gradient
        # side stream and the point of the main stream where the group was reported): the main stream never joins the
        # side stream in the middle of the backward, so the dgrad chain and the wgrad kernels keep overlapping under DP.
        self.launch_stream = None
        if grad.is_cuda:
            key = str(grad.device)
            if key not in BucketedReducer._LAUNCH_STREAMS:
                BucketedReducer._LAUNCH_STREAMS[key] = torch.cuda.Stream(device=grad.device)
            self.launch_stream = BucketedReducer._LAUNCH_STREAMS[key]

    def _launch(self, lo: int, hi: int):
        if self.launch_stream is not None:
            for ev in self.events:
                self.launch_stream.wait_event(ev)
            with torch.cuda.stream(self.launch_stream):
                w = dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append(w)
        self.done.append((lo, hi))

    def ready(self, ranges: Sequence[Tuple[int, int]], events=()):
        """`ranges` (element offsets into the flat gradient) will not be written again in this step once `events`
        (recorded on the streams that produce them) have completed.  Without events the collective orders itself after
        the work enqueued so far on the current stream."""
        if self.launch_stream is not None:
            if events:
                self.events += list(events)
            else:
                ev = torch.cuda.Event()
                ev.record()
                self.events.append(ev)
        self.pending = merge_ranges(list(self.pending) + list(ranges))
        keep = []
        for lo, hi in self.pending:
            if hi - lo >= self.MIN_ELEMS:
                self._launch(lo, hi)
            else:
                keep.append((lo, hi))
        self.pending = keep

    def finish(self):
        """reduce everything not reduced yet, then make the current stream wait for every collective."""
        if self.launch_stream is not None:
            ev = torch.cuda.Event()
            ev.record()                                  # everything the step has enqueued on the current stream
            self.events.append(ev)
        for lo, hi in complement(self.done, self.grad.numel()):
            self._launch(lo, hi)
        for w in self.works:
            w.wait()
        if self.launch_stream is not None:
            torch.cuda.current_stream().wait_stream(self.launch_stream)
        self.done, self.pending, self.works, self.events = [], [], [], []
