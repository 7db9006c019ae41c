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


import os

_TRACE_MARKERS = os.environ.get("ALDI_DP_TRACE", "0") == "1"


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


def resolve_exchange(name: str, group=None) -> str:
    """SOLVER.GRAD_EXCHANGE -> the form a reducer runs.  "auto": reduce-scatter + all-gather when the group is the 8 ranks of ONE node (the fully
    connected xGMI mesh: every rank talks to its 7 peers at once, DESIGN.md section 7 / SURVEY 8(e)), one all-reduce per bucket otherwise;
    "all_reduce" / "rs_ag" are taken as given (the key that turns the default back)."""
    name = str(name)
    if name != "auto":
        return name
    try:
        world = dist.get_world_size(group)
    except Exception:
        return "all_reduce"
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    return "rs_ag" if world == 8 and local == 8 else "all_reduce"


class BucketedReducer:
    """all-reduce(SUM) of a flat gradient tensor in the order its pieces become final."""

    MIN_ELEMS = 1 << 18          # pieces smaller than 1 MB wait for the next report / finish(): latency-bound collectives

    _LAUNCH_STREAMS = {}

    def __init__(self, grad: torch.Tensor, group=None, payload: str = "fp32", exchange: str = "all_reduce"):
        """payload "bf16": a piece travels as bf16 (half the bytes per xGMI link: the ring all-reduce over 8 GPUs is per-link
        bound) and is written back into the fp32 buffer afterwards.  The collective then ADDS in bf16: every
        partial sum of the ring (world - 1 of them per element) is rounded to 8 mantissa bits, so the error of an element is up to
        ~(world - 1) * 2^-9 of the partial sums' magnitude, not the single rounding of the payload -- an approximation to opt into;
        the reference's DDP exchanges fp32 master gradients: "fp32" (default) is the exact form.
        exchange "rs_ag": every piece as reduce_scatter_tensor + all_gather_into_tensor on the flat buffer (in place: rank r owns
        the r-th slice of the piece) instead of one all_reduce -- the two halves of the exchange as separate collectives, the form
        SURVEY 5.8 / 8(e) prefers on the fully connected xGMI mesh (each rank talks to its 7 peers at once: 2 * 7/8 of the piece
        over 7 links, against a ring that moves the same bytes over one link at a time); a tail shorter than the world size goes
        through all_reduce.  Same sums (the same pairs are added; the order inside the collective is the library's)."""
        if payload not in ("fp32", "bf16"):
            raise ValueError("gradient payload must be fp32 or bf16")
        if exchange not in ("all_reduce", "rs_ag"):
            raise ValueError("gradient exchange must be all_reduce or rs_ag")
        self.grad, self.group, self.payload, self.exchange = grad, group, payload, exchange
        self.done: List[Tuple[int, int]] = []
        self.pending: List[Tuple[int, int]] = []
        self.works = []
        self.post = []                   # bf16 payload: (lo, hi, buffer) to widen back after the collective
        self.events = []                 # producer events of the pieces not launched yet
        self.finished = False
        # Collectives are issued from a stream of their own that waits only on the PRODUCERS' events (the weight-gradient
        # side stream and the point of the main stream where the group was reported): the main stream never joins the
        # side stream in the middle of the backward, so the dgrad chain and the wgrad kernels keep overlapping under DP.
        self.launch_stream = None
        if grad.is_cuda:
            key = str(grad.device)
            if key not in BucketedReducer._LAUNCH_STREAMS:
                BucketedReducer._LAUNCH_STREAMS[key] = torch.cuda.Stream(device=grad.device)
            self.launch_stream = BucketedReducer._LAUNCH_STREAMS[key]

    def capturable(self) -> bool:
        """True when the collectives may be recorded into the fused step's phase-B hipGraph: RCCL (backend "nccl") launches are
        stream-ordered kernels; gloo's are host calls.  ALDI_DP_GRAPH=0 keeps phase B eager under data parallelism."""
        import os
        if os.environ.get("ALDI_DP_GRAPH", "1") != "1" or not self.grad.is_cuda:
            return False
        return self._stream_ordered()

    def _exchange(self, lo: int, hi: int):
        piece = self.grad[lo:hi]
        if self.payload == "bf16":
            buf = piece.to(torch.bfloat16)
            self.post.append((lo, hi, buf))
            piece = buf
        # RCCL: the plain (async_op = False) call is already asynchronous for the host -- it orders itself on the stream it is issued
        # from (here the launch stream) -- and it is the form a hipGraph can record: an async work handle's wait() inside a capture
        # segfaults in capture_end on this stack (torch 2.10 + RCCL 2.26.6; tools/probes/rccl_capture_probe.py).  gloo's plain call
        # would block the host until the exchange is done, so it keeps the work handle.
        ordered = self._stream_ordered()
        if self.exchange == "rs_ag":
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
            m = piece.numel() // world * world
            w = None
            if m:
                body = piece[:m]
                shard = body[rank * (m // world): (rank + 1) * (m // world)]
                dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.SUM, group=self.group)
                # (RCCL takes the in-place form -- the input is the rank's own slice of the output; gloo gets a copy)
                w = dist.all_gather_into_tensor(body, shard if ordered else shard.clone(), group=self.group, async_op=not ordered)
            if m < piece.numel():
                dist.all_reduce(piece[m:], op=dist.ReduceOp.SUM, group=self.group)
            return None if ordered else w
        if ordered:
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
            return None
        return dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _stream_ordered(self) -> bool:
        if not hasattr(self, "_nccl"):
            try:
                self._nccl = self.grad.is_cuda and dist.get_backend(self.group) == "nccl"
            except Exception:
                self._nccl = False
        return self._nccl

    def _launch(self, lo: int, hi: int):
        if self.launch_stream is not None:
            for ev in self.events:
                self.launch_stream.wait_event(ev)
            with torch.cuda.stream(self.launch_stream):
                if _TRACE_MARKERS:
                    # (profiling aid, ALDI_DP_TRACE=1: an empty kernel on the launch stream right where the bucket's collective is issued -- RCCL
                    # launches NO kernel for an in-place collective of a one-rank group, so on a 1-GPU box this marker is what a kernel trace
                    # can show of the point in the backward at which the bucket's exchange becomes runnable; tools/dp_trace.py)
                    from . import ops
                    ops.noop()
                w = self._exchange(lo, hi)
        else:
            w = self._exchange(lo, hi)
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
        """reduce everything not reduced yet, then make the current stream wait for every collective (idempotent: the fused step
        calls it at the end of a captured phase B, the trainer's `after_backward` again afterwards)."""
        if self.finished:
            return
        if self.launch_stream is not None:
            ev = torch.cuda.Event()
            ev.record()                                  # everything the step has enqueued on the current stream
            self.events.append(ev)
        for lo, hi in complement(self.done, self.grad.numel()):
            self._launch(lo, hi)
        ctx = torch.cuda.stream(self.launch_stream) if self.launch_stream is not None else _Null()
        for w in self.works:
            if w is not None:
                with ctx:
                    w.wait()                             # (the collective's own stream -> the launch stream)
        if self.post:
            with ctx:
                for lo, hi, buf in self.post:
                    self.grad[lo:hi].copy_(buf)          # bf16 -> fp32, behind the collectives on the launch stream
        if self.launch_stream is not None:
            torch.cuda.current_stream().wait_stream(self.launch_stream)
        self.done, self.pending, self.works, self.events, self.post = [], [], [], [], []
        self.finished = True


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
