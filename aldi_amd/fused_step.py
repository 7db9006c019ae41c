"""The fused ALDI iteration of the R50-FPN engine as three explicit phases, replayable as HIP graphs.

    phase A  (device, label-free)   student trunk + RPN head + proposals | teacher inference + pseudo-labels (own stream)
                                    -> anchor matching, ROI candidate lists, list lengths -> pinned host memory
    host                            the ONE synchronisation of the step; every `torch.randperm` draw of the reference
                                    schedule (RPN / ROI sampling per micro-step, the teacher's replay, the fresh
                                    RPN-distillation sample), in the reference's order -> one pinned upload buffer
    phase B  (device)               sampled labels / ROIs, box heads (student + teacher), all losses, the single backward

What the phases replace: the body of `run_model_labeled_unlabeled` (reference aldi/trainer.py:28-117) with the student
micro-steps batched into one trunk pass, `ALDIDistiller._distill_forward` (aldi/distill.py:144-168) and the
loss-dict arithmetic.  Same loss keys, values, 1/accum scaling, `v*0` masking and global-RNG stream as the sequential
driver (tests/test_engine_gpu.py::test_fused_step_equals_sequential).

Neither device phase reads anything from the host except through fixed buffers, and no launch argument changes from
step to step, so each phase is captured ONCE into a hipGraph (`torch.cuda.CUDAGraph`: torch's allocator keeps the
captured tensors resident) and replayed: the Python sequencer then issues ~10 launches per step instead of ~450
(SOLVER.STEP_GRAPH, on by default in bench.py).  Phase B's shape depends on the number of sampled ROIs (512 per image
unless an image has fewer candidates): one graph per distinct row tuple.  Under data parallelism phase B stays eager (its
backward launches the overlapped gradient exchange, reduce.BucketedReducer)."""
from __future__ import annotations

import contextlib
import os
import time
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

from . import ops
from .arch import pad_to
from .engine import GMAX, RCNN, Ctx
from .structures import as_record


def _pinned(nbytes: int) -> torch.Tensor:
    return torch.empty(max(nbytes, 16), dtype=torch.uint8).pin_memory()


class _Packed:
    """several small int32 / float32 host arrays in ONE pinned buffer with a device mirror: a single asynchronous copy"""

    def __init__(self, spec, device):
        self.spec, off = {}, 0
        for name, shape, dtype in spec:
            n = 1
            for s_ in shape:
                n *= s_
            nb = n * 4
            self.spec[name] = (off, nb, tuple(shape), dtype)
            off += (nb + 15) // 16 * 16
        self.host = _pinned(off)
        self.host.zero_()
        self.words = self.host.numpy().view("int32")         # same storage: scalar stores without a tensor op each
        self.dev = torch.zeros(self.host.numel(), dtype=torch.uint8, device=device)

    def word0(self, name) -> int:
        return self.spec[name][0] // 4

    def h(self, name) -> torch.Tensor:
        o, nb, shape, dt = self.spec[name]
        return self.host[o:o + nb].view(dt).view(shape)

    def d(self, name) -> torch.Tensor:
        o, nb, shape, dt = self.spec[name]
        return self.dev[o:o + nb].view(dt).view(shape)

    def upload(self):
        self.dev.copy_(self.host, non_blocking=True)


class _GroupSGD:
    """`grad_ready` receiver of the single-GPU fused step: SGD over the reported ranges of the flat parameter buffer, on its own
    stream behind the producers' events; `finish` covers whatever was not reported and joins the main stream"""
    def __init__(self, eng, hyper):
        self.eng, self.hyper, self.done = eng, hyper, []
        if getattr(eng, "_sgd_side", None) is None:          # (False: on the caller's stream -- ALDI_SGD_STREAM=0, single-stream profiles)
            eng._sgd_side = torch.cuda.Stream(device=eng.device) if os.environ.get("ALDI_SGD_STREAM", "1") == "1" else False
        self.stream = eng._sgd_side or torch.cuda.current_stream()

    @staticmethod
    def _merge(ranges):
        out = []
        for lo, hi in sorted((int(a), int(b)) for a, b in ranges):
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return out

    def ready(self, ranges, evs):
        for ev in evs:
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            for lo, hi in self._merge(ranges):
                self.eng.wts.sgd_range_dev(lo, hi, self.hyper)
        self.done += [(int(a), int(b)) for a, b in ranges]

    def finish(self):
        main = torch.cuda.current_stream()
        rest, at = [], 0
        for lo, hi in self._merge(self.done) + [[self.eng.wts.layout.n_train, self.eng.wts.layout.n_train]]:
            if lo > at:
                rest.append((at, min(lo, self.eng.wts.layout.n_train)))
            at = max(at, hi)
        if rest:
            self.stream.wait_stream(main)                  # (the backward has joined its weight-gradient stream by now)
            with torch.cuda.stream(self.stream):
                for lo, hi in rest:
                    self.eng.wts.sgd_range_dev(lo, hi, self.hyper)
        main.wait_stream(self.stream)


class FusedStep:
    def __init__(self, trainer):
        self.tr = trainer
        self.static: Dict[tuple, SimpleNamespace] = {}
        self.steps_done = 0
        self.pool = None
        self.graph_enabled = bool(trainer.model.cfg.SOLVER.get("STEP_GRAPH", True)) and os.environ.get("ALDI_STEP_GRAPH", "1") == "1"
        # student + teacher trunk / RPN head as ONE launch per layer.  Off by default: measured 11.98 vs 11.20 ms/step -- what the
        # shared launches save (~0.5 ms of fixed per-launch cost) is less than what the lost concurrency costs (the teacher's
        # latency-bound proposal / box-head / detection chain then runs alone after the paired trunk, and the EMA tick before it)
        self.pair_forward = os.environ.get("ALDI_PAIR_FORWARD", "0") == "1"
        self.teacher_first = os.environ.get("ALDI_TEACHER_FIRST", "0") == "1"
        self.interleave = os.environ.get("ALDI_INTERLEAVE", "1") == "1"
        self.spin_wait = os.environ.get("ALDI_SPIN_WAIT", "1") == "1"
        # stem + res2 of batch k + 1 under step k's proposal chain (needs the next batch: trainer._fetch_batch); ALDI_PIPELINE_PREFIX=0 for A/B runs
        self.pipeline = (bool(trainer.model.cfg.SOLVER.get("PIPELINE_PREFIX", False)) or os.environ.get("ALDI_PIPELINE_PREFIX") == "1") and os.environ.get("ALDI_PIPELINE_PREFIX") != "0"
        self.prefix_at = os.environ.get("ALDI_PREFIX_AT", "b")     # "a": behind the RPN head, under the proposal chain; "b": beside the head of phase B
        self._parity = 0
        self._pslots: Dict[tuple, SimpleNamespace] = {}
        self._pre_ready = None               # (slot, the list objects of the batch whose prefix that slot holds)
        self.warmup = 3                     # eager steps before capturing (lazy initialisation: anchors, dgrad weights, workspaces)
        self.dp_graph_ok = True             # cleared if recording phase B together with its collectives ever fails
        self.stats = dict(captures=0, replays_a=0, replays_b=0, eager=0)

    # ------------------------------------------------------------------------------------------------ phase 0: inputs
    def _stage_images(self, S, slot: str, images):
        """uint8 images (host or device) -> the static padded batch of this slot (same storage every step)"""
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
        Hs, Ws = pad_to(max(s[0] for s in sizes), 32), pad_to(max(s[1] for s in sizes), 32)
        key = (slot, len(images), Hs, Ws)
        st = S.bufs.get(key)
        dev = self.eng.device
        if st is None:
            st = SimpleNamespace(img=torch.zeros((len(images), 3, Hs, Ws), dtype=torch.uint8, device=dev), sizes=None,
                                 hw=torch.zeros((len(images), 2), dtype=torch.int32, device=dev))
            S.bufs[key] = st
        if st.sizes != sizes:
            if st.sizes is not None:
                st.img.zero_()                               # smaller images than last step: the padding must read zero
            st.hw.copy_(torch.tensor(sizes, dtype=torch.int32))
            st.sizes = sizes
        if all(im.is_cuda and im.dtype == torch.uint8 and im.is_contiguous() for im in images) and len(images) <= 16:
            ops.stage_images(images, st.img)                 # one launch for the batch
        else:
            for i, im in enumerate(images):
                st.img[i, :, : sizes[i][0], : sizes[i][1]].copy_(im, non_blocking=True)
        return st

    def _stage_gt(self, S, rows, N: int):
        """ground truth of ALL N images of the fused batch in static device buffers through one pinned upload: the labeled images'
        records (rows = [(image index, record)]), zero rows for the others -- the distillation chunk's rows are then written by
        the teacher's detection kernel itself (its pseudo-labels), so the matcher reads one buffer and nothing is concatenated"""
        if S.gt_pack is None or S.gt_pack.n != N:
            S.gt_pack = _Packed([("boxes", (N, GMAX, 4), torch.float32), ("classes", (N, GMAX), torch.int32), ("count", (N,), torch.int32)], self.eng.device)
            S.gt_pack.n = N
        P = S.gt_pack
        gb, gc, cnt = P.h("boxes"), P.h("classes"), P.h("count")
        gb.zero_(); gc.zero_(); cnt.zero_()
        for i, inst in rows:
            b = inst["gt_boxes"]
            b = b.tensor if hasattr(b, "tensor") else b
            g = int(b.shape[0])
            if g > GMAX:
                raise ValueError(f"more than {GMAX} GT boxes in one image")
            if g:
                gb[i, :g] = b.reshape(-1, 4).to(torch.float32).cpu()
                gc[i, :g] = inst["gt_classes"].to(torch.int32).cpu()
            cnt[i] = g
        P.upload()
        return {"boxes": P.d("boxes"), "classes": P.d("classes"), "count": P.d("count")}

    # ------------------------------------------------------------------------------------------------ cross-step pipelining of the frozen prefix
    def _fused_images(self, data, do_align, do_distill):
        """the student images of an iteration in the fused batch's order (what `run` assembles for the current one)"""
        from .trainer import plan_micro_steps
        plan = plan_micro_steps(*data, do_align=do_align, do_distill=do_distill)
        return [d["image"] for row in plan for d in row.data]

    def _prefix_slot(self, par: int, images):
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
        Hs, Ws = pad_to(max(s[0] for s in sizes), 32), pad_to(max(s[1] for s in sizes), 32)
        k = (par, len(images), Hs, Ws)
        sl = self._pslots.get(k)
        if sl is None:
            dev = self.eng.device
            Hc, Wc = Hs // 2, Ws // 2
            sl = SimpleNamespace(img=torch.zeros((len(images), 3, Hs, Ws), dtype=torch.uint8, device=dev), sizes=None,
                                 hw=torch.zeros((len(images), 2), dtype=torch.int32, device=dev),
                                 out=torch.empty((len(images), (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, 256), dtype=torch.bfloat16, device=dev), key=k)
            self._pslots[k] = sl
        return sl, sizes

    def _stage_into_slot(self, sl, images, sizes):
        if sl.sizes != sizes:
            if sl.sizes is not None:
                sl.img.zero_()
            sl.hw.copy_(torch.tensor(sizes, dtype=torch.int32))
            sl.sizes = sizes
        if all(im.is_cuda and im.dtype == torch.uint8 and im.is_contiguous() for im in images) and len(images) <= 16:
            ops.stage_images(images, sl.img)
        else:
            for i, im in enumerate(images):
                sl.img[i, :, : sizes[i][0], : sizes[i][1]].copy_(im, non_blocking=True)

    def _prefix_begin(self, S, par, images, next_data, do_align, do_distill):
        """S.stu = this parity's slot holding the current student images AND their res2 output (computed by the previous step's phase A when the
        trainer had handed this batch over as `next_data`; otherwise computed here, in order); S.nxt = the other parity's slot with the next
        iteration's images staged, whose prefix this step's phase A computes.  Nothing is cached: every step's prefix is computed once, from
        that step's images -- one step early."""
        eng = self.eng
        cur, sizes = self._prefix_slot(par, images)
        ready = self._pre_ready
        self._pre_ready = None
        have = (ready is not None and ready[0] is cur and len(ready[1]) == len(images) and all(a is b for a, b in zip(ready[1], images))
                and cur.sizes == sizes)
        if not have:
            self._stage_into_slot(cur, images, sizes)
            with torch.no_grad():
                eng._drive(eng.trunk_steps(cur.img, sizes, False, prefix_out=cur.out))
            self.stats["prefix_inline"] = self.stats.get("prefix_inline", 0) + 1
        else:
            self.stats["prefix_ahead"] = self.stats.get("prefix_ahead", 0) + 1
        S.stu = cur
        S.nxt = None
        S.nxt_images = None
        nxt_slot, _ = self._prefix_slot(par ^ 1, images)            # what this S's graphs are recorded with: the same shapes, the other parity
        S.nxt = nxt_slot
        if next_data is not None:
            try:
                nimg = self._fused_images(next_data, do_align, do_distill)
            except Exception:
                nimg = None
            if nimg is not None and len(nimg) == len(images):
                nsizes = [(int(im.shape[1]), int(im.shape[2])) for im in nimg]
                if nsizes == sizes:                                 # (the stem kernel's launch carries the image sizes: same sizes, or no look-ahead)
                    self._stage_into_slot(nxt_slot, nimg, nsizes)
                    S.nxt_images = nimg
        if nxt_slot.sizes is None:                                   # never staged: the recorded prefix pass still runs on it (zeros)
            nxt_slot.sizes = sizes
            nxt_slot.hw.copy_(torch.tensor(sizes, dtype=torch.int32))

    def _prefix_stream(self):
        if not hasattr(self, "_pfx_stream"):
            self._pfx_stream = torch.cuda.Stream(device=self.eng.device, priority=int(os.environ.get("ALDI_PREFIX_PRIO", "0")))
        return self._pfx_stream

    # ------------------------------------------------------------------------------------------------ phase A
    def _phase_a(self, S):
        eng, teng, dist_ = self.eng, self.teng, self.tr.distiller
        main = torch.cuda.current_stream()
        if os.environ.get("ALDI_PROBE_SPIN_CYCLES"):         # (probe: a busy-wait kernel in front of the fork)
            torch.cuda._sleep(int(os.environ["ALDI_PROBE_SPIN_CYCLES"]))
        ev0 = torch.cuda.Event()
        ev0.record(main)                                     # the teacher may start here: beside the student's trunk
        N = S.N
        stu = S.stu
        tea, tside = S.tea, S.tside
        shapes, geom, anchors = eng.geometry(stu.img.shape[2], stu.img.shape[3])
        pair = S.distill and self.pair_forward and type(eng) is RCNN and type(teng) is RCNN
        inter = S.distill and self.interleave and not pair and tside is not None and type(eng) is RCNN and type(teng) is RCNN
        tcx = None
        pre_in = S.stu.out if getattr(S, "pipe", False) else None
        if inter:
            # Student on the main stream, teacher on its own, their launches ISSUED alternately layer by layer: both branches of the
            # captured graph are fed from the first microsecond (captured one after the other, the second branch's first node
            # reaches its queue when the host has submitted the whole first branch: the teacher's serial chain of small launches
            # -- which the student's anchor matching waits for -- then starts 2 ms late and runs its last 1.5 ms alone)
            tside.wait_event(ev0)
            if S.ema_mode is not None:
                with torch.cuda.stream(tside):
                    teng.wts.ema_from(eng.wts, S.ema_alpha, copy_only=S.ema_mode == "copy")
            with torch.no_grad():
                c, tcx = RCNN.drive_pair(eng, eng.trunk_steps(stu.img, stu.sizes, True, pre=pre_in), teng, teng.trunk_steps(tea.img, tea.sizes, False), streams=(main, tside))
                RCNN.drive_pair(eng, eng.rpn_head_steps(c, True), teng, teng.rpn_head_steps(tcx, False), streams=(main, tside))
        elif pair:
            # Student (N = 4) and teacher (N = 2) go through the same layers with different weights: ONE launch per layer for both
            # (engine.RCNN.drive_pair -> aldi_conv_igemm_group) instead of two half-empty ones on two streams.  The EMA tick has
            # to come first then (the teacher's weights are read from the first layer on).
            if S.ema_mode is not None:
                teng.wts.ema_from(eng.wts, S.ema_alpha, copy_only=S.ema_mode == "copy")
            c, tcx = RCNN.drive_pair(eng, eng.trunk_steps(stu.img, stu.sizes, True, pre=pre_in), teng, teng.trunk_steps(tea.img, tea.sizes, False))
            RCNN.drive_pair(eng, eng.rpn_head_steps(c, True), teng, teng.rpn_head_steps(tcx, False))
        else:
            if S.distill and self.teacher_first and tside is not None:
                # issue order inside the captured graph: the teacher's chain (trunk -> proposals -> box head -> detections -> pseudo
                # labels) is the longer dependency chain of phase A -- the student's anchor matching waits for it -- so its nodes go
                # first; the student's trunk is enqueued right behind and fills the chip beside it
                tside.wait_event(ev0)
                with torch.cuda.stream(tside), torch.no_grad():
                    if S.ema_mode is not None:
                        teng.wts.ema_from(eng.wts, S.ema_alpha, copy_only=S.ema_mode == "copy")
                    tc_early = teng.inference(None, dist_.pseudo_label_threshold, staged=(tea.img, tea.sizes, tea.hw), pl_out=S.pl_out)
            # (the flat-container engines override trunk(), not trunk_steps())
            c = eng._drive(eng.trunk_steps(stu.img, stu.sizes, True, pre=pre_in)) if pre_in is not None else eng.trunk(stu.img, stu.sizes, save=True)
            eng.rpn_head(c, save=True)
        c.N, c.sizes, c.hw, c.geom, c.anchors, c.shapes = N, stu.sizes, stu.hw, geom, anchors, shapes
        pfx = None
        if getattr(S, "pipe", False) and self.prefix_at == "a":
            # the NEXT iteration's frozen prefix: from here on the student's chain is a handful of latency-bound launches (keys, top-k, NMS, merge,
            # ROI preparation: ~0.34 ms with the chip idle) -- dense work that depends on nothing of this step runs under it
            pfx = self._prefix_stream()
            pfx.wait_stream(main)
            with torch.cuda.stream(pfx), torch.no_grad():
                eng._drive(eng.trunk_steps(S.nxt.img, S.nxt.sizes, False, prefix_out=S.nxt.out))
        # proposal generation (top-k, NMS: latency-bound, a handful of workgroups) beside anchor matching on the second stream
        side = eng._wgrad_stream()
        ev_props = None

        def housekeeping():
            # nothing before phase B reads these: they must not sit between the proposals and the list lengths the host is waiting for
            # (on the side stream BEHIND the proposals they did -- the join below waited for them: 107 us of the critical path in the
            # kernel trace, profiles/r04_kernel_stats.txt); now they run while the host draws the samples
            if S.lazy_wt:
                eng.wts._refresh_wt()                       # the backward's dgrad weights of the weights SGD just wrote
            if S.zero_grad:
                eng.wts.grad.zero_()                        # 164 MB
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                c.props, c.prop_scores, c.prop_count = eng.proposals(c, geom, anchors, stu.hw, N, training=True)
                ev_props = torch.cuda.Event()
                ev_props.record(side)
                housekeeping()
        else:
            c.props, c.prop_scores, c.prop_count = eng.proposals(c, geom, anchors, stu.hw, N, training=True)
        tc = None
        if S.distill:
            # The teacher's inference (N = 2, mostly small launches) runs on its own stream beside the student's label-free work
            # and is ENQUEUED after it (issued first its launches would run alone while the student's are still being queued).
            def teacher_pass():
                if pair or inter:
                    return teng.inference_heads(tcx, tea.img, tea.sizes, tea.hw, dist_.pseudo_label_threshold, pl_out=S.pl_out)
                if S.ema_mode is not None:                   # the EMA tick of this iteration (aldi/trainer.py:242-246), beside the student's forward
                    teng.wts.ema_from(eng.wts, S.ema_alpha, copy_only=S.ema_mode == "copy")
                return teng.inference(None, dist_.pseudo_label_threshold, staged=(tea.img, tea.sizes, tea.hw), pl_out=S.pl_out)
            if tside is not None and self.teacher_first and not pair and not inter:      # (the knob applies to the non-interleaved issue order only)
                tc = tc_early
                main.wait_stream(tside)
            elif tside is not None:
                if pair:
                    tside.wait_stream(main)
                elif not inter:
                    tside.wait_event(ev0)
                with torch.cuda.stream(tside), torch.no_grad():
                    tc = teacher_pass()
                main.wait_stream(tside)
            else:
                with torch.no_grad():
                    tc = teacher_pass()
        # ground truth of all chunks: the staged buffers (labels uploaded in phase 0, zero rows for unlabeled chunks, the
        # distillation chunk's rows written by the teacher's detection kernel above)
        gt = S.gt_all
        c.gt = gt
        hook = getattr(self, "discrete_inputs_hook", None)
        if hook is not None:
            # parity tooling (tests/test_configs_gpu.py): called with everything the DISCRETE stages below read -- the student's proposals
            # and the ground truth incl. the teacher's pseudo-label rows -- so that a run in another arithmetic mode can be given the same
            # ones (matching / sampling are discontinuous in them); eager steps only, nothing on the product path sets it
            if side is not None:
                main.wait_stream(side)
            hook(S, c, tc)
        _, matched, lists, counts = eng.rpn_match(geom, anchors, gt, N)
        c.rpn_matched, c.rpn_lists, c.rpn_counts = matched, lists, counts
        S.h_counts.view(torch.int32)[: 2 * N].copy_(counts.view(-1), non_blocking=True)      # (the anchor lists' lengths: ready long before the proposals)
        if ev_props is not None:
            if os.environ.get("ALDI_HOUSEKEEPING_LATE", "1") == "1":
                main.wait_event(ev_props)                    # the proposals only: the side stream's housekeeping is joined behind the hand-over
            else:
                main.wait_stream(side)                       # (A/B: the round-3 order)
        # ... and the two engines' error words ride along (bits set by this phase, or by the previous step's phase B): the host
        # raises on them right after the hand-over instead of training on silently wrong gradients
        errs = [eng.err.view(-1), (teng if teng is not None else eng).err.view(-1)]
        prep = eng._roi_prepare(c.props, c.prop_count, gt, N, tail=(errs[0], errs[1]))
        if prep.get("counts_tail") is not None:
            # no concatenation launch: the anchor lists' lengths left above, the hand-over is the ROI lists' lengths with the error words the
            # preparation kernel appended
            S.h_counts.view(torch.int32)[2 * N: 4 * N + 2].copy_(prep["counts_tail"], non_blocking=True)
        else:
            S.h_counts.copy_(torch.cat([counts.view(-1), prep["counts"].view(-1)] + errs).view(torch.uint8), non_blocking=True)
        if side is not None:
            main.wait_stream(side)                           # (phase B starts behind the housekeeping)
        else:
            housekeeping()
        if pfx is not None:
            main.wait_stream(pfx)
        return SimpleNamespace(c=c, tc=tc, prep=prep)

    # ------------------------------------------------------------------------------------------------ host phase
    def _hooks_are_standard(self) -> bool:
        """the only forward pre-hooks on the two roi_heads are the distiller's own (ManualSeed on both, ReplaceProposalsOnce on the
        teacher: aldi/distill.py:131-138): then the host phase can run as one scripted C call instead of firing Python hooks"""
        dist_, model = self.tr.distiller, self.tr.model
        seeder = getattr(dist_, "seeder", None)
        if seeder is None or model.roi_heads.pre_hooks != [seeder]:
            return False
        if self.teacher is not None:
            rep = getattr(dist_, "teacher_proposal_replacer", None)
            if self.teacher.roi_heads.pre_hooks != [seeder, rep] or rep is None or rep.proposals is not None:
                return False
        return True

    def _wait_counts(self, S):
        """block until phase A's last operation -- the copy of the list lengths into pinned memory -- has landed.  The host polls
        the buffer itself (the lengths overwrite a -1 fill; a stream synchronize wakes up tens of microseconds later, with the
        device idle meanwhile); anything unexpected falls back to the synchronize, which also surfaces device errors."""
        cnt = getattr(S, "h_counts_np", None)
        if cnt is not None and self.spin_wait:
            t_end = time.perf_counter() + 2.0
            while (cnt < 0).any():
                if time.perf_counter() > t_end:
                    break
            else:
                return
        torch.cuda.current_stream().synchronize()

    def _scripted(self) -> bool:
        """the host phase as one C call (aldi_torch_rng_script) instead of Python hooks + torch.randperm"""
        from . import engine as E
        if E._FAST_RANDPERM is None:
            E.randperm_prefix(1, 1)                            # (first use verifies the C generator against torch.randperm; draws nothing)
        return self._hooks_are_standard() and bool(E._FAST_RANDPERM) and os.environ.get("ALDI_HOST_RNG_SCRIPT", "1") == "1"

    def _prefetch_draws(self, S, n_anchors: int):
        """While the device runs phase A: pre-generate the Mersenne streams the draws will come from -- the one continuing the
        global CPU generator, the seeder's current seed, and the seed it will draw next (aldi/distill.py:148-150; Python's
        `random` is peeked, not advanced).  The list lengths are not known yet, only the streams they index into; with them the
        host phase between the two device phases no longer skips through ~430 state refills per 268k-entry negative list."""
        if not self._scripted() or os.environ.get("ALDI_HOST_RNG_PREFETCH", "1") != "1":
            return
        import ctypes as C
        import random
        from . import _lib as L
        seeds = [int(self.tr.distiller.seeder.seed)]
        if S.distill:
            keep = random.getstate()
            for _ in range(min(S.nk, 6)):                       # what ManualSeed.reset_seed will draw, once per distillation micro-step
                seeds.append(random.randint(0, 2**32 - 1))      # (aldi/helpers.py:21-23; the library keeps at most 8 streams)
            random.setstate(keep)
        depth = S.N * (n_anchors + 2 * 4096)                   # every image's RPN lists + ROI lists in ONE segment: an upper bound
        st = torch.get_rng_state()
        L.call("aldi_torch_rng_prefetch", st.data_ptr(), (C.c_long * len(seeds))(*seeds), len(seeds), depth)

    def _draws_prepare(self, S):
        """what the native host phase needs that does NOT depend on the list lengths, done while the device still runs phase A: the seeds the
        `ManualSeed` hooks will use (Python's `random` advances here, as in the reference: aldi/distill.py:148-150 -- nothing else draws from it
        between the two phases), the argument arrays, the CPU generator's state.  None when the host phase is not the one native call."""
        if not (self._scripted() and os.environ.get("ALDI_HOST_DRAWS_C", "1") == "1"):
            return None
        import ctypes as C
        seeder = self.tr.distiller.seeder
        seeds = [int(seeder.seed)]
        for _ in range(S.nk):
            seeder.reset_seed()
            seeds.append(int(seeder.seed))
        if getattr(S, "chunk_arr", None) is None:
            flat = []
            for ch in S.chunks:
                flat += [1 if ch["kind"] == "distill" else 0, ch["n0"], ch["n1"]]
            U = S.up
            S.chunk_arr = (C.c_int * len(flat))(*flat)
            S.word0_arr = (C.c_int * 8)(*[U.word0(k) for k in ("rsel", "rnsel", "osel", "onsel", "row_off", "dsel", "dnsel", "nvf")])
            S.rows_arr = (C.c_int * S.N)()
        from . import _lib as L
        L.call("aldi_torch_rng_prefetch_wait")                 # (the stream fillers of _prefetch_draws: ~1 ms of a 4.5 ms phase A; reaped here, not between the phases)
        prep = SimpleNamespace(seeds=(C.c_long * len(seeds))(*seeds), n_seeds=len(seeds), st=torch.get_rng_state())
        # A rehearsal with the PREVIOUS iteration's list lengths, on a copy of the generator and into a scratch buffer: the real call's few
        # thousand draws land in nearly the same places of the freshly written streams (the lengths move by tens of entries from step to step),
        # and its tables and code are in this core's caches when the lengths arrive -- measured 100 -> 46 us for the call between the phases.
        prev = getattr(S, "prev_counts", None)
        if prev is not None and os.environ.get("ALDI_HOST_DRAWS_REHEARSAL", "1") == "1":
            P_ = self.eng.p
            if getattr(S, "scratch_words", None) is None:
                import numpy as np
                S.scratch_words = np.zeros(S.up.words.shape[0], dtype=np.int32)
                S.scratch_rows = (C.c_int * S.N)()
            st_copy = prep.st.clone()
            L.call("aldi_step_draws", st_copy.data_ptr(), prev.ctypes.data, S.N, S.chunk_arr, len(S.chunks), prep.seeds, prep.n_seeds,
                   P_.rpn_batch, int(P_.rpn_batch * P_.rpn_pos_frac), P_.roi_batch, int(P_.roi_batch * P_.roi_pos_frac), S.scratch_words.ctypes.data,
                   S.word0_arr, S.scratch_rows, 4)
        return prep

    def _host_draws(self, S, A, prep=None):
        """every sampling draw of the iteration on the global CPU generator, in the reference's order, chunk by chunk =
        micro-step by micro-step of the reference schedule (aldi/trainer.py:51-52,86-89):
        the RPN sample (two randperm per image), `torch.manual_seed(seed)` by the roi_heads pre-hook, the ROI sample; a distillation
        micro-step starts with the teacher's eval inference re-seeding with the CURRENT seed and the seeder drawing a new one
        (aldi/distill.py:148-150), and ends with the teacher's train-mode forward repeating the ROI draws under the same seed and
        `get_rpn_losses` drawing a fresh RPN sample (aldi/distill.py:160-162,200-202)."""
        eng, dist_, model = self.eng, self.tr.distiller, self.tr.model
        N = S.N
        P_ = eng.p
        RPN_BATCH, RPN_POS_FRAC, ROI_BATCH, ROI_POS_FRAC = P_.rpn_batch, P_.rpn_pos_frac, P_.roi_batch, P_.roi_pos_frac
        both = S.h_counts.view(torch.int32).tolist()
        rpn_counts = [both[2 * i: 2 * i + 2] for i in range(N)]
        roi_counts = [both[2 * N + 2 * i: 2 * N + 2 * i + 2] for i in range(N)]
        U = S.up
        from . import engine as E
        scripted = self._scripted()
        script: List[int] = []
        hw = U.words
        nv = U.word0("nvf")

        def finish(rows):
            r0 = 0
            for ch in S.chunks:
                n0, n1 = ch["n0"], ch["n1"]
                r1 = r0 + sum(rows[n0:n1])
                ch["r0"], ch["r1"] = r0, r1
                ch["rpn_counts"], ch["roi_counts"] = rpn_counts[n0:n1], roi_counts[n0:n1]
                r0 = r1
            nvf = [(int(hw[nv + 2 * k]), int(hw[nv + 2 * k + 1])) for k in range(S.nk)]
            return SimpleNamespace(rows=rows, R=sum(rows), nvf=nvf, key=tuple(rows))
        if scripted and os.environ.get("ALDI_HOST_DRAWS_C", "1") == "1":
            # the whole host phase as ONE native call (aldi_step_draws): same draws, same order, none of the Python below
            from . import _lib as L
            if prep is None:
                prep = self._draws_prepare(S)
            L.call("aldi_step_draws", prep.st.data_ptr(), S.h_counts.data_ptr(), N, S.chunk_arr, len(S.chunks), prep.seeds, prep.n_seeds,
                   P_.rpn_batch, int(P_.rpn_batch * P_.rpn_pos_frac), P_.roi_batch, int(P_.roi_batch * P_.roi_pos_frac), U.host.data_ptr(), S.word0_arr,
                   S.rows_arr, 4)
            out = finish(list(S.rows_arr))
            out.rng_state = prep.st                          # the generator's state after the draws: installed by the caller, behind phase B's launch
            S.prev_counts = S.h_counts_np[: 4 * N].copy()    # (the next iteration's rehearsal)
            return out

        def sample(name, nname, row0, counts, batch, frac):
            """subsample_labels for the images `row0 ...`: positives then negatives, two randperm per image"""
            per = []
            if not scripted:
                a, b, hh = eng._sample_host(counts, batch, frac)
                if name != "-":
                    U.h(name)[row0:row0 + len(counts)], U.h(nname)[row0:row0 + len(counts)] = a, b
                return hh
            keep = name != "-"
            o0 = U.word0(name) if keep else 0
            n0_ = U.word0(nname) if keep else 0
            for i, (npos, nneg) in enumerate(counts):
                num_pos = min(npos, int(batch * frac))
                num_neg = min(nneg, batch - num_pos)
                base = o0 + (row0 + i) * 2 * batch
                script.extend((0, npos, num_pos, base if keep else -1, 0, nneg, num_neg, base + batch if keep else -1))
                if keep:
                    hw[n0_ + 2 * (row0 + i)] = num_pos
                    hw[n0_ + 2 * (row0 + i) + 1] = num_neg
                per.append([num_pos, num_neg])
            return per

        def reseed():
            if scripted:
                script.extend((1, dist_.seeder.seed, 0, 0))
            else:
                torch.manual_seed(dist_.seeder.seed)
        rows: List[int] = []
        k = d0 = 0
        for ch in S.chunks:
            n0, n1 = ch["n0"], ch["n1"]
            distill = ch["kind"] == "distill"
            if distill:
                if scripted:
                    reseed()                               # the teacher's eval inference fired ManualSeed with the current seed
                else:
                    self.teacher.roi_heads.fire_pre()
                dist_.seeder.reset_seed()
            sample("rsel", "rnsel", n0, rpn_counts[n0:n1], RPN_BATCH, RPN_POS_FRAC)
            if scripted:
                reseed()                                   # ManualSeed pre-hook of the student's roi_heads (aldi/helpers.py:25-26)
            else:
                model.roi_heads.fire_pre()
            oh = sample("osel", "onsel", n0, roi_counts[n0:n1], ROI_BATCH, ROI_POS_FRAC)
            rows += [x + y for x, y in oh]
            if distill:
                reseed()
                sample("-", None, 0, roi_counts[n0:n1], ROI_BATCH, ROI_POS_FRAC)            # the teacher's identical ROI draws (aldi/distill.py:160-162)
                dh = sample("dsel", "dnsel", d0, rpn_counts[n0:n1], RPN_BATCH, RPN_POS_FRAC)     # fresh sample of get_rpn_losses (aldi/distill.py:200-202)
                hw[nv + 2 * k], hw[nv + 2 * k + 1] = sum(x + y for x, y in dh), sum(x for x, _ in dh)
                k += 1
                d0 += n1 - n0
        off, ro = 0, U.word0("row_off")
        for i, r in enumerate(rows):
            hw[ro + i] = off
            off += r
        if scripted and script:
            import ctypes as C
            from . import _lib as L
            arr = (C.c_long * len(script))(*script)
            st = torch.get_rng_state()
            L.call("aldi_torch_rng_script", st.data_ptr(), arr, len(script) // 4, U.host.data_ptr(), 4)
            torch.set_rng_state(st)
        return finish(rows)

    # ------------------------------------------------------------------------------------------------ phase B
    def _phase_b(self, S, A, Hst):
        eng, teng, dist_, model = self.eng, self.teng, self.tr.distiller, self.tr.model
        c, tc, prep = A.c, A.tc, A.prep
        dev = eng.device
        main = torch.cuda.current_stream()
        N = S.N
        U = S.up
        if os.environ.get("ALDI_PROBE_SPIN_CYCLES"):         # (probe, as in phase A: lets a tracer's slow node submission finish before the phase runs)
            torch.cuda._sleep(int(os.environ["ALDI_PROBE_SPIN_CYCLES"]))
        U.upload()
        pfx = None
        if getattr(S, "pipe", False) and self.prefix_at == "b":
            # the NEXT iteration's frozen prefix beside the head of phase B: sample scatter, RoIAlign, the box head's forward, the loss kernels and
            # the box head's backward are ~0.5 ms of launches that fill a fraction of the chip (profiles/r05_timeline_spin.txt)
            pfx = self._prefix_stream()
            pfx.wait_stream(main)
            with torch.cuda.stream(pfx), torch.no_grad():
                eng._drive(eng.trunk_steps(S.nxt.img, S.nxt.sizes, False, prefix_out=S.nxt.out))
        sumA = c.anchors.shape[0]
        labels = torch.empty((N, sumA), dtype=torch.int32, device=dev)
        RPN_BATCH = eng.p.rpn_batch
        c.rpn_labels = labels
        # The RPN side of the losses reads phase A's head outputs and the sampled labels only.  Its buffers are allocated HERE (on the main
        # stream, before anything of phase B ran) and handed, with an event that marks the upload, to the auxiliary stream: label scatter,
        # the distillation chunks' fresh RPN sample, the zero fills and (engine.backward_fused) the RPN loss kernels run there beside the box
        # head's forward instead of on the chain into the backward.
        aux = eng._aux_stream() if (c.get("head_flat") is not None and os.environ.get("ALDI_RPN_LOSS_EARLY", "1") == "1") else None
        dls = [(ch, torch.empty((ch["n1"] - ch["n0"], sumA), dtype=torch.int32, device=dev)) for ch in S.chunks if ch["kind"] == "distill"]
        c.rpn_early = None
        if aux is not None:
            Rn = sum(Hst.rows)
            c.rpn_early = dict(arena=torch.empty(2 + 8 * len(S.chunks), dtype=torch.float32, device=dev), gf=torch.empty_like(c.head_flat),
                               arena_box=torch.empty(2 + 8 * len(S.chunks), dtype=torch.float32, device=dev),
                               gpred=torch.empty((max(Rn, 1), eng.Cp), dtype=torch.float32, device=dev))
            ev_up = torch.cuda.Event()
            ev_up.record(main)
            aux.wait_event(ev_up)
            c.rpn_early["ev"] = ev_up          # (engine.backward_fused orders its auxiliary stream behind the upload through this event as well)
        with torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext():
            ops.rpn_apply_sample(labels, sumA, N, c.rpn_lists, U.d("rsel"), U.d("rnsel"), RPN_BATCH)
            for ch, dl in dls:
                n0, n1 = ch["n0"], ch["n1"]
                t0, t1 = n0 - S.d0, n1 - S.d0
                ops.rpn_apply_sample(dl, sumA, n1 - n0, c.rpn_lists[n0:n1], U.d("dsel")[t0:t1], U.d("dnsel")[t0:t1], RPN_BATCH)
                ch["_dl"] = dl
            if aux is not None:
                c.rpn_early["arena_box"].zero_()
                c.rpn_early["gpred"].zero_()
                c.rpn_early["ev_zero"] = torch.cuda.Event()
                c.rpn_early["ev_zero"].record(aux)
        oh = [[0, r] for r in Hst.rows]                       # only the row sums are used downstream
        eng._roi_gather(c, prep, U.d("osel"), U.d("onsel"), oh, c.gt, N, row_off_dev=U.d("row_off"))
        t_pred, t_ev = None, None
        if S.distill:
            # the teacher's box head on the student's sampled proposals of ALL distillation chunks (consecutive images d0 .. N of the
            # student batch = images 0 .. of the teacher's), beside the student's own
            dch = [ch for ch in S.chunks if ch["kind"] == "distill"]
            tr0, tr1 = dch[0]["r0"], dch[-1]["r1"]
            tside = S.tside
            if tside is not None:
                tside.wait_stream(main)
                with torch.cuda.stream(tside), torch.no_grad():
                    rois_t = c.rois[tr0:tr1].clone()           # (the teacher's copy of the rows, on ITS stream: not in front of the student's RoIAlign)
                    rois_t[:, 0] -= S.d0
                    t_pred = teng.box_head_on(tc, rois_t, tr1 - tr0)
                    t_ev = torch.cuda.Event()                  # (only the RoI distillation kernel waits for the teacher's predictions)
                    t_ev.record(tside)
            else:
                rois_t = c.rois[tr0:tr1].clone()
                rois_t[:, 0] -= S.d0
                with torch.no_grad():
                    t_pred = teng.box_head_on(tc, rois_t, tr1 - tr0)
        eng.roi_forward(c)
        accum = S.accum
        # ---- describe every chunk's losses and their gradient scales (host only), then ONE pass of loss kernels that yields
        # values and gradients (engine.backward_fused) and the backward; the loss-dict arithmetic comes last, off the chain
        LOSS_KEYS = ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")
        scales = []
        for ch in S.chunks:
            n0, n1 = ch["n0"], ch["n1"]
            nc = n1 - n0
            ch["values_in_backward"] = True
            ch["align"], ch["distill"] = {}, None
            if ch["do_align"]:
                eng._align_forward_chunk(c, ch)
            keys = list(LOSS_KEYS) + [k for k in ("loss_da_img", "loss_da_ins") if k[8:] in ch["align"]]
            keep = ch["keep"]
            if ch["kind"] == "distill":
                hard = {"loss_cls": dist_.do_hard_cls, "loss_rpn_cls": dist_.do_hard_obj, "loss_rpn_loc": dist_.do_hard_rpn_reg,
                        "loss_box_reg": dist_.do_hard_roi_reg}
                t0, t1, kd = n0 - S.d0, n1 - S.d0, ch["kd"]             # this chunk's images in the teacher's batch / its index among the distillation chunks
                dl = ch.pop("_dl")
                eng.distill_forward_chunk(c, ch, [h[t0:t1] for h in tc.head], t_pred[ch["r0"] - tr0: ch["r1"] - tr0], dl, Hst.nvf[kd][0], Hst.nvf[kd][1],
                                          values=False, obj_T=float(dist_.obj_temperature),
                                          cls_T=float(dist_.cls_temperature), kl=dist_.cls_loss_type == "KL", do_obj=dist_.do_obj_dst,
                                          do_rpn_reg=dist_.do_rpn_reg_dst, do_cls=dist_.do_cls_dst, do_roih_reg=dist_.do_roih_reg_dst,
                                          counts_dev=U.d("nvf")[2 * kd: 2 * kd + 2], t_ev=t_ev)
                ch["hard"] = hard
                sc = {k: (1.0 if hard.get(k, False) else 0.0) / accum for k in keys}
                k_ = ch["distill"]
                for k, on in (("loss_obj_bce", k_["do_obj"]), ("loss_rpn_l1", k_["do_rpn_reg"]), ("loss_cls_ce", k_["do_cls"]), ("loss_roih_l1", k_["do_roih_reg"])):
                    if on:
                        sc[k] = 1.0 / accum
            else:
                sc = {k: (1.0 / accum if keep(k) else 0.0) for k in keys}
            scales.append(sc)
        c.chunks = S.chunks
        c.align, c.distill = {}, None
        holder = {}
        if S.sgd:
            # The optimizer step rides inside the backward: a layer group's parameters are updated (on a third stream, behind the
            # group's producer events) as soon as its weight gradients are complete -- box head first, res3 last -- so that the
            # 0.9 GB the update streams through HBM overlaps the MFMA-bound rest of the backward instead of following it.
            applier = _GroupSGD(eng, S.hyper)
            prev_cb = getattr(eng, "grad_ready", None)
            eng.grad_ready = applier.ready
            try:
                eng.backward_fused(c, scales, after_losses=lambda: holder.update(loss_dict=self._loss_dict(S, accum, main)))
            finally:
                eng.grad_ready = prev_cb
            applier.finish()
        else:
            eng.backward_fused(c, scales, after_losses=lambda: holder.update(loss_dict=self._loss_dict(S, accum, main)))
        loss_dict = holder["loss_dict"]
        if pfx is not None:
            main.wait_stream(pfx)
        if S.tside is not None:
            main.wait_stream(S.tside)
        fields = {k: c[k] for k in ("rpn_labels", "R", "rows", "rois", "r_cls", "r_gt", "r_idx", "pred", "pooled", "fc1", "fc2", "ghead", "gpred") if k in c}
        return SimpleNamespace(loss_dict=loss_dict, fields=fields, chunks=[dict(ch) for ch in S.chunks])

    def _loss_dict(self, S, accum, main):
        """the logged loss dict from the values the loss pass just wrote (`v * 0.0`, `/ accum` for all entries in a handful of
        launches) -- on the teacher's stream, idle by now, so that these launches are not on the chain into the backward"""
        eng, dev = self.eng, self.eng.device
        side = S.tside
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            return self._loss_dict_body(S, accum, dev, eng)

    def _loss_dict_body(self, S, accum, dev, eng):
        entries = []
        for ch in S.chunks:
            losses = eng.chunk_loss_dict(ch)
            keep = ch["keep"]
            if ch["kind"] == "distill":
                hard = ch["hard"]
                out = {}
                for k, v in losses.items():
                    out[k] = v if hard.get(k, False) else (v, 0.0)      # the reference's `v * 0.0` (aldi/distill.py:181-186)
                if S.has_disc:
                    out["_da"] = torch.zeros((), device=dev)
                for k, v in eng.chunk_distill_loss_dict(ch).items():
                    out[k] = v
            else:
                out = losses
            for k, v in out.items():
                if keep(k):
                    entries.append((f"{k}_{ch['name']}", v))
        # (a row that spans several micro-batches contributes one entry per chunk under the SAME key: they add up, as the
        # sequential driver's `metrics[key] = metrics.get(key, 0) + v` does)
        kept = [i for i, (_, v) in enumerate(entries) if not isinstance(v, tuple)]
        masked = [i for i, (_, v) in enumerate(entries) if isinstance(v, tuple)]
        vals = [None] * len(entries)
        if kept:
            kv = (torch.stack([entries[i][1] for i in kept]) / accum).detach()
            for j, i in enumerate(kept):
                vals[i] = kv[j]
        if masked:
            mv = ((torch.stack([entries[i][1][0] for i in masked]) * 0.0) / accum).detach()
            for j, i in enumerate(masked):
                vals[i] = mv[j]
        loss_dict = {}
        for (n_, _), v in zip(entries, vals):                  # original key order
            loss_dict[n_] = loss_dict[n_] + v if n_ in loss_dict else v
        return loss_dict

    # ------------------------------------------------------------------------------------------------ driver
    def _static_for(self, key):
        S = self.static.get(key)
        if S is None:
            if len(self.static) >= 8:                          # multi-scale input: keep the most recent shapes only (two parities each)
                self.static.pop(next(iter(self.static)))
            S = SimpleNamespace(bufs={}, gt_pack=None, up=None, h_counts=None, graph_a=None, A=None, graphs_b={})
            self.static[key] = S
        else:
            self.static[key] = self.static.pop(key)
        return S

    def run(self, labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, ema=None, zero_grad=False, reducer=None, sgd=None, next_data=None):
        """next_data = the 4-tuple of the NEXT iteration (or None): its student images are staged now and their frozen prefix (stem + res2) runs
        inside this step's phase A, under the latency-bound proposal chain; the next call starts its student pass at res3 (`_prefix_*`).
        sgd = (lr, momentum, weight_decay): the optimizer step the caller would run right after this (EngineSGD.step), applied here
        instead, layer group by layer group as the backward completes their gradients (sets eng.wts._sgd_applied)"""
        from .model import DevicePseudoLabels
        from .trainer import _schedule_flags, _teacher_stream, plan_micro_steps
        tr = self.tr
        model, dist_ = tr.model, tr.distiller
        self.eng = eng = model.engine
        do_align, do_distill = _schedule_flags(tr)
        plan = plan_micro_steps(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, do_align=do_align, do_distill=do_distill)
        bs = tr.model_batch_size
        accum = sum(len(s_ or []) for s_ in (labeled_weak, labeled_strong, unlabeled_weak)) // bs
        if do_distill and dist_.cls_loss_type not in ("CE", "KL"):
            raise ValueError("cls_loss_type must be one of {CE, KL}")
        self.teacher = teacher = (dist_.teacher.module if hasattr(dist_.teacher, "module") else dist_.teacher) if do_distill else None
        self.teng = teacher.engine if teacher is not None else None
        dev = eng.device
        # ---- phase 0: describe the chunks, stage the inputs into their fixed buffers
        images, lab_rows, chunks, n0 = [], [], [], 0
        da = model.cfg.DOMAIN_ADAPT.ALIGN
        for row in plan:
            kind = "distill" if row.teacher_data is not None else ("labeled" if row.kwargs.get("labeled", True) else "unlabeled")
            if row.teacher_data is not None:
                assert len(row.teacher_data) == len(row.data), "Teacher and student data must be the same length."
            # one chunk per IMS_PER_GPU-sized micro-batch of the row, in the sequential driver's order (aldi/trainer.py:51-52,86-89):
            # the loss normalisers, the sampling draws and the `ManualSeed` resets are per micro-step
            for lo in range(0, len(row.data), bs):
                part = row.data[lo:lo + bs]
                n1 = n0 + len(part)
                chunks.append(dict(name=row.name, kind=kind, n0=n0, n1=n1, labeled=row.kwargs.get("labeled", True),
                                   do_align=row.kwargs.get("do_align", False) and kind != "distill", da_weights=(da.IMG_DA_WEIGHT, da.INS_DA_WEIGHT),
                                   keep=row.keep))
                images += [d["image"] for d in part]
                if kind == "labeled":
                    lab_rows += [(n0 + j, as_record(d["instances"])) for j, d in enumerate(part)]
                n0 = n1
        kd = 0
        for ch in chunks:
            if ch["kind"] == "distill":
                ch["kd"] = kd
                kd += 1
        N = n0
        # the EMA tick handed over by ALDITrainer.before_step: copy while iter <= start_iter, else lerp (aldi/ema.py:52-57)
        ema_mode = None
        if ema is not None:
            if do_distill and ema[0].model is teacher:
                ema_mode = "copy" if ema[1] <= ema[0].start_iter else "lerp"
            else:
                ema[0].update_weights(model, ema[1])               # not the teacher of this step's distiller: nothing to overlap with
        key = (tuple((ch["name"], ch["n1"] - ch["n0"]) for ch in chunks), tuple(tuple(im.shape[1:]) for im in images),
               tuple(tuple(d["image"].shape[1:]) for d in (unlabeled_weak or [])) if do_distill else (), ema_mode, bool(zero_grad), sgd is not None)
        # cross-step pipelining of the frozen prefix: the steps alternate between two sets of static buffers / graphs (parity), each reading the
        # student's res2 output from ITS prefix slot and writing the next step's into the other one
        pipe = self.pipeline and eng.prefix_pipelinable()
        par = self._parity = (self._parity ^ 1) if pipe else 0
        if pipe:
            key = key + (("pipe", par),)
        S = self._static_for(key)
        S.pipe = pipe
        S.sgd = sgd is not None
        if S.sgd:
            if getattr(S, "hyper_host", None) is None:
                S.hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory()
                S.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
            S.hyper_vals = (float(sgd[0]), float(sgd[1]), float(sgd[2]))
        S.N, S.chunks, S.accum, S.distill, S.has_disc = N, chunks, accum, do_distill, do_align
        S.nk = kd                                                  # distillation micro-steps; their images are the last ones: d0 .. N
        S.d0 = min([ch["n0"] for ch in chunks if ch["kind"] == "distill"], default=N)
        S.ema_mode, S.ema_alpha = ema_mode, (ema[0].alpha if ema is not None else None)
        S.zero_grad = bool(zero_grad)                              # clear the gradient buffer inside phase A, beside the forward
        if zero_grad:
            eng.wts._gscale = 1.0                                  # (the host half of Weights.zero_grad: a replayed graph does not run it)
        S.lazy_wt = hasattr(eng.wts, "_refresh_wt")               # (the flat-container models re-derive theirs inside adamw_step)
        if S.lazy_wt:
            eng.wts.lazy_wt = True
        S.tside = _teacher_stream(dev) if do_distill else None
        if pipe:
            self._prefix_begin(S, par, images, next_data, do_align, do_distill)
        else:
            S.stu = self._stage_images(S, "student", images)
        S.tea = self._stage_images(S, "teacher", [d["image"] for d in unlabeled_weak]) if do_distill else None
        S.gt_all = self._stage_gt(S, lab_rows, N)
        S.pl_out = None
        if do_distill:
            S.pl_out = tuple(S.gt_all[k][S.d0:N] for k in ("boxes", "classes", "count"))
        if S.up is None:
            nd = max(N - S.d0, 1)
            RPN_BATCH, ROI_BATCH = eng.p.rpn_batch, eng.p.roi_batch
            S.up = _Packed([("rsel", (N, 2, RPN_BATCH), torch.int32), ("rnsel", (N, 2), torch.int32), ("osel", (N, 2, ROI_BATCH), torch.int32),
                            ("onsel", (N, 2), torch.int32), ("row_off", (N,), torch.int32), ("dsel", (nd, 2, RPN_BATCH), torch.int32),
                            ("dnsel", (nd, 2), torch.int32), ("nvf", (2 * max(S.nk, 1),), torch.int32)], dev)
            S.h_counts = _pinned((4 * N + 2) * 4)
            S.h_counts_np = S.h_counts.numpy().view("int32")[: 4 * N + 2]
        S.h_counts_np.fill(-1)                                    # (phase A's last copy overwrites every word with a length >= 0)
        # (the ViTDet / ConvNeXt trunks draw their stochastic-depth masks on the host every step: their launches are not replayable as recorded)
        # -- they stage those masks in a persistent device buffer now (vitdet._staged_drop_scales), refreshed here before every pass:
        graph_flat = getattr(eng, "graph_safe", False) and os.environ.get("ALDI_STEP_GRAPH_FLAT", "1") == "1"
        if hasattr(eng, "refresh_drop_scales"):
            eng.refresh_drop_scales(N)
            # (their trunks mask the padding with the image sizes on the device: the persistent buffers of the staged batches, keyed by the batch)
            eng._hw_dev = {S.stu.img.data_ptr(): S.stu.hw}
            if S.tea is not None and self.teng is not None and hasattr(self.teng, "refresh_drop_scales"):
                self.teng._hw_dev = {S.tea.img.data_ptr(): S.tea.hw}
        use_graph = self.graph_enabled and self.steps_done >= self.warmup and (type(eng) is RCNN or graph_flat)
        # captured graphs read the dgrad-weight buffers of the plan they were recorded with: a layer first requested later rebuilds
        # that plan (new buffers), so everything recorded before is dropped
        epoch = (getattr(eng.wts, "wt_epoch", 0), ops.WGRAD_WS_EPOCH)     # (dgrad-weight plan, weight-gradient workspace: both are baked into the graphs)
        if getattr(S, "wt_epoch", epoch) != epoch:
            S.graph_a, S.A = None, None
            S.graphs_b.clear()
        S.wt_epoch = epoch
        # ---- phase A
        # device-side phase times of the PREVIOUS step (its events have completed by now: no extra synchronisation)
        evs = getattr(self, "_phase_events", None)
        if evs is not None and evs[3].query():
            for k_, (x, y) in (("gpu_ms_phase_a", (0, 1)), ("gpu_ms_host_gap", (1, 2)), ("gpu_ms_phase_b", (2, 3))):
                self.stats[k_] = round(evs[x].elapsed_time(evs[y]), 3)
        evs = self._phase_events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        evs[0].record()
        t0 = time.perf_counter()
        if use_graph:
            if S.graph_a is None:
                S.graph_a, S.A = self._capture(lambda: self._phase_a(S))
            S.graph_a.replay()
            self.stats["replays_a"] += 1
            A = S.A
            if S.lazy_wt:
                eng.wts._wt_dirty = False                          # (the replayed graph re-derived the dgrad weights on the device)
        else:
            A = self._phase_a(S)
        c, tc = A.c, A.tc
        if getattr(S, "pipe", False) and S.nxt_images is not None:
            self._pre_ready = (S.nxt, S.nxt_images)                # this step (phase A or B, recorded or eager) computes the next iteration's res2 output into that slot
        evs[1].record()
        self._prefetch_draws(S, int(c.anchors.shape[0]))
        # Everything the host can do WITHOUT the list lengths happens here, while the device still runs phase A: between the lengths' arrival
        # and phase B's launch the device is idle (0.15 ms of an 8.7 ms step before this was hoisted), so that section holds only the draws
        # themselves and the launch.
        if S.sgd:                                                   # this iteration's learning rate etc. for the recorded optimizer launches
            S.hyper_host[0], S.hyper_host[1], S.hyper_host[2], S.hyper_host[3] = S.hyper_vals + (float(getattr(eng.wts, "_gscale", 1.0)),)
            S.hyper.copy_(S.hyper_host, non_blocking=True)         # (stream-ordered: behind phase A and the previous step's optimizer launches)
        dp = getattr(eng, "grad_ready", None) is not None
        dp_graph = dp and reducer is not None and self.dp_graph_ok and reducer.capturable()
        graph_b = use_graph and (not dp or dp_graph) and os.environ.get("ALDI_STEP_GRAPH_B", "1") == "1"
        from .engine import raise_on_error
        from . import _lib as L_
        try:
            prep = self._draws_prepare(S)
            t1 = time.perf_counter()
            self._wait_counts(S)                                   # the ONE device->host sync: list lengths for the host RNG
            t2 = time.perf_counter()
            raise_on_error(int(S.h_counts_np[4 * N]), "student")
            raise_on_error(int(S.h_counts_np[4 * N + 1]), "teacher")
            # ---- host: all sampling draws
            Hst = self._host_draws(S, A, prep)
        except BaseException:
            L_.call("aldi_torch_rng_prefetch", None, None, 0, 0)   # joins the background stream fillers of _prefetch_draws
            raise
        t3 = time.perf_counter()
        # ---- phase B
        # Under data parallelism the backward launches the gradient exchange (reduce.BucketedReducer: collectives on a launch stream
        # behind the producers' events).  With RCCL those launches are stream-ordered kernels, so the whole of phase B -- backward,
        # collectives, their join -- is recorded and replayed like the single-GPU one; a host-driven backend (gloo) keeps phase B eager.
        evs[2].record()
        ent = S.graphs_b.get(Hst.key) if graph_b else None
        rng_after = getattr(Hst, "rng_state", None)               # (native host phase: the CPU generator's state behind the draws)
        if ent is None and rng_after is not None:
            torch.set_rng_state(rng_after)                         # phase B runs (or is recorded) from Python: the generator first
            rng_after = None
        if graph_b:
            if ent is None:
                if len(S.graphs_b) >= 4:
                    S.graphs_b.pop(next(iter(S.graphs_b)))

                def phase_b_recorded():
                    out = self._phase_b(S, A, Hst)
                    if dp_graph:
                        reducer.finish()                       # inside the recording: launch stream and collectives join the main stream
                    return out
                try:
                    ent = self._capture(phase_b_recorded)
                    S.graphs_b[Hst.key] = ent
                except Exception:
                    if not dp_graph:
                        raise
                    # the collectives of this stack cannot be recorded: say so once, keep phase B eager under DP from here on
                    import logging
                    logging.getLogger(__name__).warning("phase B with its collectives could not be captured; data-parallel phase B stays eager", exc_info=True)
                    self.dp_graph_ok = False
                    torch.cuda.synchronize()
                    reducer.__init__(reducer.grad, reducer.group, reducer.payload, reducer.exchange)
                    ent = None
        if ent is not None:
            ent[0].replay()
            if rng_after is not None:
                torch.set_rng_state(rng_after)                     # (behind the launch: the device is already working)
            if dp_graph:
                reducer.finished = True                        # the replayed graph contains the whole exchange
                self.stats["replays_b_dp"] = self.stats.get("replays_b_dp", 0) + 1
            self.stats["replays_b"] += 1
            B = ent[1]
            c.update(B.fields)
            for ch, saved in zip(S.chunks, B.chunks):
                keep_host = {k: ch[k] for k in ("rpn_counts", "roi_counts")}
                ch.update(saved)
                ch.update(keep_host)
            c.chunks = S.chunks
        else:
            B = self._phase_b(S, A, Hst)
            if not use_graph:
                self.stats["eager"] += 1
        if S.sgd:
            eng.wts._sgd_applied = True            # recorded or eager, phase B contained this iteration's optimizer step: EngineSGD.step skips its launch
        evs[3].record()
        t4 = time.perf_counter()
        self.stats["rng_stream_hits"] = int(L_.lib.aldi_torch_rng_prefetch_hits())
        for k_, v_ in (("host_us_issue_a", t1 - t0), ("host_us_wait_a", t2 - t1), ("host_us_draws", t3 - t2), ("host_us_issue_b", t4 - t3)):
            self.stats[k_] = round(0.8 * self.stats.get(k_, (v_ * 1e6)) + 0.2 * v_ * 1e6, 1)       # running mean, microseconds
        self.steps_done += 1
        # what was staged for THIS step must not be picked up by a later eager pass (a replayed step never runs trunk(), which consumes the mark;
        # the sizes buffers are keyed by an address another batch may be staged at later): ADVICE r05
        for e_ in (eng, self.teng):
            if e_ is not None:
                e_.__dict__.pop("_ds_fresh", None)
                e_.__dict__.pop("_hw_dev", None)
        if do_distill:
            teacher._last_inference = tc
            labels_ = [DevicePseudoLabels(tc.sizes[i], tc.pseudo, i) for i in range(len(unlabeled_weak))]
            for dw, ds, lab in zip(unlabeled_weak, unlabeled_strong, labels_):
                dw["instances"] = lab
                ds["instances"] = lab
        model._last_fused = c
        if use_graph and B.loss_dict:
            # the values are views into the graphs' pool, overwritten by the next replay: hand out a private copy (one launch), as
            # the eager path does by construction (loggers / DEBUG dumps read them after the next run_step)
            vec = torch.stack(list(B.loss_dict.values()))
            return {k: vec[i] for i, k in enumerate(B.loss_dict)}
        return dict(B.loss_dict)

    def _capture(self, fn):
        """record `fn`'s launches (all streams it forks to and joins from) into one hipGraph; its tensors stay allocated in the
        graphs' shared pool, so the Python objects `fn` returns remain valid views of what every replay recomputes"""
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        prio = int(os.environ.get("ALDI_MAIN_PRIO", "0"))
        kw = {}
        if prio:                      # the serial chain of the step on a high-priority stream: its small launches are not queued
            if not hasattr(self, "_cap_stream"):      # behind the (long) workgroups of the weight-gradient stream
                self._cap_stream = torch.cuda.Stream(device=self.eng.device, priority=prio)
            kw["stream"] = self._cap_stream
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local", **kw):
            out = fn()
        self.stats["captures"] += 1
        return g, out
