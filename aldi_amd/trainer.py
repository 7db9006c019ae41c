"""ALDITrainer and the step drivers behind the reference's names (aldi/trainer.py:28-246,
aldi/dropin.py:29-184).  The iteration's schedule -- source weak/strong, target-weak alignment,
distillation, loss-key suffixes, 1/accum scaling -- is a table (``plan_micro_steps``) executed either
chunk by chunk (``run_model_labeled_unlabeled``, the reference's order) or as one fused student pass
(``fused_run_model``) on the HIP engine."""
from __future__ import annotations

import contextlib
import copy
import logging
import os
import time
import weakref
from typing import Callable, Dict, List, NamedTuple, Optional

from collections import OrderedDict

import torch
import torch.distributed as dist

from . import synthetic
from .dataloader import SyntheticDetectionLoader, WeakStrongDataloader
from .distill import build_distiller
from .ema import EMA
from .model import build_aldi

DEBUG = False
debug_dict = {}


class MicroStep(NamedTuple):
    """One row of the iteration's schedule: which batch goes through the student (or through the distiller together with
    the teacher's batch), with which forward flags, under which loss-key suffix, and which of its losses count."""
    name: str                                   # loss-key suffix
    data: Optional[list]                        # student inputs
    teacher_data: Optional[list]                # set => the distiller runs this row (teacher inputs = weak views)
    kwargs: dict                                # forward flags of model(...)
    keep: Callable[[str], bool]                 # losses that contribute (the others are multiplied by 0 / not reported)


def plan_micro_steps(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, *, do_align: bool, do_distill: bool) -> List[MicroStep]:
    """The schedule of one Mean-Teacher iteration as data (contract: reference aldi/trainer.py:28-117):
    labeled weak and strong views are plain supervised rows that keep every loss; with alignment on the unlabeled weak
    views go through the student only for their "_da_" losses; with distillation on the distiller sees (weak, strong)
    pairs and keeps everything it names.  Both drivers below -- the sequential one and the fused one -- execute this plan."""
    everything = lambda k: True
    plan: List[MicroStep] = []
    if labeled_weak is not None:
        plan.append(MicroStep("source_weak", labeled_weak, None, {"do_align": do_align}, everything))
    if labeled_strong is not None:
        plan.append(MicroStep("source_strong", labeled_strong, None, {"do_align": do_align}, everything))
    if do_align:
        plan.append(MicroStep("target_weak", unlabeled_weak, None, {"labeled": False, "do_align": True}, lambda k: "_da_" in k))
    if do_distill:
        plan.append(MicroStep("distill", unlabeled_strong, unlabeled_weak, {}, lambda k: k != "_"))
    return plan


def _schedule_flags(trainer):
    core = getattr(trainer.model, "module", trainer.model)
    do_align = any(getattr(core, a, None) is not None for a in ("img_align", "ins_align"))
    return do_align, trainer.distiller.distill_enabled()


def run_model_labeled_unlabeled(trainer, labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong):
    """Sequential driver of the plan (the reference's schedule, aldi/trainer.py:28-117): IMS_PER_GPU-sized chunks one after
    the other, each followed by its own backward unless BACKWARD_AT_END; every reported loss is divided by the number of
    gradient-accumulation steps = (|labeled_weak| + |labeled_strong| + |unlabeled_weak|) // IMS_PER_GPU (unlabeled_strong is
    not counted).  Returns {f"{loss}_{suffix}": 0-d tensor} (detached when the backward already ran)."""
    do_align, do_distill = _schedule_flags(trainer)
    plan = plan_micro_steps(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, do_align=do_align, do_distill=do_distill)
    bs = trainer.model_batch_size
    accum = sum(len(part or []) for part in (labeled_weak, labeled_strong, unlabeled_weak)) // bs
    early = not trainer.backward_at_end
    if DEBUG:
        debug_dict.update(last_labeled_weak=copy.deepcopy(labeled_weak), last_labeled_strong=copy.deepcopy(labeled_strong),
                          last_unlabeled_weak=copy.deepcopy(unlabeled_weak), last_unlabeled_strong=copy.deepcopy(unlabeled_strong))
    metrics: Dict[str, torch.Tensor] = {}
    for row in plan:
        if row.teacher_data is not None:
            assert len(row.teacher_data) == len(row.data), "Teacher and student data must be the same length."
        for lo in range(0, len(row.data), bs):
            chunk = row.data[lo:lo + bs]
            if row.teacher_data is not None:
                losses = trainer.distiller(row.teacher_data[lo:lo + bs], chunk)
            else:
                losses = trainer.model(chunk, **row.kwargs)
            if early:               # non-contributing losses still enter the graph with weight 0 (every parameter gets a gradient)
                trainer.do_backward(sum(v if row.keep(k) else v * 0 for k, v in losses.items()) / accum, override=True)
            for k, v in losses.items():
                if row.keep(k):
                    key = f"{k}_{row.name}"
                    v = v / accum
                    metrics[key] = metrics.get(key, 0) + (v.detach() if early else v)
    if DEBUG and do_distill:
        debug_dict["last_pseudolabeled"] = copy.deepcopy(unlabeled_strong)
    return metrics


def fused_run_model_detr(trainer, labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong):
    """the plan of `run_model_labeled_unlabeled` for the Deformable-DETR detector with its student passes FUSED (reference schedule:
    aldi/trainer.py:28-117 with HardDistiller, aldi/distill.py:62-84): the teacher pseudo-labels every target chunk first (the EMA tick of
    this iteration has run; the teacher does not depend on the student's passes), then all student chunks -- source and pseudo-labelled
    target -- go through ONE trunk + transformer pass and ONE backward (`DeformableDETR.forward_fused`: the set criterion per chunk, so
    matching, normalisers, loss keys, the 1 / accum scaling and the early-backward semantics are the sequential driver's)."""
    model = trainer.model
    do_align, do_distill = _schedule_flags(trainer)
    plan = plan_micro_steps(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, do_align=do_align, do_distill=do_distill)
    bs = trainer.model_batch_size
    accum = sum(len(part or []) for part in (labeled_weak, labeled_strong, unlabeled_weak)) // bs
    parts, rows, pairs = [], [], []
    for row in plan:
        for lo in range(0, len(row.data), bs):
            chunk = row.data[lo:lo + bs]
            if row.teacher_data is not None:
                pairs.append((row.teacher_data[lo:lo + bs], chunk))
            parts.append(chunk)
            rows.append(row)
    # the teacher's inference (what HardDistiller.__call__ runs before the student's pass) on the second stream, beside the student's label-free
    # forward: its pseudo labels are first read by the set criterion
    side = _teacher_stream(model.device) if pairs else None
    main = torch.cuda.current_stream()
    if side is not None:
        side.wait_stream(main)
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        for weak, strong in pairs:
            trainer.distiller.pseudo_labeler(weak, strong)
    ctx = model.forward_fused_begin(parts)
    if side is not None:
        main.wait_stream(side)
    ctx, losses = model.forward_fused_finish(ctx, parts)
    model.backward_fused(ctx, [1.0 / accum] * len(parts))
    trainer._detr_fused_steps = getattr(trainer, "_detr_fused_steps", 0) + 1
    metrics: Dict[str, torch.Tensor] = {}
    for row, ld in zip(rows, losses):
        for k, v in ld.items():
            if row.keep(k):
                key = f"{k}_{row.name}"
                metrics[key] = metrics.get(key, 0) + (v / accum).detach()
    if DEBUG and do_distill:
        debug_dict["last_pseudolabeled"] = copy.deepcopy(unlabeled_strong)
    return metrics


def _data_parallel() -> bool:
    """more than one rank -- or ALDI_DP_FORCE=1 with an initialised process group of any size: the data-parallel code path (bucketed
    exchange, its stream plumbing, the captured collectives) on a single GPU, which is how the RCCL path is exercised on a 1-GPU box"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("ALDI_DP_FORCE", "0") == "1"


_TEACHER_STREAMS = {}


def _teacher_stream(device):
    """second HIP stream for the teacher's inference (ALDI_TEACHER_STREAM=0 keeps everything on one stream)"""
    if os.environ.get("ALDI_TEACHER_STREAM", "1") != "1" or torch.device(device).type != "cuda":
        return None
    key = str(device)
    if key not in _TEACHER_STREAMS:
        # (ALDI_TEACHER_PRIO: HIP stream priority of the teacher's branch, e.g. -1 = above the student's; measured)
        _TEACHER_STREAMS[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("ALDI_TEACHER_PRIO", "0")))
    return _TEACHER_STREAMS[key]


def fused_run_model(trainer, labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong):
    """MI355X-first form of `run_model_labeled_unlabeled` for the reference's FPN configurations
    (BATCH_CONTENTS = labeled_strong [+ unlabeled_strong], one IMS_PER_GPU chunk each): the source, the
    target-weak alignment and the distillation student passes go through ONE trunk/head launch sequence
    (`RCNN.forward_train_fused`) and ONE backward, with a single device->host sync for the sampling counts.
    Same loss-dict keys, values, 1/accum scaling, `v*0` masking and global-RNG stream as the sequential
    driver (tests/test_engine_gpu.py::test_fused_step_equals_sequential)."""
    from .model import DevicePseudoLabels
    from .structures import as_record
    model, dist_ = trainer.model, trainer.distiller
    eng = model.engine
    bs = trainer.model_batch_size
    do_align, do_distill = _schedule_flags(trainer)
    plan = plan_micro_steps(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong, do_align=do_align, do_distill=do_distill)
    total = sum(len(s_ or []) for s_ in [labeled_weak, labeled_strong, unlabeled_weak])
    accum = total // bs
    has_disc = do_align
    da = model.cfg.DOMAIN_ADAPT.ALIGN
    da_w = (da.IMG_DA_WEIGHT, da.INS_DA_WEIGHT)
    fire_student = lambda: model.roi_heads.fire_pre()
    specs = []
    for row in plan:
        if row.teacher_data is None:                # supervised / alignment rows: one chunk of the fused student pass each
            specs.append(dict(images=[d["image"] for d in row.data], instances=[as_record(d["instances"]) for d in row.data],
                              labeled=row.kwargs.get("labeled", True), do_align=row.kwargs.get("do_align", False), da_weights=da_w,
                              pre_roi=fire_student))
    tc = None
    if do_distill:
        if dist_.cls_loss_type not in ("CE", "KL"):
            raise ValueError("cls_loss_type must be one of {CE, KL}")
        teacher = dist_.teacher.module if hasattr(dist_.teacher, "module") else dist_.teacher
        # The teacher's inference (N = 2, mostly small launches) runs on its own HIP stream beside the student's trunk / RPN head
        # / proposal generation, none of which needs the pseudo-labels.  It is ENQUEUED after them (the engine calls
        # `gt_lazy` once the student's label-free work is in the queue): the host needs ~3 ms to issue the teacher's launches,
        # and issued first they would simply run alone while the student's kernels are still being queued behind them.
        side = _teacher_stream(model.device)
        ev0 = None
        if side is not None:
            ev0 = torch.cuda.Event()
            ev0.record()                          # previous step's optimizer / EMA / readers of the teacher's outputs
        st_ = {}

        def teacher_gt():
            if side is not None:
                side.wait_event(ev0)
                with torch.cuda.stream(side), torch.no_grad():
                    st_["tc"] = teacher.engine.inference([d["image"] for d in unlabeled_weak], dist_.pseudo_label_threshold)
            else:
                with torch.no_grad():
                    st_["tc"] = teacher.engine.inference([d["image"] for d in unlabeled_weak], dist_.pseudo_label_threshold)
            return st_["tc"].pseudo
        gt_wait = (lambda: torch.cuda.current_stream().wait_stream(side)) if side is not None else None

        def pre_rpn_distill():
            teacher.roi_heads.fire_pre()         # the teacher's eval inference re-seeds with the OLD seed
            dist_.seeder.reset_seed()
        t_out = {}

        def teacher_box_head(c_, n0, r0, r1):
            """the teacher's box head on the student's sampled proposals (aldi/distill.py:160-162), beside the student's own"""
            rois_t = c_.rois[r0:r1].clone()
            rois_t[:, 0] -= n0
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                rois_t.record_stream(side)
                with torch.cuda.stream(side), torch.no_grad():
                    t_out["pred"] = teacher.engine.box_head_on(st_["tc"], rois_t, r1 - r0)
            else:
                t_out["pred"] = teacher.engine.box_head_on(st_["tc"], rois_t, r1 - r0)
        specs.append(dict(images=[d["image"] for d in unlabeled_strong], gt_lazy=teacher_gt, gt_wait=gt_wait, labeled=True, do_align=False,
                          pre_rpn=pre_rpn_distill, pre_roi=fire_student, post_rois=teacher_box_head))
    c = eng.forward_train_fused(specs)
    if do_distill:
        tc = st_["tc"]
        teacher._last_inference = tc
        labels_ = [DevicePseudoLabels(tc.sizes[i], tc.pseudo, i) for i in range(len(unlabeled_weak))]
        for dw, ds, lab in zip(unlabeled_weak, unlabeled_strong, labels_):
            dw["instances"] = lab
            ds["instances"] = lab
    loss_dict = {}
    scales = []
    entries = []
    for ch, row in zip(c.chunks, plan):
        name = row.name
        losses = eng.chunk_loss_dict(ch)
        sc = {}
        if row.teacher_data is not None:
            hard = {"loss_cls": dist_.do_hard_cls, "loss_rpn_cls": dist_.do_hard_obj, "loss_rpn_loc": dist_.do_hard_rpn_reg,
                    "loss_box_reg": dist_.do_hard_roi_reg}
            n0, n1, r0, r1 = ch["n0"], ch["n1"], ch["r0"], ch["r1"]
            torch.manual_seed(dist_.seeder.seed)
            eng._sample_host(ch["roi_counts"], eng.p.roi_batch, eng.p.roi_pos_frac)   # the teacher's identical ROI draws
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            t_pred = t_out["pred"]
            labels, n_valid, n_fg, _ = eng.rpn_sample(c.rpn_lists[n0:n1], None, n1 - n0, host_counts=ch["rpn_counts"])
            eng.distill_forward_chunk(c, ch, tc.head, t_pred, labels, n_valid, n_fg, obj_T=float(dist_.obj_temperature),
                                      cls_T=float(dist_.cls_temperature), kl=dist_.cls_loss_type == "KL", do_obj=dist_.do_obj_dst,
                                      do_rpn_reg=dist_.do_rpn_reg_dst, do_cls=dist_.do_cls_dst, do_roih_reg=dist_.do_roih_reg_dst)
            out = {}
            for k, v in losses.items():
                out[k] = v if hard.get(k, False) else (v, 0.0)      # the reference's `v * 0.0`, applied in the batched arithmetic below
                sc[k] = (1.0 if hard.get(k, False) else 0.0) / accum
            if has_disc:
                out["_da"] = torch.zeros((), device=model.device)
            for k, v in eng.chunk_distill_loss_dict(ch).items():
                out[k] = v
                sc[k] = 1.0 / accum
        else:
            out = losses
            sc = {k: (1.0 / accum if row.keep(k) else 0.0) for k in losses}
        scales.append(sc)
        for k, v in out.items():
            if row.keep(k):
                entries.append((f"{k}_{name}", v))
    # loss-dict arithmetic (`v * 0.0` masking, `/ accum`) for all entries in a handful of launches instead of two or three
    # 0-d kernels per entry (the host issues those at ~10 us apiece in the middle of the step)
    kept = [(n_, v) for n_, v in entries if not isinstance(v, tuple)]
    masked = [(n_, v[0]) for n_, v in entries if isinstance(v, tuple)]
    vals = {}
    if kept:
        kv = (torch.stack([v for _, v in kept]) / accum).detach()
        vals.update({n_: kv[i] for i, (n_, _) in enumerate(kept)})
    if masked:
        mv = ((torch.stack([v for _, v in masked]) * 0.0) / accum).detach()
        vals.update({n_: mv[i] for i, (n_, _) in enumerate(masked)})
    for n_, _ in entries:                                        # original key order
        loss_dict[n_] = loss_dict[n_] + vals[n_] if n_ in loss_dict else vals[n_]
    eng.backward_fused(c, scales)
    model._last_fused = c
    return loss_dict


def _num_classes(cfg) -> int:
    """the detector's class count: ROI_HEADS.NUM_CLASSES, or DEFORMABLE_DETR.NUM_CLASSES for that meta-architecture"""
    if cfg.MODEL.META_ARCHITECTURE == "DeformableDETR":
        return int(cfg.MODEL.DEFORMABLE_DETR.NUM_CLASSES)
    return int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)


class EngineSGD:
    """torch.optim.SGD-shaped handle on the fused HIP optimizer (momentum / weight decay of detectron2's build_optimizer)."""
    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4):
        self.model = model
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]

    def zero_grad(self, set_to_none: bool = False):
        self.model.weights.zero_grad()

    def step(self):
        W = self.model.weights
        if getattr(W, "_sgd_applied", False):        # the fused step ran this iteration's update inside its backward (FusedStep.run(sgd=...))
            W._sgd_applied = False
            W._after_sgd()
            return
        g = self.param_groups[0]
        W.sgd_step(g["lr"], g["momentum"], g["weight_decay"])

    def state_dict(self):
        """checkpointable (detectron2 saves the optimizer through `trainer._trainer.optimizer`): the flat momentum buffer"""
        W = self.model.weights
        return {"format": "aldi_amd.flat_sgd", "param_groups": [dict(g) for g in self.param_groups],
                "momentum_buffer": None if W._mom is None else W.mom.detach().cpu().clone(), "first_step": bool(W.first_step)}

    def load_state_dict(self, sd):
        W = self.model.weights
        if sd.get("format") != "aldi_amd.flat_sgd":
            logging.getLogger(__name__).warning("optimizer state of a foreign format: momentum restarts from zero")
            return
        for g, h in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in h.items() if k in ("momentum", "weight_decay")})
        if sd.get("momentum_buffer") is not None:
            W.mom.copy_(sd["momentum_buffer"].to(W.mom.device))
        W.first_step = bool(sd.get("first_step", sd.get("momentum_buffer") is None))


class EngineAdamW:
    """torch.optim.AdamW-shaped handle on the fused HIP AdamW of the ViTDet / ConvNeXt models (reference aldi/backbone.py:66-84:
    detectron2 common/optim.py AdamW -- betas (0.9, 0.999), weight_decay 0.1, none on the norms and on pos_embed).
    `lr_decay_rate` = ViTDet's layer-wise lr decay, which the reference turns on exactly for build_vitdet_b_backbone
    (aldi/trainer.py:204 -> get_vit_lr_decay_rate(num_layers=12, lr_decay_rate=0.7))."""
    def __init__(self, model, lr, weight_decay=0.1, betas=(0.9, 0.999), lr_decay_rate=None, num_layers=12):
        self.model = model
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": betas}]
        self.lr_decay_rate, self.num_layers = lr_decay_rate, num_layers

    def zero_grad(self, set_to_none: bool = False):
        self.model.weights.zero_grad()

    def step(self):
        g = self.param_groups[0]
        kw = {}
        if self.lr_decay_rate is not None:
            kw = dict(lr_decay_rate=self.lr_decay_rate, num_layers=self.num_layers)
        self.model.weights.adamw_step(g["lr"], betas=g["betas"], weight_decay=g["weight_decay"], **kw)

    def state_dict(self):
        W = self.model.weights
        has = getattr(W, "_m", None) is not None
        return {"format": "aldi_amd.flat_adamw", "param_groups": [dict(g) for g in self.param_groups], "step": int(getattr(W, "step_count", 0)),
                "exp_avg": W._m.detach().cpu().clone() if has else None, "exp_avg_sq": W._v.detach().cpu().clone() if has else None}

    def load_state_dict(self, sd):
        W = self.model.weights
        if sd.get("format") != "aldi_amd.flat_adamw":
            logging.getLogger(__name__).warning("optimizer state of a foreign format: AdamW moments restart from zero")
            return
        if sd.get("exp_avg") is not None:
            W._m = sd["exp_avg"].to(W.device).clone()
            W._v = sd["exp_avg_sq"].to(W.device).clone()
        W.step_count = int(sd.get("step", 0))


class EngineDetrAdamW:
    """AdamW of the Deformable-DETR detector over its flat state: the parameter groups and the full-model gradient clip of
    configs/Base-DETR.yaml:59-69 (DetrWeights.adamw_step)"""
    def __init__(self, model, cfg):
        S = cfg.SOLVER
        self.model = model
        self.param_groups = [{"lr": S.BASE_LR, "weight_decay": S.WEIGHT_DECAY, "betas": (0.9, 0.999)}]
        cg = S.CLIP_GRADIENTS
        self.clip = float(cg.CLIP_VALUE) if cg.ENABLED else 0.0
        if cg.ENABLED and (cg.CLIP_TYPE != "full_model" or float(cg.NORM_TYPE) != 2.0):
            raise ValueError("DeformableDETR: only CLIP_TYPE full_model with the L2 norm is implemented")
        self.backbone_mult, self.proj_mult, self.proj_names = float(S.BACKBONE_LR_MULTIPLIER), float(S.LR_LINEAR_PROJ_MULTIPLIER), tuple(S.LR_LINEAR_PROJ_NAMES)

    def zero_grad(self, set_to_none: bool = False):
        self.model.weights.zero_grad()

    def step(self):
        g = self.param_groups[0]
        self.model.weights.adamw_step(g["lr"], betas=g["betas"], weight_decay=g["weight_decay"], clip=self.clip, backbone_mult=self.backbone_mult,
                                      proj_mult=self.proj_mult, proj_names=self.proj_names)

    def state_dict(self):
        W = self.model.weights
        return {"format": "aldi_amd.detr_adamw", "param_groups": [dict(g) for g in self.param_groups], "step": int(W.step),
                "exp_avg": None if W.m is None else W.m.detach().cpu().clone(), "exp_avg_sq": None if W.v is None else W.v.detach().cpu().clone()}

    def load_state_dict(self, sd):
        W = self.model.weights
        if sd.get("format") != "aldi_amd.detr_adamw":
            logging.getLogger(__name__).warning("optimizer state of a foreign format: the Deformable-DETR AdamW moments restart from zero")
            return
        if sd.get("exp_avg") is not None:
            W.m, W.v = sd["exp_avg"].to(W.master.device).clone(), sd["exp_avg_sq"].to(W.master.device).clone()
        W.step = int(sd.get("step", 0))


class WarmupMultiStepLR:
    """detectron2 WarmupMultiStepLR (linear warmup), stepped once per iteration."""
    def __init__(self, optimizer, base_lr, steps, gamma, warmup_factor, warmup_iters):
        self.opt, self.base_lr, self.steps, self.gamma = optimizer, base_lr, tuple(steps), gamma
        self.warmup_factor, self.warmup_iters = warmup_factor, warmup_iters
        self.last_iter = 0
        self._apply()

    def lr_at(self, it):
        wf = 1.0
        if it < self.warmup_iters:
            alpha = it / self.warmup_iters
            wf = self.warmup_factor * (1 - alpha) + alpha
        return self.base_lr * wf * self.gamma ** sum(1 for s in self.steps if s <= it)

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.lr_at(self.last_iter)

    def step(self):
        self.last_iter += 1
        self._apply()

    def state_dict(self):
        return {"last_epoch": self.last_iter}           # torch's scheduler key name

    def load_state_dict(self, sd):
        self.last_iter = int(sd["last_epoch"])
        self._apply()


class SimpleTrainer:
    """One iteration = fetch, zero_grad, run_model, backward, (all-reduce), optimizer step (aldi/dropin.py:87-130)."""
    def __init__(self, model, data_loader, optimizer, zero_grad_before_forward=False):
        self.model, self.data_loader, self.optimizer = model, data_loader, optimizer
        self._data_loader_iter_obj = None
        self.zero_grad_before_forward = zero_grad_before_forward
        self.last_loss_dict: Dict[str, float] = {}

    @property
    def _data_loader_iter(self):
        if self._data_loader_iter_obj is None:
            self._data_loader_iter_obj = iter(self.data_loader)
        return self._data_loader_iter_obj

    def run_step(self):
        """One optimizer step = fetch a batch, run the model (which may already contain its backward), reduce gradients across ranks,
        apply the update.  Same order of effects as the reference's step (aldi/dropin.py:94-121), organised around what THIS engine
        can take over: `run_model` may clear the gradients itself, run the backward itself (`_fused_done`) and even the update."""
        if not self.model.training:
            raise AssertionError("[SimpleTrainer] model was changed to eval mode!")
        t0 = time.perf_counter()
        batch = self._fetch_batch()
        fetch_s = time.perf_counter() - t0
        clear_first = self.zero_grad_before_forward
        if clear_first and self._defers_zero_grad():
            self._zero_pending = True                # the fused step clears them on a side stream beside its forward
        elif clear_first:
            self.optimizer.zero_grad()
        if self._defers_sgd():
            self._sgd_pending = True                 # the fused step may apply this iteration's update inside its backward
        result = self.run_model(batch)
        self.__dict__.pop("_sgd_pending", None)
        fused = bool(getattr(self, "_fused_done", False))
        if isinstance(result, torch.Tensor):
            loss_dict, total = {"total_loss": result}, result
        else:
            # (a fused step has nothing left to differentiate; summing its entries would only put a dozen tiny launches before the update)
            loss_dict, total = result, (None if fused else sum(result.values()))
        if not clear_first and not fused:
            self.optimizer.zero_grad()
        self.do_backward(total)
        self.after_backward()
        self._write_metrics(loss_dict, fetch_s)
        self.optimizer.step()

    def run_model(self, data):
        return self.model(data)

    def _wants_lookahead(self) -> bool:
        """True when run_model can use batch k + 1 while it runs step k (the fused step's cross-step pipelining of the frozen prefix)"""
        return False

    def _fetch_batch(self):
        """the batch of this step; with look-ahead the NEXT one is fetched too (the reference's loader is asynchronous and always a batch ahead,
        aldi/trainer.py:211-238, aldi/dataloader.py:45-55) and left in `_next_batch` for run_model -- the order of batches is unchanged"""
        if not self._wants_lookahead():
            ahead = self.__dict__.pop("_ahead_batch", None)
            self._next_batch = None
            return ahead if ahead is not None else next(self._data_loader_iter)
        ahead = self.__dict__.pop("_ahead_batch", None)
        batch = ahead if ahead is not None else next(self._data_loader_iter)
        try:
            self._ahead_batch = next(self._data_loader_iter)
        except StopIteration:
            self._ahead_batch = None
        self._next_batch = self._ahead_batch
        return batch

    def _defers_zero_grad(self) -> bool:
        """True when run_model clears the gradients itself (the fused step does it on a side stream beside its forward)"""
        return False

    def _defers_sgd(self) -> bool:
        """True when run_model may run this iteration's optimizer step itself (and then sets weights._sgd_applied for EngineSGD.step)"""
        return False

    def do_backward(self, losses):
        losses.backward()

    def after_backward(self):
        """One all-reduce of the student gradients per step (the reference's DDP reduces on every
        micro-step backward, aldi/dropin.py:53; the sum is linear so one reduction at the end is equivalent)."""
        if _data_parallel():
            red = getattr(self, "_reducer", None)
            if red is not None:                  # fused step: most of the exchange already ran under the backward
                red.finish()
                self._reducer = None
            else:
                dist.all_reduce(self.model.weights.grad, op=dist.ReduceOp.SUM)
            self.model.weights.scale_grad(1.0 / dist.get_world_size())

    def _write_metrics(self, loss_dict, data_time):
        self.last_loss_dict = loss_dict
        self.last_data_time = data_time


class AMPTrainer(SimpleTrainer):
    """Reference: fp16 autocast + GradScaler (aldi/dropin.py:131-184).  Here AMP means the bf16 compute mode of the
    engine (fp32 master weights / accumulators); bf16 needs no loss scaling, so there is no GradScaler."""
    def run_step(self):
        assert self.model.training, "[AMPTrainer] model was changed to eval mode!"
        assert torch.cuda.is_available(), "[AMPTrainer] CUDA is required for AMP training!"
        super().run_step()


class _ALDITrainer:
    def __init__(self, model, data_loader, optimizer, distiller, backward_at_end=True, model_batch_size=None):
        super().__init__(model, data_loader, optimizer, zero_grad_before_forward=not backward_at_end)
        self.distiller = distiller
        self.backward_at_end = backward_at_end
        self.model_batch_size = model_batch_size
        self.fused = False

    def _fusable_distiller(self) -> bool:
        """the fused driver evaluates ALDIDistiller's losses with its own kernels and does not fire the models' hook points: it stands in
        only for the built-in distiller -- the class itself, or a subclass that overrides none of the loss / forward methods -- and only
        while nobody else listens on the student's or the teacher's hook points (a third-party distiller or a user's SaveIO tap would be
        silently bypassed: those take the reference's sequential schedule, logged once)"""
        from .distill import ALDIDistiller
        from .helpers import foreign_hooks
        d = self.distiller
        why = None
        if not isinstance(d, ALDIDistiller):
            why = "distiller %s is not ALDIDistiller" % type(d).__name__
        elif any(getattr(type(d), m) is not getattr(ALDIDistiller, m) for m in ("__call__", "_distill_forward", "get_rpn_losses", "get_roih_losses", "register_hooks")):
            why = "distiller %s overrides a loss / forward method of ALDIDistiller" % type(d).__name__
        else:
            own = [getattr(d, a) for side in d._TAPS.values() for a, _ in side] + [d.seeder, d.teacher_proposal_replacer]
            models = [m.module if hasattr(m, "module") else m for m in (d.student, d.teacher)]
            if any(foreign_hooks(m, own) for m in models):
                why = "a forward (pre-)hook that the distiller did not register sits on the student or the teacher"
        if why is not None and not self.__dict__.get("_fuse_warned"):
            self._fuse_warned = True
            import logging
            logging.getLogger(__name__).warning("fused step not used (%s): taking the reference's sequential schedule", why)
        return why is None

    def _can_fuse(self, data):
        """the fused driver takes the reference's FPN batch contents (labeled_strong [+ unlabeled weak / strong pairs]) in any whole
        number of IMS_PER_GPU-sized micro-batches per part -- e.g. the shipped IMS_PER_BATCH 48 / IMS_PER_GPU 2 on 8 GPUs = three
        source and three distillation micro-steps (reference configs/Base-RCNN-FPN.yaml:15-16, aldi/trainer.py:51-52) -- as long as
        the whole iteration fits one student pass (16 images: the staging kernels' limit)"""
        lw, ls, uw, us = data
        bs = self.model_batch_size
        from .engine import RCNN
        if not isinstance(getattr(self.model, "engine", None), RCNN):        # (e.g. the Deformable-DETR detector: sequential driver)
            return False
        if not self.fused or lw is not None or ls is None or len(ls) == 0 or len(ls) % bs or hasattr(self.model, "module"):
            return False
        do_align, do_distill = _schedule_flags(self)
        total = len(ls)
        if do_distill:
            if not self._fusable_distiller() or uw is None or us is None:
                return False
            if len(uw) != len(us) or len(us) == 0 or len(us) % bs:
                return False
            total += len(us)
        if do_align:
            if uw is None or len(uw) == 0 or len(uw) % bs:
                return False
            total += len(uw)
        if total > 16:
            return False
        if os.environ.get("ALDI_FUSED_LEGACY", "0") == "1" and (len(ls) != bs or (do_distill and len(us) != bs)):
            return False                                     # (the older single-chunk driver, kept for A/B runs)
        return True

    def _can_fuse_detr(self, data) -> bool:
        """the Deformable-DETR detector's fused student pass (fused_run_model_detr): plain source + hard-pseudo-label batches with an early
        backward per chunk (every loss kept with the weight 1 / accum), the built-in HardDistiller (or none), nobody listening on the hook
        points, and chunks that share ONE padded canvas -- the input projections' GroupNorm runs over the canvas, so a chunk padded to a
        larger one than its own would normalise differently than `model(chunk)` does"""
        lw, ls, uw, us = data
        bs = self.model_batch_size
        model = self.model
        if not getattr(model, "detr", False) or not self.fused or hasattr(model, "module") or self.backward_at_end:
            return False
        if lw is not None or ls is None or len(ls) == 0 or len(ls) % bs:
            return False
        do_align, do_distill = _schedule_flags(self)
        if do_align:
            return False
        chunks = [ls[i:i + bs] for i in range(0, len(ls), bs)]
        if do_distill:
            from .distill import HardDistiller
            from .helpers import foreign_hooks
            d = self.distiller
            if type(d) is not HardDistiller or uw is None or us is None or len(uw) != len(us) or len(us) == 0 or len(us) % bs:
                return False
            if any(foreign_hooks(m.module if hasattr(m, "module") else m, []) for m in (d.student, d.teacher)):
                return False
            chunks += [us[i:i + bs] for i in range(0, len(us), bs)]
        if sum(len(c_) for c_ in chunks) > 16:
            return False
        canvas = {(max(int(b["image"].shape[-2]) for b in c_), max(int(b["image"].shape[-1]) for b in c_)) for c_ in chunks}
        return len({((h + 31) // 32, (w + 31) // 32) for h, w in canvas}) == 1

    def _wants_lookahead(self) -> bool:
        from .engine import RCNN
        model = self.model.module if hasattr(self.model, "module") else self.model
        eng = getattr(model, "engine", None)
        return (bool(self.fused) and (bool(model.cfg.SOLVER.get("PIPELINE_PREFIX", False)) or os.environ.get("ALDI_PIPELINE_PREFIX") == "1") and os.environ.get("ALDI_PIPELINE_PREFIX") != "0"
                and os.environ.get("ALDI_FUSED_LEGACY", "0") != "1" and type(eng) is RCNN and eng.prefix_pipelinable())

    def _defers_zero_grad(self) -> bool:
        from .engine import RCNN
        model = self.model.module if hasattr(self.model, "module") else self.model
        # (only the flat R50 weight container: the ViTDet / ConvNeXt models keep their trunk's gradients in a second buffer)
        return bool(self.fused) and os.environ.get("ALDI_FUSED_LEGACY", "0") != "1" and type(getattr(model, "engine", None)) is RCNN

    def _defers_sgd(self) -> bool:
        W = getattr(self.model, "weights", None)
        return (self._defers_zero_grad() and isinstance(self.optimizer, EngineSGD) and not _data_parallel() and W is not None
                and not W.first_step and os.environ.get("ALDI_SGD_IN_STEP", "0") == "1")

    def run_model(self, data):
        self._fused_done = False
        pending_ema = self.__dict__.pop("_pending_ema", None)       # the EMA tick `before_step` left for the fused step to run
        if pending_ema is not None and not self._can_fuse(data):
            pending_ema[0].update_weights(self.model, pending_ema[1])
            pending_ema = None
        defer = self.__dict__.pop("_zero_pending", False)          # run_step left the zero_grad of this iteration to this method
        if defer and not self._can_fuse(data):
            self.optimizer.zero_grad()               # (this batch takes the unfused path)
            defer = False
        if self._can_fuse(data):
            self._fused_done = True
            if not self.zero_grad_before_forward:    # the fused driver runs its backward inside run_model
                if self._defers_zero_grad():
                    defer = True
                else:
                    self.optimizer.zero_grad()
            eng = self.model.engine
            reducer = None
            if _data_parallel():
                from .reduce import BucketedReducer, resolve_exchange
                reducer = self._reducer = BucketedReducer(self.model.weights.grad, payload=str(self.model.cfg.SOLVER.get("GRAD_PAYLOAD", "fp32")),
                                                          exchange=resolve_exchange(self.model.cfg.SOLVER.get("GRAD_EXCHANGE", "auto")))
                eng.grad_ready = reducer.ready
            ok = False
            try:
                if os.environ.get("ALDI_FUSED_LEGACY", "0") == "1":
                    out = fused_run_model(self, *data)
                    ok = True
                    return out
                if getattr(self, "_fused_step", None) is None:
                    from .fused_step import FusedStep
                    self._fused_step = FusedStep(self)
                sgd = None
                if self.__dict__.pop("_sgd_pending", False) and reducer is None:
                    g_ = self.optimizer.param_groups[0]
                    sgd = (g_["lr"], g_["momentum"], g_["weight_decay"])
                out = self._fused_step.run(*data, ema=pending_ema, zero_grad=defer, reducer=reducer, sgd=sgd,
                                           next_data=self.__dict__.get("_next_batch") if self._wants_lookahead() else None)
                ok = True
                return out
            finally:
                eng.grad_ready = None
                if not ok:
                    self._reducer = None         # a failed step must not leave its half-used reducer to the next `after_backward`
        if self._can_fuse_detr(data):
            self._fused_done = True
            return fused_run_model_detr(self, *data)
        return run_model_labeled_unlabeled(self, *data)

    def do_backward(self, losses, override=False):
        if getattr(self, "_fused_done", False):
            return                                   # the fused driver already ran its single backward
        if self.backward_at_end or override:
            super().do_backward(losses)


class ALDIAMPTrainer(_ALDITrainer, AMPTrainer):
    pass


class ALDISimpleTrainer(_ALDITrainer, SimpleTrainer):
    pass


class DefaultTrainer:
    """Constructor sequence of the reference's DefaultTrainer (aldi/dropin.py:35-70) without the Detectron2 hook zoo."""
    def __init__(self, cfg):
        self.cfg = cfg
        model = self.build_model(cfg)
        optimizer = self.build_optimizer(cfg, model)
        data_loader = self.build_train_loader(cfg)
        model = self.create_ddp_model(model, broadcast_buffers=False, cfg=cfg)
        self._trainer = self._create_trainer(cfg, model, data_loader, optimizer)
        self.scheduler = self.build_lr_scheduler(cfg, optimizer)
        self.checkpointer = self._create_checkpointer(model, cfg)
        self.start_iter = 0
        self.iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER

    def _create_trainer(self, cfg, model, data_loader, optimizer):
        return (AMPTrainer if cfg.SOLVER.AMP.ENABLED else SimpleTrainer)(model, data_loader, optimizer)

    def create_ddp_model(self, model, broadcast_buffers, cfg):
        """One process per GPU; gradients are all-reduced over RCCL in SimpleTrainer.after_backward.  Rank 0's weights are broadcast once."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(model.weights.master, src=0)
            model.weights.refresh()
        return model

    @classmethod
    def build_lr_scheduler(cls, cfg, optimizer):
        S = cfg.SOLVER
        return WarmupMultiStepLR(optimizer, S.BASE_LR, S.STEPS, S.GAMMA, S.WARMUP_FACTOR, S.WARMUP_ITERS)

    @classmethod
    def build_optimizer(cls, cfg, model):
        return EngineSGD(model, cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY)

    def _create_checkpointer(self, model, cfg, ckpt_cls=None):
        """aldi/dropin.py:77-82: the trainer itself (iteration, LR-scheduler hook, optimizer) is a checkpointable"""
        from .checkpoint import DetectionCheckpointer
        return (ckpt_cls or DetectionCheckpointer)(model, cfg.OUTPUT_DIR, trainer=weakref.proxy(self))

    def state_dict(self):
        """detectron2 TrainerBase / SimpleTrainer.state_dict layout"""
        ret = {"iteration": self.iter, "hooks": {"LRScheduler": self.scheduler.state_dict()}}
        opt = self._trainer.optimizer
        if hasattr(opt, "state_dict"):
            ret["_trainer"] = {"optimizer": opt.state_dict()}
        return ret

    def load_state_dict(self, sd):
        self.iter = int(sd["iteration"])
        sched = sd.get("hooks", {}).get("LRScheduler")
        if sched is not None:
            self.scheduler.load_state_dict(sched)
        else:                                            # file without scheduler state: the schedule is a function of the iteration
            self.scheduler.load_state_dict({"last_epoch": self.iter + 1})
        opt = sd.get("_trainer", {}).get("optimizer")
        if opt is not None and hasattr(self._trainer.optimizer, "load_state_dict"):
            self._trainer.optimizer.load_state_dict(opt)

    def resume_or_load(self, resume=True):
        """detectron2 DefaultTrainer.resume_or_load: last checkpoint of OUTPUT_DIR when resuming (model, EMA, optimizer momentum,
        LR schedule position and iteration continue), else cfg.MODEL.WEIGHTS (an empty string = keep the initial weights)."""
        ret = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            if "trainer" not in ret:                     # e.g. a file written by an evaluation-only tool
                self.iter = int(ret.get("iteration", -1))
                self.scheduler.load_state_dict({"last_epoch": self.iter + 1})
            self.start_iter = self.iter + 1
            self.iter = self.start_iter
        return ret

    @property
    def model(self):
        return self._trainer.model

    def before_step(self):
        pass

    def run_step(self):
        self._trainer.run_step()

    def _is_main_process(self):
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def after_step(self):
        """LRScheduler hook, then detectron2's PeriodicCheckpointer (fvcore): `model_{iter:07d}` every SOLVER.CHECKPOINT_PERIOD
        iterations and `model_final` after the last one, main process only."""
        self.scheduler.step()
        period = int(self.cfg.SOLVER.get("CHECKPOINT_PERIOD", 0) or 0)
        if self._is_main_process() and self.cfg.OUTPUT_DIR:
            if period > 0 and (self.iter + 1) % period == 0 and self.iter < self.max_iter - 1:
                self.checkpointer.save("model_{:07d}".format(self.iter), iteration=self.iter)
            if self.iter >= self.max_iter - 1:
                self.checkpointer.save("model_final", iteration=self.iter)

    def train(self):
        for self.iter in range(self.start_iter, self.max_iter):
            self.before_step()
            self.run_step()
            self.after_step()
        self.iter += 1


class ALDITrainer(DefaultTrainer):
    """Mean-Teacher trainer (aldi/trainer.py:140-246): builds student, EMA teacher and distiller; EMA tick before every step."""

    def _create_trainer(self, cfg, model, data_loader, optimizer):
        self.ema = EMA(build_aldi(cfg), cfg.EMA.ALPHA, cfg.EMA.START_ITER) if cfg.EMA.ENABLED else None
        distiller = build_distiller(cfg=cfg, teacher=self.ema.model if cfg.EMA.ENABLED else model, student=model)
        trainer = (ALDIAMPTrainer if cfg.SOLVER.AMP.ENABLED else ALDISimpleTrainer)(model, data_loader, optimizer, distiller,
                                                                                    backward_at_end=cfg.SOLVER.BACKWARD_AT_END,
                                                                                    model_batch_size=cfg.SOLVER.IMS_PER_GPU)
        trainer.fused = bool(cfg.SOLVER.get("FUSED_STEP", True)) and os.environ.get("ALDI_FUSED_STEP", "1") == "1"
        return trainer

    def _create_checkpointer(self, model, cfg):
        """aldi/trainer.py:151-156: EMA-aware start + the teacher as a checkpointable"""
        from .checkpoint import DetectionCheckpointer, DetectionCheckpointerWithEMA
        checkpointer = super()._create_checkpointer(model, cfg, ckpt_cls=DetectionCheckpointerWithEMA if cfg.EMA.LOAD_FROM_EMA_ON_START else DetectionCheckpointer)
        if cfg.EMA.ENABLED:
            checkpointer.add_checkpointable("ema", self.ema)
        return checkpointer

    @classmethod
    def build_model(cls, cfg):
        model = build_aldi(cfg)
        logging.getLogger(__name__).info("Model: %s", type(model).__mro__)
        return model

    @classmethod
    def build_optimizer(cls, cfg, model):
        if cfg.SOLVER.OPTIMIZER is None or cfg.SOLVER.OPTIMIZER.upper() == "SGD":
            if getattr(model, "adamw", False):
                raise ValueError("the ViTDet / ConvNeXt models are trained with SOLVER.OPTIMIZER ADAMW (their Base-RCNN-*.yaml)")
            return super(ALDITrainer, cls).build_optimizer(cfg, model)
        if cfg.SOLVER.OPTIMIZER.upper() == "ADAMW" and getattr(model, "detr", False):
            return EngineDetrAdamW(model, cfg)
        if cfg.SOLVER.OPTIMIZER.upper() == "ADAMW" and getattr(model, "adamw", False):       # reference aldi/trainer.py:200-209
            vitdet_b = cfg.MODEL.BACKBONE.NAME == "build_vitdet_b_backbone"          # include_vit_lr_decay of the reference
            return EngineAdamW(model, cfg.SOLVER.BASE_LR, lr_decay_rate=0.7 if vitdet_b else None, num_layers=12)
        raise ValueError(f"Unsupported optimizer/backbone combination {cfg.SOLVER.OPTIMIZER} {cfg.MODEL.BACKBONE.NAME}.")

    @classmethod
    def build_train_loader(cls, cfg):
        """Same batch-size arithmetic as the reference (aldi/trainer.py:211-240); the loaders are synthetic
        (datasets / decode / augmentation are outside the hot path: SURVEY.md section 8a row a20)."""
        contents, ratios = tuple(cfg.DATASETS.BATCH_CONTENTS), tuple(cfg.DATASETS.BATCH_RATIOS)
        if len(contents) != len(ratios):
            raise AssertionError("len(cfg.DATASETS.BATCH_CONTENTS) must equal len(cfg.DATASETS.BATCH_RATIOS).")
        total = cfg.SOLVER.IMS_PER_BATCH
        # every batch part gets its (truncated) share of the global batch; the shares must add up again
        # (a LIST, one entry per batch part: a content named twice counts twice, as in the reference's loop, aldi/trainer.py:211-222)
        sizes = [int(total * r / sum(ratios)) for r in ratios]
        if sum(sizes) != total:
            raise AssertionError(f"sum(batch_sizes)={sum(sizes)} must equal total_batch_size={total}")
        share = {}
        for part, n in zip(contents, sizes):
            share[part] = max(share.get(part, 0), n)
        # weak and strong views of one domain are cut from the same images: a domain's loader serves the larger of its parts
        labeled_bs = max((n for part, n in share.items() if part.startswith("labeled")), default=0)
        unlabeled_bs = max((n for part, n in share.items() if part.startswith("unlabeled")), default=0)
        batch_contents = contents
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        syn = cfg.get("SYNTHETIC", {})
        h, w = syn.get("HEIGHT", 800), syn.get("WIDTH", 1333)
        K = _num_classes(cfg)
        fixed = bool(syn.get("FIXED", False))
        labeled_loader = SyntheticDetectionLoader(labeled_bs // world, h, w, K, 1000 + 17 * rank, True, fixed=fixed) if labeled_bs > 0 else None
        unlabeled_loader = SyntheticDetectionLoader(unlabeled_bs // world, h, w, K, 2000 + 17 * rank, False, fixed=fixed) if unlabeled_bs > 0 else None
        return WeakStrongDataloader(labeled_loader, unlabeled_loader, batch_contents)

    def before_step(self):
        super(ALDITrainer, self).before_step()
        if self.cfg.EMA.ENABLED:
            t = self._trainer
            if getattr(t, "fused", False) and os.environ.get("ALDI_FUSED_LEGACY", "0") != "1" and type(t.model.engine).__name__ == "RCNN":
                # the fused step runs the tick on the teacher's stream, beside the student's forward (only the teacher's
                # inference needs the new weights); `run_model` falls back to running it right away when it cannot fuse
                t._pending_ema = (self.ema, self.iter)
            else:
                self.ema.update_weights(t.model, self.iter)

    # ---- evaluation (reference aldi/trainer.py:166-196: COCO evaluator, EvalHook on the EMA model, BestCheckpointer on bbox/AP50)
    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        """synthetic validation split: a finite list of batches of {image, image_id, height, width} + detectron2-format records"""
        syn_ = cfg.get("SYNTHETIC", {})
        h, w = syn_.get("HEIGHT", 800), syn_.get("WIDTH", 1333)
        n, K = int(syn_.get("VAL_IMAGES", 8)), _num_classes(cfg)
        g = torch.Generator().manual_seed(4242)
        batches, records = [], []
        for i in range(n):
            img, inst = synthetic.make_image(h, w, int(torch.randint(5, 21, (1,), generator=g)), K, g)
            batches.append([{"image": img, "image_id": i, "height": h, "width": w}])
            records.append(dict(image_id=i, height=h, width=w,
                                annotations=[dict(bbox=b.tolist(), bbox_mode="XYXY_ABS", category_id=int(c))
                                             for b, c in zip(inst["gt_boxes"], inst["gt_classes"])]))
        return batches, records

    @classmethod
    def build_evaluator(cls, cfg, dataset_name, output_folder=None, dataset_dicts=None):
        """Just do COCO Evaluation."""
        from .evaluation import Detectron2COCOEvaluatorAdapter
        if output_folder is None:
            output_folder = os.path.join(cfg.OUTPUT_DIR, "inference")
        return Detectron2COCOEvaluatorAdapter(dataset_name, dataset_dicts or [], _num_classes(cfg), output_dir=output_folder)

    @classmethod
    def test(cls, cfg, model, evaluators=None):
        """detectron2 DefaultTrainer.test: {dataset: {"bbox": {...}}} (flattened to the single dict when there is one dataset)"""
        from .evaluation import inference_on_dataset
        names = list(cfg.DATASETS.TEST) or ["synthetic_val"]
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        results = OrderedDict()
        for i, name in enumerate(names):
            loader, records = cls.build_test_loader(cfg, name)
            ev = evaluators[i] if evaluators is not None else cls.build_evaluator(cfg, name, dataset_dicts=records)
            # detectron2's InferenceSampler: every rank runs a disjoint shard of the test set, the evaluator gathers the
            # predictions on rank 0 (each detection counted once; the ground truth stays the full set)
            results[name] = inference_on_dataset(model, loader[rank::world] if world > 1 else loader, ev)
        return results[names[0]] if len(results) == 1 else results

    def after_step(self):
        """aldi/trainer.py:173-196: EvalHook on the EMA model (the student without EMA) every TEST.EVAL_PERIOD iterations and
        after the last one; BestCheckpointer on `bbox/AP50` (one test set) or `<test_set>/bbox/AP50` (several), "max",
        written as `<test_set>_model_best.pth` by the main process."""
        super(ALDITrainer, self).after_step()
        period = self.cfg.TEST.EVAL_PERIOD
        last = self.iter >= self.max_iter - 1
        if period > 0 and ((self.iter + 1) % period == 0 or last):
            self._last_eval_results = self.test(self.cfg, self.ema.model if self.cfg.EMA.ENABLED else self.model)
            names = list(self.cfg.DATASETS.TEST) or ["synthetic_val"]
            if not self._is_main_process():
                return
            best = self.__dict__.setdefault("_best_ap50", {})
            for name in names:
                res = self._last_eval_results if len(names) == 1 else self._last_eval_results.get(name, {})
                ap50 = res.get("bbox", {}).get("AP50", float("nan"))
                if ap50 == ap50 and ap50 > best.get(name, float("-inf")):
                    best[name] = ap50
                    self.checkpointer.save(f"{name}_model_best", iteration=self.iter)


Trainer = ALDITrainer   # BASELINE.json calls it aldi.trainer.Trainer
