"""Distillers with the reference's registry, names, flags and loss keys (aldi/distill.py:17-285).

``ALDIDistiller.__call__(teacher_batched_inputs, student_batched_inputs)`` returns the same loss
dict as the reference -- hard losses kept or multiplied by 0.0 (:175-186) plus ``loss_obj_bce``,
``loss_rpn_l1``, ``loss_cls_ce``, ``loss_roih_l1`` -- but the schedule is MI355X-first:

* the teacher trunk runs ONCE per micro-step: the reference runs it twice on the same weak images
  (eval-mode inference aldi/pseudolabeler.py:21, then a train-mode forward aldi/distill.py:162);
  FrozenBN makes both passes identical up to the RPN head, so the features and head outputs of
  the inference pass are reused and only the box head is re-run on the student's sampled proposals
  (what ReplaceProposalsOnce + the shared ManualSeed achieve in the reference, :131-138,160);
* pseudo-labels never leave the GPU;
* the host RNG is advanced exactly as the reference does: teacher inference
  re-seeds with the old seed, reset_seed(), student RPN draws, re-seed, ROI draws, [teacher: RPN draws,
  re-seed, identical ROI draws], then the RPN draws of get_rpn_losses.
"""
from __future__ import annotations

import torch

from .helpers import ManualSeed, ReplaceProposalsOnce, SaveIO, set_attributes
from .model import GeneralizedRCNN, wire_losses
from .pseudolabeler import PseudoLabeler
from .registry import Registry

DISTILLER_REGISTRY = Registry("DISTILLER")
DISTILLER_REGISTRY.__doc__ = """
Registry for Distillers, which calculate distillation losses between a student and a teacher.
The registered object will be constructed with `obj(teacher, student, cfg)`.
A Distiller implements __call__(teacher_batched_inputs, student_batched_inputs) -> {"loss_name": loss_value, ...}
and distill_enabled() -> bool.
"""
DISTILL_MIXIN_REGISTRY = Registry("DISTILL_MIXIN")


# constructor keyword <- key under DOMAIN_ADAPT.DISTILL (aldi/config.py:60-75)
_HARD_FLAGS = (("do_hard_cls", "HARD_ROIH_CLS_ENABLED"), ("do_hard_obj", "HARD_OBJ_ENABLED"), ("do_hard_rpn_reg", "HARD_RPN_REG_ENABLED"),
               ("do_hard_roi_reg", "HARD_ROIH_REG_ENABLED"))
_SOFT_FLAGS = (("do_cls_dst", "ROIH_CLS_ENABLED"), ("do_obj_dst", "OBJ_ENABLED"), ("do_rpn_reg_dst", "RPN_REG_ENABLED"),
               ("do_roih_reg_dst", "ROIH_REG_ENABLED"))


def _flags_from(cfg, table):
    D = cfg.DOMAIN_ADAPT.DISTILL
    return {arg: D[key] for arg, key in table}


def build_distiller(cfg, teacher, student):
    name = cfg.DOMAIN_ADAPT.DISTILL.DISTILLER_NAME
    return DISTILLER_REGISTRY.get(name).from_config(cfg, teacher, student)


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


@DISTILLER_REGISTRY.register()
class Distiller:
    """This Distiller does nothing."""
    def __init__(self, teacher, student):
        pass

    @classmethod
    def from_config(cls, cfg, teacher, student):
        return Distiller(teacher, student)

    def __call__(self, teacher_batched_inputs, student_batched_inputs):
        return {}

    def distill_enabled(self):
        return False


@DISTILLER_REGISTRY.register()
class HardDistiller(Distiller):
    """Hard pseudo-label self-distillation only (aldi/distill.py:62-84)."""
    def __init__(self, teacher, student, do_hard_cls=False, do_hard_obj=False, do_hard_rpn_reg=False,
                 do_hard_roi_reg=False, pseudo_label_threshold=0.8):
        set_attributes(self, locals())
        self.pseudo_labeler = PseudoLabeler(teacher, pseudo_label_threshold)

    @classmethod
    def from_config(cls, cfg, teacher, student):
        return cls(teacher, student, pseudo_label_threshold=cfg.DOMAIN_ADAPT.TEACHER.THRESHOLD, **_flags_from(cfg, _HARD_FLAGS))

    def __call__(self, teacher_batched_inputs, student_batched_inputs):
        self.pseudo_labeler(teacher_batched_inputs, student_batched_inputs)
        return self.student(student_batched_inputs)

    def distill_enabled(self):
        return any(getattr(self, arg) for arg, _ in _HARD_FLAGS)


@DISTILLER_REGISTRY.register()
class ALDIDistiller(Distiller):
    """Hard or soft distillation (per config) for Faster R-CNN students/teachers (aldi/distill.py:87-278)."""

    def __init__(self, teacher, student, do_hard_cls=False, do_hard_obj=False, do_hard_rpn_reg=False, do_hard_roi_reg=False,
                 do_cls_dst=False, do_obj_dst=False, do_rpn_reg_dst=False, do_roih_reg_dst=False,
                 cls_temperature=1.0, obj_temperature=1.0, cls_loss_type="CE", pseudo_label_threshold=0.8):
        set_attributes(self, locals())
        self.register_hooks()
        self.pseudo_labeler = PseudoLabeler(teacher, pseudo_label_threshold)

    @classmethod
    def from_config(cls, cfg, teacher, student):
        D = cfg.DOMAIN_ADAPT.DISTILL
        return cls(teacher, student, cls_temperature=D.CLS_TMP, obj_temperature=D.OBJ_TMP, cls_loss_type=cfg.DOMAIN_ADAPT.CLS_LOSS_TYPE,
                   pseudo_label_threshold=cfg.DOMAIN_ADAPT.TEACHER.THRESHOLD, **_flags_from(cfg, _HARD_FLAGS + _SOFT_FLAGS))

    # where the reference taps the two models (aldi/distill.py:122-138): attribute that receives the SaveIO <- module path.  The attribute
    # names are API (third-party distillers read `student_rpn_io` ...); the engine fires these hook points with the tensors it computed.
    _TAPS = {"student": (("student_rpn_io", "proposal_generator"), ("student_rpn_head_io", "proposal_generator.rpn_head"),
                         ("student_boxpred_io", "roi_heads.box_predictor")),
             "teacher": (("teacher_backbone_io", "backbone"), ("teacher_rpn_head_io", "proposal_generator.rpn_head"),
                         ("teacher_boxpred_io", "roi_heads.box_predictor"), ("teacher_anchor_io", "proposal_generator.anchor_generator"))}

    def register_hooks(self):
        models = {"student": _unwrap(self.student), "teacher": _unwrap(self.teacher)}
        for side, taps in self._TAPS.items():
            for attr, path in taps:
                point = models[side]
                for part in path.split("."):
                    point = getattr(point, part)
                io = SaveIO()
                setattr(self, attr, io)
                point.register_forward_hook(io)
        # one seeder on BOTH roi_heads (teacher first: its eval inference fires it with the previous seed), so that the two
        # models draw the same proposal samples; the teacher's train-mode forward then takes the student's proposals, once
        self.seeder = ManualSeed()
        self.teacher_proposal_replacer = ReplaceProposalsOnce()
        for pre_hook, side in ((self.seeder, "teacher"), (self.seeder, "student"), (self.teacher_proposal_replacer, "teacher")):
            models[side].roi_heads.register_forward_pre_hook(pre_hook)

    def distill_enabled(self):
        return any(getattr(self, arg) for arg, _ in _HARD_FLAGS + _SOFT_FLAGS)

    def _distill_forward(self, teacher_batched_inputs, student_batched_inputs):
        if self.cls_loss_type not in ("CE", "KL"):
            raise ValueError("cls_loss_type must be one of {CE, KL}")
        student, teacher = _unwrap(self.student), _unwrap(self.teacher)
        # 1) hard pseudo labels, in place (teacher eval inference; its roi_heads pre-hook re-seeds with the OLD seed)
        tc = self.pseudo_labeler(teacher_batched_inputs, student_batched_inputs)
        self.seeder.reset_seed()
        was_eval = not teacher.training
        if was_eval:
            teacher.train()
        # 2) student on the strong views, pseudo-GT
        standard_losses = self.student(student_batched_inputs)
        holder = student._last
        c = holder.ctx
        # 3) teacher "train-mode forward" on the weak views with the student's proposals:
        #    trunk + RPN head are the ones computed in (1); RNG advanced as the reference would
        torch.manual_seed(self.seeder.seed)
        student.engine._sample(c.roi_host_counts, student.engine.p.roi_batch, student.engine.p.roi_pos_frac)        # the teacher's identical ROI draws
        t_pred = teacher.engine.box_head_on(tc, c.rois, c.R)
        teacher.proposal_generator.rpn_head.fire(None, tc.head)
        teacher.roi_heads.box_predictor.fire(None, t_pred)
        if was_eval:
            teacher.eval()
        # 4) labels for the RPN distillation: a FRESH sample on the teacher's anchors and pseudo-GT (aldi/distill.py:200-202)
        labels, n_valid, n_fg, _ = student.engine.rpn_sample(c.rpn_lists, c.rpn_counts, c.N, host_counts=c.rpn_host_counts)
        student.engine.distill_forward(c, tc.head, t_pred, labels, n_valid, n_fg,
                                       obj_T=float(self.obj_temperature), cls_T=float(self.cls_temperature), kl=self.cls_loss_type == "KL",
                                       do_obj=self.do_obj_dst, do_rpn_reg=self.do_rpn_reg_dst,
                                       do_cls=self.do_cls_dst, do_roih_reg=self.do_roih_reg_dst)
        self._soft = wire_losses(holder, student.engine.distill_loss_dict(c))
        return standard_losses

    # hard (pseudo-label) loss of the student's own forward -> the flag that keeps it; everything else is reported as 0 * value so that the
    # loss dict keeps its keys and the graph its shape (aldi/distill.py:175-186)
    _HARD_LOSS_FLAG = {"loss_cls": "do_hard_cls", "loss_rpn_cls": "do_hard_obj", "loss_rpn_loc": "do_hard_rpn_reg", "loss_box_reg": "do_hard_roi_reg"}

    def __call__(self, teacher_batched_inputs, student_batched_inputs):
        hard = self._distill_forward(teacher_batched_inputs, student_batched_inputs)
        kept = {k for k, flag in self._HARD_LOSS_FLAG.items() if getattr(self, flag)}
        losses = {k: (v if k in kept else v * 0.0) for k, v in hard.items()}
        losses.update(self.get_rpn_losses(teacher_batched_inputs))
        losses.update(self.get_roih_losses())
        return losses

    def get_rpn_losses(self, teacher_batched_inputs=None):
        return {k: v for k, v in self._soft.items() if k in ("loss_obj_bce", "loss_rpn_l1")}

    def get_roih_losses(self):
        return {k: v for k, v in self._soft.items() if k in ("loss_cls_ce", "loss_roih_l1")}


@DISTILL_MIXIN_REGISTRY.register()
class DistillMixin(GeneralizedRCNN):
    pass
