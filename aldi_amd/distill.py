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
        D = cfg.DOMAIN_ADAPT.DISTILL
        return HardDistiller(teacher, student, do_hard_cls=D.HARD_ROIH_CLS_ENABLED, do_hard_obj=D.HARD_OBJ_ENABLED,
                             do_hard_rpn_reg=D.HARD_RPN_REG_ENABLED, do_hard_roi_reg=D.HARD_ROIH_REG_ENABLED,
                             pseudo_label_threshold=cfg.DOMAIN_ADAPT.TEACHER.THRESHOLD)

    def __call__(self, teacher_batched_inputs, student_batched_inputs):
        self.pseudo_labeler(teacher_batched_inputs, student_batched_inputs)
        return self.student(student_batched_inputs)

    def distill_enabled(self):
        return any([self.do_hard_cls, self.do_hard_obj, self.do_hard_rpn_reg, self.do_hard_roi_reg])


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
        return ALDIDistiller(teacher, student,
                             do_hard_cls=D.HARD_ROIH_CLS_ENABLED, do_hard_obj=D.HARD_OBJ_ENABLED,
                             do_hard_rpn_reg=D.HARD_RPN_REG_ENABLED, do_hard_roi_reg=D.HARD_ROIH_REG_ENABLED,
                             do_cls_dst=D.ROIH_CLS_ENABLED, do_obj_dst=D.OBJ_ENABLED,
                             do_rpn_reg_dst=D.RPN_REG_ENABLED, do_roih_reg_dst=D.ROIH_REG_ENABLED,
                             cls_temperature=D.CLS_TMP, obj_temperature=D.OBJ_TMP,
                             cls_loss_type=cfg.DOMAIN_ADAPT.CLS_LOSS_TYPE,
                             pseudo_label_threshold=cfg.DOMAIN_ADAPT.TEACHER.THRESHOLD)

    def register_hooks(self):
        self.student_rpn_io, self.student_rpn_head_io, self.student_boxpred_io = SaveIO(), SaveIO(), SaveIO()
        self.teacher_backbone_io, self.teacher_rpn_head_io, self.teacher_boxpred_io, self.teacher_anchor_io = SaveIO(), SaveIO(), SaveIO(), SaveIO()
        student_model, teacher_model = _unwrap(self.student), _unwrap(self.teacher)
        student_model.proposal_generator.register_forward_hook(self.student_rpn_io)
        student_model.proposal_generator.rpn_head.register_forward_hook(self.student_rpn_head_io)
        student_model.roi_heads.box_predictor.register_forward_hook(self.student_boxpred_io)
        teacher_model.backbone.register_forward_hook(self.teacher_backbone_io)
        teacher_model.proposal_generator.rpn_head.register_forward_hook(self.teacher_rpn_head_io)
        teacher_model.roi_heads.box_predictor.register_forward_hook(self.teacher_boxpred_io)
        teacher_model.proposal_generator.anchor_generator.register_forward_hook(self.teacher_anchor_io)
        # same seeds for proposal sampling in teacher/student
        self.seeder = ManualSeed()
        teacher_model.roi_heads.register_forward_pre_hook(self.seeder)
        student_model.roi_heads.register_forward_pre_hook(self.seeder)
        self.teacher_proposal_replacer = ReplaceProposalsOnce()
        teacher_model.roi_heads.register_forward_pre_hook(self.teacher_proposal_replacer)

    def distill_enabled(self):
        return any([self.do_hard_cls, self.do_hard_obj, self.do_hard_rpn_reg, self.do_hard_roi_reg,
                    self.do_cls_dst, self.do_obj_dst, self.do_rpn_reg_dst, self.do_roih_reg_dst])

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

    def __call__(self, teacher_batched_inputs, student_batched_inputs):
        losses = {}
        hard_losses = self._distill_forward(teacher_batched_inputs, student_batched_inputs)
        loss_to_attr = {"loss_cls": self.do_hard_cls, "loss_rpn_cls": self.do_hard_obj,
                        "loss_rpn_loc": self.do_hard_rpn_reg, "loss_box_reg": self.do_hard_roi_reg}
        for k, v in hard_losses.items():
            losses[k] = v if loss_to_attr.get(k, False) else v * 0.0
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
