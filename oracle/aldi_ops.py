"""CPU restatement (torch fp32) of the ALDI-owned arithmetic on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED against golden
vectors generated from the reference itself (tests/golden/make_golden.py,
tests/test_oracle_golden.py).  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import copy
import random
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from . import d2_rcnn as d2


# ----------------------------------------------------------------------------
# distillation losses
# ----------------------------------------------------------------------------

def rpn_distill_losses(student_logits: List[torch.Tensor], student_deltas: List[torch.Tensor],
                       teacher_logits: List[torch.Tensor], teacher_deltas: List[torch.Tensor],
                       pseudo_gt_labels: torch.Tensor, obj_temperature: float = 1.0,
                       do_obj: bool = True, do_reg: bool = True) -> Dict[str, torch.Tensor]:
    """reference aldi/distill.py:193-229 (get_rpn_losses).

    ``*_logits``/``*_deltas`` are the RAW rpn_head outputs per level,
    (N, A, H, W) / (N, 4A, H, W); ``pseudo_gt_labels`` is the stacked (N, sumA)
    label tensor from ``label_and_sample_anchors`` on the teacher's anchors and
    pseudo-GT.  The masks are applied across mismatched index spaces exactly as
    the reference does."""
    losses = {}
    valid_mask = torch.flatten(pseudo_gt_labels >= 0)                           # :203
    fg_mask = pseudo_gt_labels == 1                                             # :204
    t_prob = torch.sigmoid(torch.cat([torch.flatten(t) for t in teacher_logits]) / obj_temperature)   # :207
    if do_obj:
        losses["loss_obj_bce"] = F.binary_cross_entropy_with_logits(            # :210-216
            torch.cat([torch.flatten(t) for t in student_logits])[valid_mask],
            t_prob[valid_mask], reduction="mean")
    if do_reg:
        fg4 = torch.repeat_interleave(fg_mask, repeats=4)                       # :220
        s = torch.cat([torch.flatten(t) for t in student_deltas])[fg4]
        t = torch.cat([torch.flatten(t) for t in teacher_deltas])[fg4]
        losses["loss_rpn_l1"] = smooth_l1_beta0(s, t, "mean")                   # :221-227
    return losses


def smooth_l1_beta0(x, y, reduction):
    """fvcore.nn.smooth_l1_loss with beta < 1e-5: |x-y|; 'mean' of empty -> 0*sum."""
    loss = torch.abs(x - y)
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


def roih_distill_losses(student_logits, student_deltas, teacher_logits, teacher_deltas,
                        cls_temperature: float = 1.0, cls_loss_type: str = "CE",
                        do_cls: bool = True, do_reg: bool = True) -> Dict[str, torch.Tensor]:
    """reference aldi/distill.py:231-278 (get_roih_losses)."""
    losses = {}
    t_prob = F.softmax(teacher_logits / cls_temperature, dim=1)                 # :237
    if do_cls:
        if cls_loss_type == "CE":
            losses["loss_cls_ce"] = F.cross_entropy(student_logits, t_prob)     # :241-242 (soft targets, mean)
        elif cls_loss_type == "KL":
            losses["loss_cls_ce"] = F.kl_div(F.log_softmax(student_logits, dim=1),      # :243-247
                                             F.log_softmax(teacher_logits / cls_temperature, dim=1),
                                             reduction="batchmean", log_target=True)
        else:
            raise ValueError("cls_loss_type must be one of {CE, KL}")           # :248-249
    if do_reg:
        bg_idx = teacher_logits.shape[1] - 1                                    # :255
        fg_cls = torch.argmax(teacher_logits, dim=1)
        fg_mask = fg_cls != bg_idx
        ft = teacher_deltas.view(-1, bg_idx, 4)[fg_mask, fg_cls[fg_mask], :]
        fs = student_deltas.view(-1, bg_idx, 4)[fg_mask, fg_cls[fg_mask], :]
        losses["loss_roih_l1"] = smooth_l1_beta0(fs, ft, "sum") / teacher_logits.shape[0]   # :266-276
    return losses


def mask_hard_losses(hard_losses: Dict[str, torch.Tensor], do_hard_cls, do_hard_obj, do_hard_rpn_reg, do_hard_roi_reg):
    """reference aldi/distill.py:175-186."""
    attr = {"loss_cls": do_hard_cls, "loss_rpn_cls": do_hard_obj,
            "loss_rpn_loc": do_hard_rpn_reg, "loss_box_reg": do_hard_roi_reg}
    out = {}
    for k, v in hard_losses.items():
        out[k] = v if attr.get(k, False) else v * 0.0
    return out


# ----------------------------------------------------------------------------
# alignment
# ----------------------------------------------------------------------------

def conv_discriminator(x, *params):
    """reference aldi/align.py:103-119: [Conv2d(k=3, pad 0) -> ReLU] per hidden dim -> AdaptiveAvgPool2d(1) -> Flatten -> Linear.
    params = (w, b) pairs in module order, the last pair is the Linear."""
    h = x
    for i in range(0, len(params) - 2, 2):
        h = F.relu(F.conv2d(h, params[i], params[i + 1]))
    h = h.mean(dim=(2, 3))
    return F.linear(h, params[-2], params[-1])


def fc_discriminator(x, *params):
    """reference aldi/align.py:121-135: Flatten -> [Linear -> ReLU] per hidden dim -> Linear."""
    h = x.flatten(1)
    for i in range(0, len(params) - 2, 2):
        h = F.relu(F.linear(h, params[i], params[i + 1]))
    return F.linear(h, params[-2], params[-1])


def disc_params(P: dict, prefix: str):
    """(w, b) pairs of one discriminator from a state_dict, in nn.Sequential index order"""
    idx = sorted({int(k.split(".")[2]) for k in P if k.startswith(prefix + ".model.")})
    out = []
    for i in idx:
        out += [P[f"{prefix}.model.{i}.weight"], P[f"{prefix}.model.{i}.bias"]]
    return out


class _GradReverse(torch.autograd.Function):
    """reference aldi/helpers.py:51-63 (weight -1.0)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -1.0 * g.clone()


def grad_reverse(x):
    return _GradReverse.apply(x)


def domain_loss(preds, labeled: bool, weight: float):
    """reference aldi/align.py:76-90: BCE-with-logits against a constant domain label, times weight."""
    tgt = torch.full_like(preds, 1.0 if labeled else 0.0)
    return weight * F.binary_cross_entropy_with_logits(preds, tgt)


# ----------------------------------------------------------------------------
# pseudo labels, EMA
# ----------------------------------------------------------------------------

def process_bbox(pred: dict, thres: float) -> dict:
    """reference aldi/pseudolabeler.py:51-67: keep scores > thres (strict)."""
    valid = pred["scores"] > thres
    return {"image_size": pred["image_size"], "gt_boxes": pred["pred_boxes"][valid, :],
            "gt_classes": pred["pred_classes"][valid], "scores": pred["scores"][valid]}


def ema_update(teacher_sd: "OrderedDict[str, torch.Tensor]", student_sd, alpha: float, it: int, start_iter: int = 0,
               exclude_keys=("query_embed",)):
    """reference aldi/ema.py:29-57. Returns the new teacher dict."""
    if it <= start_iter:
        return OrderedDict((k, student_sd[k].clone()) for k in teacher_sd)      # :29-30,54-55
    new = OrderedDict()
    for key, value in teacher_sd.items():
        if key in student_sd:
            if any(k in key for k in exclude_keys):
                new[key] = student_sd[key] * 1                                   # :39-41
            else:
                new[key] = student_sd[key] * (1 - alpha) + value * alpha         # :43-46
        else:
            raise Exception("{} is not found in student model".format(key))     # :47-48
    return new


# ----------------------------------------------------------------------------
# step driver
# ----------------------------------------------------------------------------

def run_model_labeled_unlabeled(model: Callable, distiller, backward: Callable, labeled_weak, labeled_strong,
                                unlabeled_weak, unlabeled_strong, *, do_align: bool, backward_at_end: bool,
                                model_batch_size: int):
    """reference aldi/trainer.py:28-117, with the model / distiller / backward as callables."""
    do_weak = labeled_weak is not None
    do_strong = labeled_strong is not None
    do_distill = distiller.distill_enabled()
    total = sum(len(s or []) for s in (labeled_weak, labeled_strong, unlabeled_weak))       # :51
    accum = total // model_batch_size                                                       # :52
    loss_dict = {}

    def add(losses, suffix, cond):
        for k, v in losses.items():
            if cond(k):
                v = v / accum
                if not backward_at_end:
                    v = v.detach()
                loss_dict[f"{k}_{suffix}"] = loss_dict.get(f"{k}_{suffix}", 0) + v

    def maybe_backward(losses, cond):
        if not backward_at_end:
            ls = {k: v * 0 if not cond(k) else v for k, v in losses.items()}
            backward(sum(ls.values()) / accum)

    def train_step(data, name, cond, **kw):
        for i in range(0, len(data), model_batch_size):
            loss = model(data[i:i + model_batch_size], **kw)
            maybe_backward(loss, cond)
            add(loss, name, cond)

    if do_weak:
        train_step(labeled_weak, "source_weak", lambda k: True, do_align=do_align)
    if do_strong:
        train_step(labeled_strong, "source_strong", lambda k: True, do_align=do_align)
    if do_align:
        train_step(unlabeled_weak, "target_weak", lambda k: "_da_" in k, labeled=False, do_align=True)
    if do_distill:
        assert len(unlabeled_weak) == len(unlabeled_strong), "Teacher and student data must be the same length."
        for i in range(0, len(unlabeled_weak), model_batch_size):
            dl = distiller(unlabeled_weak[i:i + model_batch_size], unlabeled_strong[i:i + model_batch_size])
            maybe_backward(dl, lambda k: k != "_")
            add(dl, "distill", lambda k: k != "_")
    return loss_dict


class OracleALDI:
    """The reference schedule of one ALDI iteration on the CPU restatement.

    Mirrors ALDITrainer.before_step + run_step with ALDISimpleTrainer (fp32),
    reference aldi/trainer.py:122-149,242-246 and aldi/dropin.py:94-121.  It
    deliberately keeps the reference's redundancies (teacher trunk evaluated
    twice per distill micro-step, full-state-dict EMA each iteration)."""

    def __init__(self, cfg, student_sd, *, ema_alpha=0.9996, ema_start_iter=0, threshold=0.8,
                 distill=dict(do_hard_cls=False, do_hard_obj=False, do_hard_rpn_reg=False, do_hard_roi_reg=False,
                              do_cls_dst=True, do_obj_dst=True, do_rpn_reg_dst=True, do_roih_reg_dst=True,
                              cls_temperature=1.0, obj_temperature=1.0, cls_loss_type="CE"),
                 align=None, lr=0.06, momentum=0.9, weight_decay=1e-4, ims_per_gpu=2, backward_at_end=False,
                 teacher_sd=None, py_seed=0):
        self.cfg = cfg
        self.sd = OrderedDict((k, v.clone()) for k, v in student_sd.items())
        self.train_keys = d2.trainable_keys(cfg, self.sd)
        for k in self.train_keys:
            self.sd[k].requires_grad_(True)
        self.teacher = OrderedDict((k, v.detach().clone()) for k, v in (teacher_sd or student_sd).items())
        self.align = align                   # dict(img=bool, ins=bool, img_w, ins_w, params={...}) or None
        self.distill = distill
        self.threshold = threshold
        self.ema_alpha, self.ema_start = ema_alpha, ema_start_iter
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.ims_per_gpu, self.backward_at_end = ims_per_gpu, backward_at_end
        self.bufs = {}
        self.iter = 0
        self.rand = random.Random(py_seed)
        self.seed = self.rand.randint(0, 2 ** 32 - 1)        # ManualSeed.__init__ (aldi/helpers.py:19-23)
        self.last = {}
        self.pseudo_override = None
        self.proposal_override = None

    # -- model(...) = ALDI.forward -> AlignMixin.forward -> GeneralizedRCNN.forward
    def model(self, data, labeled=True, do_align=False):
        cap = d2.Captured()
        rp = self.proposal_override.pop(0) if self.proposal_override else None     # parity tests: identical inputs to the ROI stage
        losses = d2.forward_train(self.cfg, self.sd, data, roi_seed=self.seed, cap=cap, replace_proposals=rp)
        a = self.align
        if a is not None:
            if do_align:                                                         # aldi/align.py:75-90
                P = a["params"]
                if a.get("img"):
                    f = grad_reverse(cap["features"][a.get("img_layer", "p2")])
                    pr = conv_discriminator(f, *disc_params(P, "img_align"))
                    losses["loss_da_img"] = domain_loss(pr, labeled, a["img_w"])
                if a.get("ins"):
                    f = grad_reverse(cap["box_head_out"])
                    pr = fc_discriminator(f, *disc_params(P, "ins_align"))
                    losses["loss_da_ins"] = domain_loss(pr, labeled, a["ins_w"])
            elif a.get("img") or a.get("ins"):                                   # aldi/align.py:91-100
                fake = 0
                for pref, on in (("img_align", a.get("img")), ("ins_align", a.get("ins"))):
                    if on:
                        fake = fake + sum(p.sum() for k, p in a["params"].items() if k.startswith(pref)) * 0
                losses["_da"] = fake
        self.last["student_cap"] = cap
        return losses

    def distill_enabled(self):
        d = self.distill
        return any(d[k] for k in ("do_hard_cls", "do_hard_obj", "do_hard_rpn_reg", "do_hard_roi_reg",
                                  "do_cls_dst", "do_obj_dst", "do_rpn_reg_dst", "do_roih_reg_dst"))

    def distiller(self, teacher_inputs, student_inputs):
        """reference aldi/distill.py:144-191."""
        d = self.distill
        # pseudo_label_inplace (aldi/pseudolabeler.py:15-30): teacher eval inference; its roi_heads pre-hook re-seeds
        preds = d2.inference(self.cfg, self.teacher, teacher_inputs, roi_seed=self.seed)
        pls = [process_bbox(p, self.threshold) for p in preds]
        self.last["pseudo_own"] = pls
        if self.pseudo_override is not None:
            # parity tests feed the device path's pseudo-labels ("identical inputs"): discrete sampling
            # downstream is discontinuous in these coordinates, the oracle's own are compared separately
            pls = self.pseudo_override.pop(0)
        for ti, si, pl in zip(teacher_inputs, student_inputs, pls):
            ti["instances"] = pl
            si["instances"] = pl
        self.seed = self.rand.randint(0, 2 ** 32 - 1)                           # seeder.reset_seed() :150
        hard = self.model(student_inputs)                                        # :157
        scap = self.last["student_cap"]
        tcap = d2.Captured()
        with torch.no_grad():                                                    # :160-162 teacher in train mode
            d2.forward_train(self.cfg, self.teacher, teacher_inputs, replace_proposals=scap["proposals_used"],
                             roi_seed=self.seed, cap=tcap)
        losses = mask_hard_losses(hard, d["do_hard_cls"], d["do_hard_obj"], d["do_hard_rpn_reg"], d["do_hard_roi_reg"])
        labels = torch.stack(d2.label_and_sample_anchors(self.cfg, tcap["anchors"],
                                                         [i["instances"] for i in teacher_inputs])[0])   # :200-202
        losses.update(rpn_distill_losses(scap["rpn_logits"], scap["rpn_deltas"], tcap["rpn_logits"], tcap["rpn_deltas"],
                                         labels, d["obj_temperature"], d["do_obj_dst"], d["do_rpn_reg_dst"]))
        losses.update(roih_distill_losses(scap["box_scores"], scap["box_deltas"], tcap["box_scores"], tcap["box_deltas"],
                                          d["cls_temperature"], d["cls_loss_type"], d["do_cls_dst"], d["do_roih_reg_dst"]))
        self.last.update(teacher_cap=tcap, pseudo=pls, rpn_distill_labels=labels)
        return losses

    def __call__(self, teacher_inputs, student_inputs):
        return self.distiller(teacher_inputs, student_inputs)

    def _params(self):
        ps = {k: self.sd[k] for k in self.train_keys}
        if self.align is not None:
            ps.update(self.align["params"])
        return ps

    def step(self, labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong):
        """before_step (EMA) + run_step. Returns the metrics loss dict."""
        # EMA tick (aldi/trainer.py:242-246)
        ssd = OrderedDict((k, v.detach()) for k, v in self.sd.items())
        self.teacher = ema_update(self.teacher, ssd, self.ema_alpha, self.iter, self.ema_start)
        ps = self._params()
        for p in ps.values():
            p.grad = None                                                         # zero_grad
        has_align = self.align is not None and (self.align.get("img") or self.align.get("ins"))

        def backward(loss):
            loss.backward()
        ld = run_model_labeled_unlabeled(self.model, self, backward, labeled_weak, labeled_strong, unlabeled_weak,
                                         unlabeled_strong, do_align=bool(has_align),
                                         backward_at_end=self.backward_at_end, model_batch_size=self.ims_per_gpu)
        if self.backward_at_end:
            sum(ld.values()).backward()
        grads = {k: p.grad for k, p in ps.items() if p.grad is not None}
        self.last["grads"] = {k: g.clone() for k, g in grads.items()}
        d2.sgd_step(ps, grads, self.bufs, self.lr, self.momentum, self.wd)
        self.iter += 1
        return {k: float(v) for k, v in ld.items()}
