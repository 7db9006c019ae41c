"""Teacher pseudo-labelling with the reference's API (aldi/pseudolabeler.py:7-73).  The teacher
inference + threshold filter run on device (engine.inference); labels stay GPU-resident and are
attached to BOTH the weak and the strong dicts as the same object, as the reference does."""
from __future__ import annotations

import torch

from .model import DevicePseudoLabels
from .structures import Boxes, Instances


class PseudoLabeler:
    def __init__(self, model, threshold):
        self.model = model
        self.threshold = threshold

    def __call__(self, unlabeled_weak, unlabeled_strong):
        return pseudo_label_inplace(self.model, unlabeled_weak, unlabeled_strong, self.threshold)


def pseudo_label_inplace(model, unlabeled_weak, unlabeled_strong, threshold):
    with torch.no_grad():
        was_training = model.training
        model.eval()
        model.inference(unlabeled_weak, do_postprocess=False, pl_thresh=threshold)
        if was_training:
            model.train()
        c = model._last_inference
        teacher_preds = [DevicePseudoLabels(c.sizes[i], c.pseudo, i) for i in range(len(unlabeled_weak))]
        add_label(unlabeled_weak, teacher_preds)
        if unlabeled_strong is not None:
            add_label(unlabeled_strong, teacher_preds)
        return c


def process_pseudo_label(proposals, cur_threshold):
    """-> (filtered Instances per image, mean number kept): the host-side form of what `aldi_detections` does on the device
    (reference aldi/pseudolabeler.py:40-49, same name and return shape)"""
    kept = [process_bbox(inst, thres=cur_threshold) for inst in proposals]
    return kept, sum(len(k) for k in kept) / len(proposals)


def process_bbox(proposal_bbox_inst, thres=0.7):
    """keep detections scoring strictly above `thres`, renamed to ground-truth fields and moved to the host (aldi/pseudolabeler.py:51-67)"""
    keep = proposal_bbox_inst.scores > thres
    out = Instances(proposal_bbox_inst.image_size)
    for dst, src in (("gt_classes", proposal_bbox_inst.pred_classes), ("scores", proposal_bbox_inst.scores)):
        setattr(out, dst, src[keep].to("cpu"))
    out.gt_boxes = Boxes(proposal_bbox_inst.pred_boxes.tensor[keep]).to("cpu")
    return out


def add_label(unlabled_data, label):
    """attach pseudo labels as the `instances` of each dict (the SAME object for the weak and the strong view); returns its input"""
    for datum, inst in zip(unlabled_data, label):
        datum["instances"] = inst
    return unlabled_data
