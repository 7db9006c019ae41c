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
    list_instances = []
    num_proposal_output = 0.0
    for proposal_bbox_inst in proposals:
        proposal_bbox_inst = process_bbox(proposal_bbox_inst, thres=cur_threshold)
        num_proposal_output += len(proposal_bbox_inst)
        list_instances.append(proposal_bbox_inst)
    num_proposal_output = num_proposal_output / len(proposals)
    return list_instances, num_proposal_output


def process_bbox(proposal_bbox_inst, thres=0.7):
    """Host-side statement of the filter (strict >), for Instances that are already materialised."""
    valid_map = proposal_bbox_inst.scores > thres
    new_proposal_inst = Instances(proposal_bbox_inst.image_size)
    new_proposal_inst.gt_boxes = Boxes(proposal_bbox_inst.pred_boxes.tensor[valid_map, :]).to("cpu")
    new_proposal_inst.gt_classes = proposal_bbox_inst.pred_classes[valid_map].to("cpu")
    new_proposal_inst.scores = proposal_bbox_inst.scores[valid_map].to("cpu")
    return new_proposal_inst


def add_label(unlabled_data, label):
    for unlabel_datum, lab_inst in zip(unlabled_data, label):
        unlabel_datum["instances"] = lab_inst
    return unlabled_data
