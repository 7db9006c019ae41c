"""Deformable-DETR's set criterion on the HIP library (LOSS / MATCHER sections of configs/Base-DETR.yaml:27-39): the matching costs and the
focal / L1 / GIoU losses with their gradients are device kernels (aldi_detr_match_cost, aldi_detr_set_loss), for all decoder layers in one
launch each (AUX_LOSS); the Hungarian assignment itself runs on the host with scipy, as in the authors' matcher (a sequential algorithm on
300 x ~10 matrices: one device -> host copy of the cost tensor and one upload of the assignment per step).  The arithmetic follows
oracle/deformable_detr.py (hungarian_match, set_losses, criterion)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .. import _lib as L
from ..ops import _p, stream_ptr


def _world_mean(n: float, device) -> float:
    """the authors' normaliser under data parallelism: the ranks' target counts summed, / world size, at least 1 (the pinned restatement,
    transformers' DeformableDetrLoss.forward, and the authors' SetCriterion.forward do the same reduction)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([n], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        n = float(t) / dist.get_world_size()
    return max(n, 1.0)


class SetCriterion:
    def __init__(self, *, cls_coef=2.0, bbox_coef=5.0, giou_coef=2.0, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25):
        self.coef = (float(cls_coef), float(bbox_coef), float(giou_coef))
        self.cost = (float(cost_class), float(cost_bbox), float(cost_giou))
        self.alpha = float(focal_alpha)

    @staticmethod
    def pad_targets(targets: List[Dict[str, torch.Tensor]], device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
        """[{"labels": (n,), "boxes": (n, 4) cxcywh in [0, 1]}] -> labels [B, Gmax] int32, boxes [B, Gmax, 4], counts [B] on the device"""
        B = len(targets)
        Gmax = max(1, max(len(t["labels"]) for t in targets))
        lab = torch.zeros((B, Gmax), dtype=torch.int32)
        box = torch.zeros((B, Gmax, 4), dtype=torch.float32)
        cnt = torch.zeros(B, dtype=torch.int32)
        for b, t in enumerate(targets):
            n = len(t["labels"])
            cnt[b] = n
            if n:
                lab[b, :n] = t["labels"].to(torch.int32).cpu()
                box[b, :n] = t["boxes"].to(torch.float32).cpu()
        return lab.to(device), box.to(device), cnt.to(device), Gmax

    def __call__(self, logits: torch.Tensor, boxes: torch.Tensor, targets: List[Dict[str, torch.Tensor]], num_boxes: float = None, match: torch.Tensor = None):
        """logits [Ld, B, Nq, K], boxes [Ld, B, Nq, 4] (device, fp32) -> (OrderedDict of WEIGHTED losses: loss_ce / loss_bbox / loss_giou and
        their `_i` copies for the earlier decoder layers, g_logits, g_boxes = gradients of the sum of the dict)"""
        from scipy.optimize import linear_sum_assignment
        Ld, B, Nq, K = logits.shape
        dev = logits.device
        logits, boxes = logits.contiguous(), boxes.contiguous()
        lab, tbox, cnt, Gmax = self.pad_targets(targets, dev)
        counts = [len(t["labels"]) for t in targets]
        if num_boxes is None:
            num_boxes = _world_mean(float(sum(counts)), dev)
        LB = Ld * B
        if match is None:
            cost = torch.empty((LB, Nq, Gmax), dtype=torch.float32, device=dev)
            L.call("aldi_detr_match_cost", _p(logits), _p(boxes), _p(lab), _p(tbox), _p(cnt), _p(cost), LB, B, Nq, K, Gmax, self.cost[0], self.cost[1], self.cost[2],
                   self.alpha, stream_ptr())
            ch = cost.cpu().numpy()                                     # the step's one device -> host hand-over
            match = torch.full((LB, Nq), -1, dtype=torch.int32)
            for lb in range(LB):
                n = counts[lb % B]
                if n:
                    qi, gi = linear_sum_assignment(ch[lb, :, :n])
                    match[lb, torch.as_tensor(qi, dtype=torch.long)] = torch.as_tensor(gi, dtype=torch.int32)
            match = match.to(dev)
        rows = torch.empty((LB * Nq, 3), dtype=torch.float32, device=dev)
        losses = torch.empty((Ld, 3), dtype=torch.float32, device=dev)
        g_logits, g_boxes = torch.empty_like(logits), torch.empty_like(boxes)
        L.call("aldi_detr_set_loss", _p(logits), _p(boxes), _p(match), _p(lab), _p(tbox), _p(rows), _p(losses), _p(g_logits), _p(g_boxes), LB, B, Nq, K, Gmax,
               self.alpha, self.coef[0], self.coef[1], self.coef[2], float(num_boxes), stream_ptr())
        out = OrderedDict()
        w = torch.tensor(self.coef, dtype=torch.float32, device=dev)
        weighted = losses * w                                           # (Ld x 3 scalars: glue)
        for l in range(Ld):
            suffix = "" if l == Ld - 1 else f"_{l}"
            for j, k in enumerate(("loss_ce", "loss_bbox", "loss_giou")):
                out[k + suffix] = weighted[l, j]
        self.last_match = match
        self.last_num_boxes = float(num_boxes)                           # the normaliser this call USED (the world mean under data parallelism)
        return out, g_logits, g_boxes
