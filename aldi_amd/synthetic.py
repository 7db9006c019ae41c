"""Deterministic synthetic weights and COCO-style batches (there is no network for datasets or
checkpoints).  Used by bench.py, smoke() and the tests; shared by the HIP
path and the CPU oracle so both see identical inputs.  Pure data generation (torch CPU RNG)."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from .arch import d2_convs, disc_convs


def init_state_dict(num_classes: int, seed: int = 1, head_gain: float = 12.0, img_da=False, ins_da=False,
                    input_rms: float = 75.0) -> "OrderedDict[str, torch.Tensor]":
    """Random-init weights of the R50-FPN Faster R-CNN in Detectron2 state_dict layout.

    Variance-preserving (fan-in) init keeps activations O(1) through the 16 residual blocks
    with non-trivial FrozenBN buffers; the stem's running_var absorbs the raw pixel scale; the
    last BN of each block is damped.  ``head_gain`` scales the (otherwise N(0, 0.01)) predictor
    inits so that scores / proposals / pseudo-labels are non-degenerate with random weights."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    convs = d2_convs(num_classes)
    convs.update(disc_convs(img_da, ins_da))
    for name, c in convs.items():
        kk = max(c.k, 1)
        fan_in = c.cin * kk * kk
        if c.bn:
            std = math.sqrt(2.0 / fan_in)
        else:
            std = math.sqrt(1.0 / fan_in)
        if name.endswith("objectness_logits") or name.endswith("cls_score"):
            std = 0.01 * head_gain
        if name.endswith("anchor_deltas"):
            std = 0.002 * head_gain
        if name.endswith("bbox_pred"):
            std = 0.001 * head_gain
        if name.endswith("box_head.fc1") or name.endswith("box_head.fc2") or name.endswith("rpn_head.conv"):
            std = math.sqrt(2.0 / fan_in)
        shape = (c.cout, c.cin, c.k, c.k) if c.k > 0 else (c.cout, c.cin)
        sd[name + ".weight"] = torch.randn(shape, generator=g) * std
        if c.bias:
            sd[name + ".bias"] = torch.randn(c.cout, generator=g) * 0.02
        if c.bn:
            damp = 0.3 if name.endswith("conv3") else 1.0
            var_scale = 2.0 * input_rms ** 2 if name.endswith("stem.conv1") else 1.0
            sd[name + ".norm.weight"] = (0.7 + 0.3 * torch.rand(c.cout, generator=g)) * damp
            sd[name + ".norm.bias"] = torch.randn(c.cout, generator=g) * 0.05
            sd[name + ".norm.running_mean"] = torch.randn(c.cout, generator=g) * 0.05 * math.sqrt(var_scale)
            sd[name + ".norm.running_var"] = (0.6 + 0.8 * torch.rand(c.cout, generator=g)) * var_scale
    return sd


def perturb(sd: Dict[str, torch.Tensor], rel: float = 1e-3, seed: int = 7) -> "OrderedDict[str, torch.Tensor]":
    """teacher = student + rel * |w|_rms * N(0,1) on the weights (so that teacher != student)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith(".weight") and ".norm." not in k:
            out[k] = v + rel * v.pow(2).mean().sqrt() * torch.randn(v.shape, generator=g)
        else:
            out[k] = v.clone()
    return out


def _smooth_noise(h: int, w: int, g: torch.Generator) -> torch.Tensor:
    lo = torch.randn(3, (h + 15) // 16 + 1, (w + 15) // 16 + 1, generator=g)
    up = torch.nn.functional.interpolate(lo[None], size=(h, w), mode="bilinear", align_corners=False)[0]
    return up


def make_image(h: int, w: int, n_boxes: int, num_classes: int, g: torch.Generator, min_size: float = 24.0, max_size: float = 320.0):
    """uint8 BGR CHW image = clip(128 + 40*lowpass noise + 12*white noise + rectangles) with `n_boxes` GT rectangles."""
    img = 128.0 + 40.0 * _smooth_noise(h, w, g) + 12.0 * torch.randn(3, h, w, generator=g)
    boxes, classes = [], []
    for _ in range(n_boxes):
        bw = math.exp(math.log(min_size) + torch.rand(1, generator=g).item() * (math.log(min(max_size, w * 0.8)) - math.log(min_size)))
        bh = math.exp(math.log(min_size) + torch.rand(1, generator=g).item() * (math.log(min(max_size, h * 0.8)) - math.log(min_size)))
        x1 = torch.rand(1, generator=g).item() * (w - bw)
        y1 = torch.rand(1, generator=g).item() * (h - bh)
        x2, y2 = x1 + bw, y1 + bh
        col = torch.randn(3, 1, 1, generator=g) * 50.0
        img[:, int(y1):int(y2) + 1, int(x1):int(x2) + 1] += col
        boxes.append([x1, y1, x2, y2])
        classes.append(int(torch.randint(0, num_classes, (1,), generator=g)))
    img = img.clamp(0, 255).to(torch.uint8)
    inst = {"image_size": (h, w), "gt_boxes": torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4),
            "gt_classes": torch.tensor(classes, dtype=torch.int64)}
    return img, inst


def strong_view(img: torch.Tensor, g: torch.Generator, block: int = 32, ratio: float = 0.5) -> torch.Tensor:
    """weak -> strong: per-channel gain/offset + MIC-style block mask (same geometry; mimics aldi/aug.py:45-59,154-176)."""
    x = img.to(torch.float32)
    gain = 0.8 + 0.4 * torch.rand(3, 1, 1, generator=g)
    off = 20.0 * (torch.rand(3, 1, 1, generator=g) - 0.5)
    x = x * gain + off
    h, w = x.shape[1:]
    m = (torch.rand((h + block - 1) // block, (w + block - 1) // block, generator=g) > ratio).to(torch.float32)
    m = m.repeat_interleave(block, 0).repeat_interleave(block, 1)[:h, :w]
    x = x * m[None]
    return x.clamp(0, 255).to(torch.uint8)


def make_batch(n_labeled: int, n_unlabeled: int, h: int, w: int, num_classes: int, seed: int = 0,
               boxes_per_image: Tuple[int, int] = (5, 20)):
    """-> (labeled_weak=None, labeled_strong, unlabeled_weak, unlabeled_strong) like aldi/dataloader.py:57-80
    with BATCH_CONTENTS = ("labeled_strong", "unlabeled_strong")."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = boxes_per_image
    labeled = []
    for _ in range(n_labeled):
        nb = int(torch.randint(lo, hi + 1, (1,), generator=g))
        img, inst = make_image(h, w, nb, num_classes, g)
        labeled.append({"image": strong_view(img, g), "instances": inst})
    uw, us = [], []
    for _ in range(n_unlabeled):
        nb = int(torch.randint(lo, hi + 1, (1,), generator=g))
        img, _ = make_image(h, w, nb, num_classes, g)
        empty = {"image_size": (h, w), "gt_boxes": torch.zeros(0, 4), "gt_classes": torch.zeros(0, dtype=torch.int64)}
        uw.append({"image": img, "instances": dict(empty)})
        us.append({"image": strong_view(img, g), "instances": dict(empty)})
    return None, labeled, (uw if n_unlabeled else None), (us if n_unlabeled else None)


def clone_batch(data):
    out = []
    for part in data:
        if part is None:
            out.append(None)
        else:
            out.append([{"image": d["image"].clone(), "instances": {k: (v.clone() if isinstance(v, torch.Tensor) else v)
                                                                    for k, v in d["instances"].items()}} for d in part])
    return tuple(out)
