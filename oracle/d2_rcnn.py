"""CPU restatement (torch fp32) of the Detectron2 R50-FPN Faster R-CNN path that
the reference executes through its third-party ``detectron2`` dependency.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**:
``detectron2 @ git+https://github.com/justinkay/detectron2_v07ish.git`` is not
vendored (reference ``pyproject.toml:18``), not pinned, and not installable
here; this file restates the published Detectron2 (v0.6-era) algorithm as
summarised in SURVEY.md Appendix A and is anchored on the reference's call
sites:

* ``GeneralizedRCNN.forward``  <- reference ``aldi/align.py:72``, ``aldi/distill.py:157,162``
* ``GeneralizedRCNN.inference`` <- reference ``aldi/pseudolabeler.py:21``
* ``RPN.label_and_sample_anchors`` <- reference ``aldi/distill.py:200-202``
* hyper-parameters <- reference ``configs/detectron2/Base-RCNN-FPN.yaml:1-31``,
  ``configs/Base-RCNN-FPN.yaml:1-25``

Where Detectron2 leaves an order unspecified (``topk`` / ``sort`` ties,
``max`` ties) this oracle defines it as "first / lowest index wins" (stable
descending sort); the HIP path matches that definition bit-for-bit.

Data conventions: images are uint8/float CHW tensors (BGR); an ``instances``
record is a dict ``{"image_size": (h, w), "gt_boxes": (G,4) f32 XYXY,
"gt_classes": (G,) int64}``.  All random draws use the global torch CPU
generator in exactly the order Detectron2 makes them (``torch.randperm``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# configuration (Detectron2 defaults selected by the reference's FPN configs)
# ----------------------------------------------------------------------------

DEFAULTS = dict(
    num_classes=80,                      # D2 default; Cityscapes configs set 8
    pixel_mean=(103.530, 116.280, 123.675),
    pixel_std=(1.0, 1.0, 1.0),
    size_divisibility=32,
    stage_blocks=(3, 4, 6, 3),
    stage_mid=(64, 128, 256, 512),
    stage_out=(256, 512, 1024, 2048),
    fpn_channels=256,
    anchor_sizes=(32, 64, 128, 256, 512),
    anchor_ratios=(0.5, 1.0, 2.0),
    anchor_strides=(4, 8, 16, 32, 64),
    rpn_iou_thresholds=(0.3, 0.7),
    rpn_batch_per_image=256,
    rpn_positive_fraction=0.5,
    rpn_pre_nms_topk=(2000, 1000),       # (train, test)
    rpn_post_nms_topk=(1000, 1000),
    rpn_nms_thresh=0.7,
    rpn_bbox_weights=(1.0, 1.0, 1.0, 1.0),
    roi_iou_threshold=0.5,
    roi_batch_per_image=512,
    roi_positive_fraction=0.25,
    roi_bbox_weights=(10.0, 10.0, 5.0, 5.0),
    pooler_resolution=7,
    pooler_scales=(1 / 4, 1 / 8, 1 / 16, 1 / 32),
    fc_dim=1024,
    score_thresh_test=0.05,
    nms_thresh_test=0.5,
    detections_per_image=100,
)

SCALE_CLAMP = math.log(1000.0 / 16)
BN_EPS = 1e-5


def make_cfg(**over):
    cfg = dict(DEFAULTS)
    cfg.update(over)
    return cfg


# ----------------------------------------------------------------------------
# parameters (Detectron2 state_dict key names)
# ----------------------------------------------------------------------------

def trainable_keys(cfg, sd) -> List[str]:
    """FREEZE_AT=2: stem and res2 frozen; FrozenBN buffers are never trained."""
    out = []
    for k in sd:
        if ".norm." in k:
            continue
        if k.startswith("backbone.bottom_up.stem") or k.startswith("backbone.bottom_up.res2"):
            continue
        out.append(k)
    return out


# ----------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------

def frozen_bn(x, sd, p):
    scale = sd[p + ".norm.weight"] * (sd[p + ".norm.running_var"] + BN_EPS).rsqrt()
    bias = sd[p + ".norm.bias"] - sd[p + ".norm.running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def conv_bn(x, sd, p, stride=1, pad=0):
    return frozen_bn(F.conv2d(x, sd[p + ".weight"], None, stride, pad), sd, p)


def preprocess(cfg, images: List[torch.Tensor]) -> Tuple[torch.Tensor, List[Tuple[int, int]]]:
    """(x - mean) / std, zero pad bottom/right to a multiple of 32 [D2 ImageList.from_tensors]."""
    mean = torch.tensor(cfg["pixel_mean"], dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(cfg["pixel_std"], dtype=torch.float32).view(3, 1, 1)
    xs = [(im.to(torch.float32) - mean) / std for im in images]
    sizes = [(int(x.shape[1]), int(x.shape[2])) for x in xs]
    d = cfg["size_divisibility"]
    H = max(s[0] for s in sizes)
    W = max(s[1] for s in sizes)
    H = (H + d - 1) // d * d
    W = (W + d - 1) // d * d
    out = torch.zeros(len(xs), 3, H, W, dtype=torch.float32)
    for i, x in enumerate(xs):
        out[i, :, : x.shape[1], : x.shape[2]] = x
    return out, sizes


def resnet_fpn(cfg, sd, x, stages_out: Optional[list] = None) -> "OrderedDict[str, torch.Tensor]":
    """`stages_out` (a list) receives the bottom-up stage outputs res2..res5 (secondary cross-check against an independent ResNet)"""
    bu = "backbone.bottom_up."
    x = F.relu(conv_bn(x, sd, bu + "stem.conv1", 2, 3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    cs = []
    for si, nb in enumerate(cfg["stage_blocks"]):
        stage = f"res{si + 2}"
        for b in range(nb):
            stride = 2 if (b == 0 and si > 0) else 1
            p = f"{bu}{stage}.{b}."
            sc = conv_bn(x, sd, p + "shortcut", stride, 0) if b == 0 else x
            h = F.relu(conv_bn(x, sd, p + "conv1", stride, 0))
            h = F.relu(conv_bn(h, sd, p + "conv2", 1, 1))
            h = conv_bn(h, sd, p + "conv3", 1, 0)
            x = F.relu(h + sc)
        cs.append(x)
    if stages_out is not None:
        stages_out.extend(cs)
    # FPN top-down, FUSE_TYPE="sum", NORM=""
    feats = OrderedDict()
    prev = F.conv2d(cs[3], sd["backbone.fpn_lateral5.weight"], sd["backbone.fpn_lateral5.bias"])
    outs = {5: F.conv2d(prev, sd["backbone.fpn_output5.weight"], sd["backbone.fpn_output5.bias"], 1, 1)}
    for lvl in (4, 3, 2):
        top = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        lat = F.conv2d(cs[lvl - 2], sd[f"backbone.fpn_lateral{lvl}.weight"], sd[f"backbone.fpn_lateral{lvl}.bias"])
        prev = lat + top
        outs[lvl] = F.conv2d(prev, sd[f"backbone.fpn_output{lvl}.weight"], sd[f"backbone.fpn_output{lvl}.bias"], 1, 1)
    for lvl in (2, 3, 4, 5):
        feats[f"p{lvl}"] = outs[lvl]
    feats["p6"] = F.max_pool2d(outs[5], kernel_size=1, stride=2, padding=0)   # LastLevelMaxPool
    return feats


# ----------------------------------------------------------------------------
# boxes
# ----------------------------------------------------------------------------

def generate_anchors(cfg, feat_shapes: List[Tuple[int, int]]) -> List[torch.Tensor]:
    """DefaultAnchorGenerator, offset 0: per level (H*W*A, 4), order (H, W, A)."""
    out = []
    for (h, w), size, stride in zip(feat_shapes, cfg["anchor_sizes"], cfg["anchor_strides"]):
        cell = []
        for r in cfg["anchor_ratios"]:
            area = size ** 2.0
            aw = math.sqrt(area / r)
            ah = r * aw
            cell.append([-aw / 2.0, -ah / 2.0, aw / 2.0, ah / 2.0])
        cell = torch.tensor(cell, dtype=torch.float32)
        sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx = xx.reshape(-1)
        yy = yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return out


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def pairwise_iou(b1, b2):
    """Detectron2 ``pairwise_iou`` (G, M)."""
    area1 = box_area(b1)
    area2 = box_area(b2)
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    iou = torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype))
    return iou


def get_deltas(src, tgt, weights):
    wx, wy, ww, wh = weights
    sw = src[:, 2] - src[:, 0]
    sh = src[:, 3] - src[:, 1]
    sx = src[:, 0] + 0.5 * sw
    sy = src[:, 1] + 0.5 * sh
    tw = tgt[:, 2] - tgt[:, 0]
    th = tgt[:, 3] - tgt[:, 1]
    tx = tgt[:, 0] + 0.5 * tw
    ty = tgt[:, 1] + 0.5 * th
    dx = wx * (tx - sx) / sw
    dy = wy * (ty - sy) / sh
    dw = ww * torch.log(tw / sw)
    dh = wh * torch.log(th / sh)
    return torch.stack((dx, dy, dw, dh), dim=1)


def apply_deltas(deltas, boxes, weights):
    """deltas (R, 4k), boxes (R, 4) -> (R, 4k)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def clip_boxes(b, size):
    h, w = size
    x1 = b[:, 0].clamp(min=0, max=w)
    y1 = b[:, 1].clamp(min=0, max=h)
    x2 = b[:, 2].clamp(min=0, max=w)
    y2 = b[:, 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=-1)


def stable_desc_order(scores):
    """Oracle definition of topk/sort order: descending, ties -> lower index first."""
    return torch.sort(scores, descending=True, stable=True)[1]


def _clib():
    """liboracle.so (oracle/roi_align.c) if it has been built (make -C oracle), else None -> pure torch paths."""
    global _CLIB
    if _CLIB is False:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle.so")
        _CLIB = ctypes.CDLL(path) if os.path.exists(path) else None
    return _CLIB


_CLIB = False


def nms(boxes, scores, thresh) -> torch.Tensor:
    """torchvision.ops.nms semantics; kept indices in descending-score order.
    IoU > thresh suppresses; IoU = inter/(a1+a2-inter)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    lib = _clib()
    if lib is not None:
        import ctypes
        order = stable_desc_order(scores)
        b = boxes[order].contiguous().to(torch.float32)
        keep = torch.empty(n, dtype=torch.uint8)
        lib.oracle_nms_sorted(ctypes.c_void_p(b.data_ptr()), ctypes.c_int(n), ctypes.c_float(thresh), ctypes.c_void_p(keep.data_ptr()))
        return order[keep.bool()]
    return nms_torch(boxes, scores, thresh)


def nms_torch(boxes, scores, thresh) -> torch.Tensor:
    """Blocked pure-torch formulation of the same rule."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = stable_desc_order(scores)
    b = boxes[order]
    area = box_area(b)
    keep_mask = torch.ones(n, dtype=torch.bool)
    B = 512
    suppressed = torch.zeros(n, dtype=torch.bool)
    for s in range(0, n, B):
        e = min(s + B, n)
        # resolve inside the block sequentially, then suppress everything after it
        blk = b[s:e]
        lt = torch.max(blk[:, None, :2], b[None, s:, :2])
        rb = torch.min(blk[:, None, 2:], b[None, s:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        iou = inter / (area[s:e, None] + area[None, s:] - inter)
        over = iou > thresh                       # (e-s, n-s)
        for i in range(e - s):
            if suppressed[s + i]:
                continue
            row = over[i].clone()
            row[: i + 1] = False
            suppressed[s:] |= row
    keep_mask = ~suppressed
    return order[keep_mask]


def batched_nms(boxes, scores, idxs, thresh) -> torch.Tensor:
    """torchvision ``_batched_nms_vanilla``: per-category NMS, result sorted by score.

    (torchvision switches to the coordinate-offset trick for small inputs; the
    vanilla form is the one defined here -- exact IoU on raw coordinates.)"""
    keep_mask = torch.zeros(scores.shape[0], dtype=torch.bool)
    for c in torch.unique(idxs):
        ci = torch.where(idxs == c)[0]
        k = nms(boxes[ci], scores[ci], thresh)
        keep_mask[ci[k]] = True
    keep = torch.where(keep_mask)[0]
    return keep[stable_desc_order(scores[keep])]


# ----------------------------------------------------------------------------
# matcher / sampler
# ----------------------------------------------------------------------------

def matcher(mqm, thresholds, labels, allow_low_quality):
    """Detectron2 ``Matcher.__call__``. mqm (G, M)."""
    M = mqm.shape[1]
    if mqm.numel() == 0:
        return torch.zeros(M, dtype=torch.int64), torch.full((M,), labels[0], dtype=torch.int8)
    matched_vals, matches = mqm.max(dim=0)
    match_labels = torch.full((M,), 1, dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, low, high in zip(labels, th[:-1], th[1:]):
        sel = (matched_vals >= low) & (matched_vals < high)
        match_labels[sel] = l
    if allow_low_quality:
        best_per_gt, _ = mqm.max(dim=1)
        hit = (mqm == best_per_gt[:, None]).any(dim=0)
        match_labels[hit] = 1
    return matches, match_labels


def subsample_labels(labels, num_samples, positive_fraction, bg_label):
    positive = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    negative = torch.nonzero(labels == bg_label).squeeze(1)
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = torch.randperm(positive.numel())[:num_pos]
    perm2 = torch.randperm(negative.numel())[:num_neg]
    return positive[perm1], negative[perm2]


# ----------------------------------------------------------------------------
# RPN
# ----------------------------------------------------------------------------

def rpn_head(cfg, sd, feats: List[torch.Tensor]):
    """StandardRPNHead: raw outputs, (N, A, H, W) and (N, 4A, H, W) per level."""
    p = "proposal_generator.rpn_head."
    logits, deltas = [], []
    for x in feats:
        t = F.relu(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 1))
        logits.append(F.conv2d(t, sd[p + "objectness_logits.weight"], sd[p + "objectness_logits.bias"]))
        deltas.append(F.conv2d(t, sd[p + "anchor_deltas.weight"], sd[p + "anchor_deltas.bias"]))
    return logits, deltas


def rpn_permute(logits, deltas):
    """(N,A,H,W)->(N,HWA) ; (N,4A,H,W)->(N,HWA,4)  [D2 RPN.forward]."""
    lo = [s.permute(0, 2, 3, 1).flatten(1) for s in logits]
    de = [x.view(x.shape[0], -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2) for x in deltas]
    return lo, de


def label_and_sample_anchors(cfg, anchors: List[torch.Tensor], gt_instances: List[dict]):
    """D2 ``RPN.label_and_sample_anchors`` -> (list of (sumA,) int8 labels, list of (sumA,4) matched gt)."""
    allanch = torch.cat(anchors)
    gt_labels, matched_gt = [], []
    for inst in gt_instances:
        gtb = inst["gt_boxes"].to(torch.float32).reshape(-1, 4)
        mqm = pairwise_iou(gtb, allanch)
        midx, lab = matcher(mqm, cfg["rpn_iou_thresholds"], [0, -1, 1], True)
        pos, neg = subsample_labels(lab, cfg["rpn_batch_per_image"], cfg["rpn_positive_fraction"], 0)
        lab.fill_(-1)
        lab.scatter_(0, pos, 1)
        lab.scatter_(0, neg, 0)
        if gtb.shape[0] == 0:
            mg = torch.zeros_like(allanch)
        else:
            mg = gtb[midx]
        gt_labels.append(lab)
        matched_gt.append(mg)
    return gt_labels, matched_gt


def rpn_losses(cfg, anchors, logits_p, deltas_p, gt_labels, gt_boxes):
    N = len(gt_labels)
    gl = torch.stack(gt_labels)
    pos = gl == 1
    allanch = torch.cat(anchors)
    tgt = torch.stack([get_deltas(allanch, g, cfg["rpn_bbox_weights"]) for g in gt_boxes])
    pred = torch.cat(deltas_p, dim=1)
    loc = (pred[pos] - tgt[pos]).abs().sum()           # smooth_l1 beta=0 == L1, reduction sum
    valid = gl >= 0
    obj = F.binary_cross_entropy_with_logits(torch.cat(logits_p, dim=1)[valid], gl[valid].to(torch.float32), reduction="sum")
    norm = cfg["rpn_batch_per_image"] * N
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}


def find_top_rpn_proposals(cfg, anchors, logits_p, deltas_p, image_sizes, training: bool):
    """Returns list of dict(proposal_boxes (P,4), objectness_logits (P,)), detached."""
    pre = cfg["rpn_pre_nms_topk"][0 if training else 1]
    post = cfg["rpn_post_nms_topk"][0 if training else 1]
    N = logits_p[0].shape[0]
    out = []
    with torch.no_grad():
        for n in range(N):
            boxes, scores, lvls = [], [], []
            for l, (a, lo, de) in enumerate(zip(anchors, logits_p, deltas_p)):
                k = min(lo.shape[1], pre)
                order = stable_desc_order(lo[n])[:k]
                sc = lo[n][order]
                bx = apply_deltas(de[n][order], a[order], cfg["rpn_bbox_weights"])
                boxes.append(bx)
                scores.append(sc)
                lvls.append(torch.full((k,), l, dtype=torch.int64))
            boxes = torch.cat(boxes)
            scores = torch.cat(scores)
            lvls = torch.cat(lvls)
            valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
            if not valid.all():
                if training:
                    raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
                boxes, scores, lvls = boxes[valid], scores[valid], lvls[valid]
            boxes = clip_boxes(boxes, image_sizes[n])
            keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
            if not keep.all():
                boxes, scores, lvls = boxes[keep], scores[keep], lvls[keep]
            k = batched_nms(boxes, scores, lvls, cfg["rpn_nms_thresh"])[:post]
            out.append({"proposal_boxes": boxes[k], "objectness_logits": scores[k], "image_size": image_sizes[n]})
    return out


# ----------------------------------------------------------------------------
# ROI heads
# ----------------------------------------------------------------------------

class _RoiAlignC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, scale):
        import ctypes
        lib = _clib()
        feat = feat.contiguous()
        rois = rois.contiguous().to(torch.float32)
        N, C, H, W = feat.shape
        R = rois.shape[0]
        out = torch.empty(R, C, out_size, out_size, dtype=torch.float32)
        lib.oracle_roi_align_forward(ctypes.c_void_p(feat.data_ptr()), N, C, H, W, ctypes.c_void_p(rois.data_ptr()), R, out_size,
                                     ctypes.c_float(scale), ctypes.c_void_p(out.data_ptr()))
        ctx.save_for_backward(rois)
        ctx.meta = (N, C, H, W, out_size, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes
        lib = _clib()
        (rois,) = ctx.saved_tensors
        N, C, H, W, P, scale = ctx.meta
        gout = gout.contiguous()
        gfeat = torch.zeros(N, C, H, W, dtype=torch.float32)
        lib.oracle_roi_align_backward(ctypes.c_void_p(gout.data_ptr()), N, C, H, W, ctypes.c_void_p(rois.data_ptr()), rois.shape[0], P,
                                      ctypes.c_float(scale), ctypes.c_void_p(gfeat.data_ptr()))
        return gfeat, None, None, None


def roi_align(feat, rois, out_size, scale):
    """torchvision roi_align, aligned=True, sampling_ratio=0. feat (N,C,H,W); rois (R,5) [b,x1,y1,x2,y2]."""
    if _clib() is not None and rois.shape[0] > 0 and feat.dtype == torch.float32:
        return _RoiAlignC.apply(feat, rois, out_size, float(scale))
    return roi_align_torch(feat, rois, out_size, scale)


def roi_align_torch(feat, rois, out_size, scale):
    """Pure-torch statement of the same algorithm; differentiable w.r.t. ``feat`` (gather + weighted sum)."""
    R = rois.shape[0]
    N, C, H, W = feat.shape
    P = out_size
    out = feat.new_zeros(R, C, P, P)
    if R == 0:
        return out
    b = rois[:, 0].long()
    x1 = rois[:, 1] * scale - 0.5
    y1 = rois[:, 2] * scale - 0.5
    x2 = rois[:, 3] * scale - 0.5
    y2 = rois[:, 4] * scale - 0.5
    rw = x2 - x1
    rh = y2 - y1
    bw = rw / P
    bh = rh / P
    gh = torch.ceil(rh / P).long()
    gw = torch.ceil(rw / P).long()
    flat = feat.permute(0, 2, 3, 1).reshape(N * H * W, C)
    for r in range(R):
        ngh, ngw = int(gh[r]), int(gw[r])
        count = max(ngh * ngw, 1)
        if ngh <= 0 or ngw <= 0:
            continue
        ph = torch.arange(P, dtype=torch.float32)
        iy = torch.arange(ngh, dtype=torch.float32)
        ix = torch.arange(ngw, dtype=torch.float32)
        ys = (y1[r] + ph[:, None] * bh[r] + (iy[None, :] + 0.5) * bh[r] / ngh).reshape(-1)   # (P*gh)
        xs = (x1[r] + ph[:, None] * bw[r] + (ix[None, :] + 0.5) * bw[r] / ngw).reshape(-1)   # (P*gw)

        def prep(v, size):
            oob = (v < -1.0) | (v > size)
            v = v.clamp(min=0)
            lo = v.floor().long()
            hi_clip = lo >= size - 1
            lo = torch.where(hi_clip, torch.full_like(lo, size - 1), lo)
            hi = torch.where(hi_clip, lo, lo + 1)
            v = torch.where(hi_clip, lo.to(v.dtype), v)
            l = v - lo.to(v.dtype)
            h = 1.0 - l
            return oob, lo, hi, l, h
        oy, ylo, yhi, ly, hy = prep(ys, H)
        ox, xlo, xhi, lx, hx = prep(xs, W)
        base = int(b[r]) * H * W
        def g(yi, xi):
            return flat[(base + yi[:, None] * W + xi[None, :]).reshape(-1)].reshape(yi.numel(), xi.numel(), C)
        w1 = hy[:, None] * hx[None, :]
        w2 = hy[:, None] * lx[None, :]
        w3 = ly[:, None] * hx[None, :]
        w4 = ly[:, None] * lx[None, :]
        val = (w1[..., None] * g(ylo, xlo) + w2[..., None] * g(ylo, xhi)
               + w3[..., None] * g(yhi, xlo) + w4[..., None] * g(yhi, xhi))
        dead = (oy[:, None] | ox[None, :])
        val = torch.where(dead[..., None], torch.zeros((), dtype=val.dtype), val)
        val = val.reshape(P, ngh, P, ngw, C).sum(dim=(1, 3)) / count
        out[r] = val.permute(2, 0, 1)
    return out


def assign_levels(boxes, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    sizes = torch.sqrt(box_area(boxes))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_pool(cfg, feats: List[torch.Tensor], box_lists: List[torch.Tensor]):
    """ROIPooler (ROIAlignV2) over p2..p5 -> (R, C, P, P)."""
    boxes = torch.cat(box_lists)
    bidx = torch.cat([torch.full((len(b),), i, dtype=torch.float32) for i, b in enumerate(box_lists)])
    rois = torch.cat([bidx[:, None], boxes], dim=1)
    P = cfg["pooler_resolution"]
    C = feats[0].shape[1]
    out = feats[0].new_zeros(rois.shape[0], C, P, P)
    if rois.shape[0] == 0:
        return out
    lv = assign_levels(boxes)
    pieces = []
    idxs = []
    for l, (f, s) in enumerate(zip(feats, cfg["pooler_scales"])):
        inds = torch.nonzero(lv == l).squeeze(1)
        if inds.numel() == 0:
            continue
        pieces.append(roi_align(f, rois[inds], P, s))
        idxs.append(inds)
    inds = torch.cat(idxs)
    vals = torch.cat(pieces)
    out = out.index_put((inds,), vals)
    return out


def box_head(sd, pooled):
    x = pooled.flatten(1)
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
    return x


def box_predictor(sd, x):
    p = "roi_heads.box_predictor."
    return (F.linear(x, sd[p + "cls_score.weight"], sd[p + "cls_score.bias"]),
            F.linear(x, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"]))


GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))


def label_and_sample_proposals(cfg, proposals: List[dict], targets: List[dict]):
    """StandardROIHeads.label_and_sample_proposals (PROPOSAL_APPEND_GT=True)."""
    K = cfg["num_classes"]
    out = []
    for prop, tgt in zip(proposals, targets):
        gtb = tgt["gt_boxes"].to(torch.float32).reshape(-1, 4)
        gtc = tgt["gt_classes"].to(torch.int64).reshape(-1)
        pb = prop["proposal_boxes"]
        if gtb.shape[0] > 0:                          # add_ground_truth_to_proposals
            pb = torch.cat([pb, gtb])
        mqm = pairwise_iou(gtb, pb)
        midx, mlab = matcher(mqm, [cfg["roi_iou_threshold"]], [0, 1], False)
        if gtc.numel() > 0:
            cls = gtc[midx]
            cls[mlab == 0] = K
            cls[mlab == -1] = -1
        else:
            cls = torch.zeros_like(midx) + K
        fg, bg = subsample_labels(cls, cfg["roi_batch_per_image"], cfg["roi_positive_fraction"], K)
        sidx = torch.cat([fg, bg])
        rec = {"proposal_boxes": pb[sidx], "gt_classes": cls[sidx], "sampled_idxs": sidx,
               "image_size": prop["image_size"]}
        if gtb.shape[0] > 0:
            rec["gt_boxes"] = gtb[midx[sidx]]
        else:
            rec["gt_boxes"] = torch.zeros(len(sidx), 4)
        out.append(rec)
    return out


def roi_losses(cfg, scores, deltas, sampled: List[dict]):
    K = cfg["num_classes"]
    gt_classes = torch.cat([s["gt_classes"] for s in sampled])
    pboxes = torch.cat([s["proposal_boxes"] for s in sampled])
    gboxes = torch.cat([s["gt_boxes"] for s in sampled])
    if gt_classes.numel() == 0:
        loss_cls = scores.sum() * 0.0
    else:
        loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean")
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < K)).squeeze(1)
    fgd = deltas.view(-1, K, 4)[fg, gt_classes[fg]]
    tgt = get_deltas(pboxes[fg], gboxes[fg], cfg["roi_bbox_weights"])
    loss_box = (fgd - tgt).abs().sum() / max(gt_classes.numel(), 1.0)
    return {"loss_cls": loss_cls, "loss_box_reg": loss_box}


def fast_rcnn_inference(cfg, scores, deltas, proposals: List[dict]):
    """FastRCNNOutputLayers.inference -> list of dict(pred_boxes, scores, pred_classes, image_size)."""
    K = cfg["num_classes"]
    counts = [len(p["proposal_boxes"]) for p in proposals]
    pb = torch.cat([p["proposal_boxes"] for p in proposals])
    boxes = apply_deltas(deltas, pb, cfg["roi_bbox_weights"])
    probs = F.softmax(scores, dim=-1)
    res = []
    for bx, pr, prop in zip(boxes.split(counts), probs.split(counts), proposals):
        valid = torch.isfinite(bx).all(dim=1) & torch.isfinite(pr).all(dim=1)
        if not valid.all():
            bx, pr = bx[valid], pr[valid]
        pr = pr[:, :-1]
        bx = clip_boxes(bx.reshape(-1, 4), prop["image_size"]).view(-1, K, 4)
        fmask = pr > cfg["score_thresh_test"]
        finds = fmask.nonzero()
        bsel = bx[fmask]
        ssel = pr[fmask]
        keep = batched_nms(bsel, ssel, finds[:, 1], cfg["nms_thresh_test"])
        keep = keep[: cfg["detections_per_image"]]
        res.append({"pred_boxes": bsel[keep], "scores": ssel[keep], "pred_classes": finds[keep, 1],
                    "image_size": prop["image_size"]})
    return res


# ----------------------------------------------------------------------------
# GeneralizedRCNN
# ----------------------------------------------------------------------------

class Captured(dict):
    """Intermediates the reference's forward hooks expose (aldi/distill.py:115-138, aldi/align.py:45-52)."""


def forward_train(cfg, sd, batched_inputs: List[dict], replace_proposals: Optional[List[dict]] = None,
                  roi_seed: Optional[int] = None, cap: Optional[Captured] = None, arch: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """GeneralizedRCNN.forward (training).  ``roi_seed`` mimics the ManualSeed
    pre-hook on roi_heads (aldi/helpers.py:17-26); ``replace_proposals`` the
    ReplaceProposalsOnce pre-hook (aldi/helpers.py:28-42)."""
    # `arch` swaps the architecture-specific callables (oracle/d2_vitdet.py: ViTDet trunk, 2-conv RPN head, conv+FC box head);
    # everything else -- anchors, matching, sampling, proposals, ROIAlign, losses -- is shared
    arch = arch or {}
    x, sizes = preprocess(cfg, [b["image"] for b in batched_inputs])
    x = x.to(next(iter(sd.values())).dtype)       # fp64 state_dict -> fp64 "truth" run (tests only)
    gts = [b["instances"] for b in batched_inputs]
    feats = arch.get("backbone", resnet_fpn)(cfg, sd, x)
    flist = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    anchors = generate_anchors(cfg, [tuple(f.shape[-2:]) for f in flist])
    logits, deltas = arch.get("rpn_head", rpn_head)(cfg, sd, flist)
    lo, de = rpn_permute(logits, deltas)
    gt_labels, gt_boxes = label_and_sample_anchors(cfg, anchors, gts)
    losses_rpn = rpn_losses(cfg, anchors, lo, de, gt_labels, gt_boxes)
    proposals = find_top_rpn_proposals(cfg, anchors, [t.detach() for t in lo], [t.detach() for t in de], sizes, True)
    if cap is not None:
        cap.update(features=feats, anchors=anchors, rpn_logits=logits, rpn_deltas=deltas,
                   rpn_gt_labels=gt_labels, proposals=proposals, image_sizes=sizes)
    if roi_seed is not None:
        torch.manual_seed(roi_seed)
    if replace_proposals is not None:
        proposals = replace_proposals
    if cap is not None:
        cap["proposals_used"] = proposals
    sampled = label_and_sample_proposals(cfg, proposals, gts)
    pooled = roi_pool(cfg, flist[:4], [s["proposal_boxes"] for s in sampled])
    bh = arch.get("box_head", box_head)(sd, pooled)
    scores, bdeltas = box_predictor(sd, bh)
    losses = roi_losses(cfg, scores, bdeltas, sampled)
    losses.update(losses_rpn)
    if cap is not None:
        cap.update(sampled=sampled, pooled=pooled, box_head_out=bh, box_scores=scores, box_deltas=bdeltas)
    return losses


def inference(cfg, sd, batched_inputs: List[dict], roi_seed: Optional[int] = None,
              cap: Optional[Captured] = None) -> List[dict]:
    """GeneralizedRCNN.inference(do_postprocess=False)."""
    with torch.no_grad():
        x, sizes = preprocess(cfg, [b["image"] for b in batched_inputs])
        feats = resnet_fpn(cfg, sd, x)
        flist = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        anchors = generate_anchors(cfg, [tuple(f.shape[-2:]) for f in flist])
        logits, deltas = rpn_head(cfg, sd, flist)
        lo, de = rpn_permute(logits, deltas)
        proposals = find_top_rpn_proposals(cfg, anchors, lo, de, sizes, False)
        if roi_seed is not None:
            torch.manual_seed(roi_seed)               # ManualSeed fires on every roi_heads forward (B.3)
        pooled = roi_pool(cfg, flist[:4], [p["proposal_boxes"] for p in proposals])
        bh = box_head(sd, pooled)
        scores, bdeltas = box_predictor(sd, bh)
        if cap is not None:
            cap.update(features=feats, anchors=anchors, rpn_logits=logits, rpn_deltas=deltas,
                       proposals=proposals, box_scores=scores, box_deltas=bdeltas, image_sizes=sizes)
        return fast_rcnn_inference(cfg, scores, bdeltas, proposals)


# ----------------------------------------------------------------------------
# solver (SGD momentum, D2 defaults)
# ----------------------------------------------------------------------------

def sgd_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], bufs: Dict[str, torch.Tensor],
             lr: float, momentum: float = 0.9, weight_decay: float = 1e-4):
    """torch.optim.SGD (nesterov off, dampening 0), in place."""
    with torch.no_grad():
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            d = g + weight_decay * p
            if k not in bufs:
                bufs[k] = d.clone()
            else:
                bufs[k].mul_(momentum).add_(d)
            p.add_(bufs[k], alpha=-lr)
