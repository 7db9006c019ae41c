"""COCO box-AP evaluation of a (teacher) model -- the host-side step after the hot path.

Mirrors what the reference reaches through `aldi/trainer.py:166-196`: `build_evaluator` -> `Detectron2COCOEvaluatorAdapter`
(`aldi/helpers.py:65-81`, a detectron2 `COCOEvaluator` that fills in missing `iscrowd` / `area`), `test()` on the EMA model, and
the `bbox/AP50` key `BestCheckpointer` watches.  detectron2 and pycocotools are not vendored in the reference tree (and absent
from this image), so the metric is a restatement of pycocotools' published `COCOeval` algorithm for iouType="bbox"
(evaluateImg greedy matching, accumulate's 101-point interpolated precision, summarize's 12 numbers); it is cross-checked
against an independent loop implementation in oracle/coco_eval.py and hand-computed cases -- "parity unpinned" against
pycocotools itself.

Detections are consumed as the engine produces them (boxes in network-input pixels, XYXY) and mapped back to the original
image size like detectron2's `detector_postprocess` (scale by original/network size, clip).
"""
from __future__ import annotations

from collections import OrderedDict, defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
MAX_DETS = (1, 10, 100)
AREA_RNG = OrderedDict([("all", (0.0, 1e10)), ("small", (0.0, 32.0 ** 2)), ("medium", (32.0 ** 2, 96.0 ** 2)), ("large", (96.0 ** 2, 1e10))])


def maybe_add_optional_annotations(annotations: List[dict]) -> None:
    """reference aldi/helpers.py:65-70, verbatim semantics (including `area = bbox[1] * bbox[2]`, i.e. y * w, for annotations
    that carry no area -- kept so that numbers are comparable with the reference's logs)"""
    for ann in annotations:
        if "iscrowd" not in ann:
            ann["iscrowd"] = 0
        if "area" not in ann:
            ann["area"] = ann["bbox"][1] * ann["bbox"][2]


def iou_xywh(d: np.ndarray, g: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    """(D, 4) x (G, 4) XYWH -> (D, G); against a crowd box the union is the detection's own area (maskUtils.iou)"""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    dx2, dy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3]
    gx2, gy2 = g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    iw = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = iw * ih
    da, ga = d[:, 2] * d[:, 3], g[:, 2] * g[:, 3]
    union = np.where(crowd[None], da[:, None], da[:, None] + ga[None] - inter)
    return inter / union


def evaluate_img(dt: List[dict], gt: List[dict], area_rng, max_det: int):
    """COCOeval.evaluateImg for one (image, category): -> dict(scores, matched [T, D], dt_ignore [T, D], num_gt) or None"""
    if not dt and not gt:
        return None
    g_ignore = np.array([bool(g.get("ignore", 0)) or bool(g["iscrowd"]) or g["area"] < area_rng[0] or g["area"] > area_rng[1] for g in gt], dtype=bool)
    gorder = np.argsort(g_ignore, kind="mergesort")                       # non-ignored ground truth first
    gt = [gt[i] for i in gorder]
    g_ignore = g_ignore[gorder]
    dorder = np.argsort([-d["score"] for d in dt], kind="mergesort")[:max_det]
    dt = [dt[i] for i in dorder]
    crowd = np.array([bool(g["iscrowd"]) for g in gt], dtype=bool)
    ious = iou_xywh(np.array([d["bbox"] for d in dt], dtype=np.float64).reshape(-1, 4),
                    np.array([g["bbox"] for g in gt], dtype=np.float64).reshape(-1, 4), crowd)
    T, D, G = len(IOU_THRS), len(dt), len(gt)
    dtm = -np.ones((T, D), dtype=np.int64)
    gtm = -np.ones((T, G), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(IOU_THRS):
        for di in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] >= 0 and not crowd[gi]:
                    continue                                    # already matched (a crowd box can absorb many detections)
                if m > -1 and not g_ignore[m] and g_ignore[gi]:
                    break                                       # matched a regular gt; the rest are ignore boxes
                if ious[di, gi] < best:
                    continue
                best, m = ious[di, gi], gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ignore[m]
            dtm[ti, di] = m
            gtm[ti, m] = di
    d_area = np.array([d["bbox"][2] * d["bbox"][3] for d in dt], dtype=np.float64)
    out_rng = (d_area < area_rng[0]) | (d_area > area_rng[1])
    dt_ig = dt_ig | ((dtm < 0) & out_rng[None])
    return dict(scores=np.array([d["score"] for d in dt], dtype=np.float64), matched=dtm >= 0, dt_ignore=dt_ig, num_gt=int((~g_ignore).sum()))


def accumulate(per_img: List[Optional[dict]]):
    """COCOeval.accumulate for one (category, area range, maxDet): -> (precision [T, R], recall [T]) or None without gt"""
    ev = [e for e in per_img if e is not None]
    if not ev:
        return None
    npig = sum(e["num_gt"] for e in ev)
    if npig == 0:
        return None
    scores = np.concatenate([e["scores"] for e in ev])
    order = np.argsort(-scores, kind="mergesort")
    matched = np.concatenate([e["matched"] for e in ev], axis=1)[:, order]
    ignore = np.concatenate([e["dt_ignore"] for e in ev], axis=1)[:, order]
    tps = np.cumsum(matched & ~ignore, axis=1).astype(np.float64)
    fps = np.cumsum(~matched & ~ignore, axis=1).astype(np.float64)
    T, R = len(IOU_THRS), len(REC_THRS)
    precision = np.zeros((T, R))
    recall = np.zeros(T)
    for ti in range(T):
        tp, fp = tps[ti], fps[ti]
        nd = len(tp)
        rc = tp / npig
        pr = tp / (fp + tp + np.spacing(1))
        recall[ti] = rc[-1] if nd else 0.0
        pr = pr.tolist()
        for i in range(nd - 1, 0, -1):                           # precision envelope
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        inds = np.searchsorted(rc, REC_THRS, side="left")
        q = np.zeros(R)
        for ri, pi in enumerate(inds):
            if pi < nd:
                q[ri] = pr[pi]
        precision[ti] = q
    return precision, recall


def coco_bbox_metrics(images: Sequence[dict], annotations: List[dict], detections: List[dict], category_ids: Sequence[int]) -> "OrderedDict[str, float]":
    """images: dict(id); annotations: dict(image_id, category_id, bbox XYWH[, iscrowd, area, ignore]); detections:
    dict(image_id, category_id, bbox XYWH, score) -> detectron2's `bbox` result dict (AP, AP50, AP75, APs, APm, APl in percent)"""
    maybe_add_optional_annotations(annotations)
    gts, dts = defaultdict(list), defaultdict(list)
    for a in annotations:
        gts[a["image_id"], a["category_id"]].append(a)
    for d in detections:
        dts[d["image_id"], d["category_id"]].append(d)
    img_ids = [im["id"] for im in images]
    prec: Dict[tuple, Optional[tuple]] = {}
    for c in category_ids:
        for an, rng in AREA_RNG.items():
            per_img = [evaluate_img(dts.get((i, c), []), gts.get((i, c), []), rng, MAX_DETS[-1]) for i in img_ids]
            prec[c, an] = accumulate(per_img)

    def summarize(area="all", iou=None):
        vals = []
        for c in category_ids:
            r = prec[c, area]
            if r is None:
                continue
            p = r[0]
            if iou is not None:
                p = p[np.isclose(IOU_THRS, iou)]
            vals.append(p)
        if not vals:
            return float("nan")
        return float(np.mean(np.stack(vals))) * 100.0
    return OrderedDict([("AP", summarize()), ("AP50", summarize(iou=0.5)), ("AP75", summarize(iou=0.75)), ("APs", summarize("small")),
                        ("APm", summarize("medium")), ("APl", summarize("large"))])


class Detectron2COCOEvaluatorAdapter:
    """reset / process / evaluate with detectron2 COCOEvaluator's shapes (reference aldi/helpers.py:72-81).  `dataset_dicts`:
    detectron2-format records (file_name?, image_id, height, width, annotations=[dict(bbox XYWH_ABS or XYXY_ABS, bbox_mode,
    category_id[, iscrowd, area])]); contiguous category ids 0..K-1."""

    def __init__(self, dataset_name: str, dataset_dicts: Sequence[dict], num_classes: int, output_dir: Optional[str] = None,
                 distributed: bool = True):
        self.dataset_name, self.output_dir, self.distributed = dataset_name, output_dir, distributed
        self.num_classes = num_classes
        self.images = [dict(id=r["image_id"], height=r["height"], width=r["width"]) for r in dataset_dicts]
        self.annotations = []
        for r in dataset_dicts:
            for a in r.get("annotations", []):
                x, y, w, h = a["bbox"]
                if a.get("bbox_mode", "XYWH_ABS") in ("XYXY_ABS", 0):
                    w, h = w - x, h - y
                ann = dict(image_id=r["image_id"], category_id=a["category_id"], bbox=[float(x), float(y), float(w), float(h)])
                for k in ("iscrowd", "area"):
                    if k in a:
                        ann[k] = a[k]
                self.annotations.append(ann)
        self.reset()

    def reset(self):
        self._predictions: List[dict] = []

    def process(self, inputs: Sequence[dict], outputs: Sequence) -> None:
        """inputs: dict(image_id, height, width) of the ORIGINAL image; outputs: Instances (pred_boxes XYXY in network-input
        pixels, scores, pred_classes, image_size) as `GeneralizedRCNN.inference` returns them, or {"instances": Instances}"""
        for inp, out in zip(inputs, outputs):
            inst = out["instances"] if isinstance(out, dict) else out
            boxes = inst.pred_boxes.tensor.detach().float().cpu().numpy().reshape(-1, 4).astype(np.float64)
            ih, iw = inst.image_size
            sx, sy = inp["width"] / iw, inp["height"] / ih                      # detector_postprocess: back to the original size
            boxes[:, 0::2] = np.clip(boxes[:, 0::2] * sx, 0, inp["width"])
            boxes[:, 1::2] = np.clip(boxes[:, 1::2] * sy, 0, inp["height"])
            scores = inst.scores.detach().float().cpu().numpy()
            classes = inst.pred_classes.detach().cpu().numpy()
            keep = (boxes[:, 2] > boxes[:, 0]) & (boxes[:, 3] > boxes[:, 1])    # nonempty()
            for b, s, c in zip(boxes[keep], scores[keep], classes[keep]):
                self._predictions.append(dict(image_id=inp["image_id"], category_id=int(c), score=float(s),
                                              bbox=[float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]))

    def evaluate(self) -> "OrderedDict[str, dict]":
        preds = self._predictions
        if self.distributed:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                gathered = [None] * dist.get_world_size()
                dist.all_gather_object(gathered, preds)
                preds = [p for part in gathered for p in part]
                if dist.get_rank() != 0:
                    return OrderedDict()
        res = coco_bbox_metrics(self.images, [dict(a) for a in self.annotations], preds, list(range(self.num_classes)))
        return OrderedDict(bbox=res)


def inference_on_dataset(model, data_loader, evaluator) -> "OrderedDict[str, dict]":
    """detectron2.evaluation.inference_on_dataset: eval-mode forward over the loader, evaluator.process per batch"""
    was_training = getattr(model, "training", False)
    model.eval()
    evaluator.reset()
    try:
        for inputs in data_loader:
            outputs = model(inputs)
            evaluator.process(inputs, outputs)
    finally:
        model.train(was_training)
    return evaluator.evaluate()
