"""Loop restatement of pycocotools' COCOeval (iouType="bbox") used to cross-check aldi_amd/evaluation.py.

TEST INFRASTRUCTURE ONLY.  pycocotools 2.0.x is a dependency of detectron2's COCOEvaluator (reached from the reference at
aldi/helpers.py:72-81, aldi/trainer.py:166-171); it is neither vendored in /root/reference nor installed here, so this file
restates its published algorithm from the paper-and-API description (COCO detection evaluation: greedy score-ordered matching
per IoU threshold, crowd regions absorb detections, ignore regions by area, 101-point interpolated AP) -- "parity unpinned".
It is written scalar-by-scalar on purpose: one detection, one threshold, one recall point at a time.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

_REC = np.linspace(0.0, 1.0, 101)          # the recall grid is defined through linspace (0.29 != 29 * 0.01 in binary)


def _iou(d, g, crowd: bool) -> float:
    ix = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    iy = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if ix <= 0 or iy <= 0:
        return 0.0
    inter = ix * iy
    union = d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter
    return inter / union


def average_precision(images: Sequence[int], anns: List[dict], dets: List[dict], cat: int, thr: float, area=(0.0, 1e10), max_det: int = 100):
    """AP of one category at one IoU threshold and area range (fraction in [0, 1]), or None when the category has no countable gt"""
    rows = []           # (score, is_tp, ignored) over all images
    n_gt = 0
    for img in images:
        g = [a for a in anns if a["image_id"] == img and a["category_id"] == cat]
        d = [x for x in dets if x["image_id"] == img and x["category_id"] == cat]
        if not g and not d:
            continue
        ign = [bool(a.get("iscrowd", 0)) or a["area"] < area[0] or a["area"] > area[1] for a in g]
        order = sorted(range(len(g)), key=lambda i: ign[i])            # stable: countable gt first
        g = [g[i] for i in order]
        ign = [ign[i] for i in order]
        n_gt += sum(1 for v in ign if not v)
        d = sorted(d, key=lambda x: -x["score"])[:max_det]             # stable
        taken = [False] * len(g)
        for x in d:
            best, m = min(thr, 1 - 1e-10), -1
            for gi, a in enumerate(g):
                crowd = bool(a.get("iscrowd", 0))
                if taken[gi] and not crowd:
                    continue
                if m >= 0 and not ign[m] and ign[gi]:
                    break
                v = _iou(x["bbox"], a["bbox"], crowd)
                if v < best:
                    continue
                best, m = v, gi
            if m >= 0:
                taken[m] = True
                rows.append((x["score"], True, ign[m]))
            else:
                ar = x["bbox"][2] * x["bbox"][3]
                rows.append((x["score"], False, ar < area[0] or ar > area[1]))
    if n_gt == 0:
        return None
    rows.sort(key=lambda r: -r[0])                                      # stable
    tp = fp = 0
    rec, prec = [], []
    for _, is_tp, ignored in rows:
        if not ignored:
            if is_tp:
                tp += 1
            else:
                fp += 1
        rec.append(tp / n_gt)
        prec.append(tp / (tp + fp + 2.220446049250313e-16))
    for i in range(len(prec) - 1, 0, -1):
        if prec[i] > prec[i - 1]:
            prec[i - 1] = prec[i]
    total = 0.0
    for k in range(101):
        r = float(_REC[k])
        j = 0
        while j < len(rec) and rec[j] < r:
            j += 1
        total += prec[j] if j < len(rec) else 0.0
    return total / 101.0


def bbox_metrics(images: Sequence[int], anns: List[dict], dets: List[dict], cats: Sequence[int]) -> Dict[str, float]:
    thrs = [float(v) for v in np.linspace(0.5, 0.95, 10)]

    def mean_ap(thr_list, area):
        vals = []
        for c in cats:
            per = [average_precision(images, anns, dets, c, t, area) for t in thr_list]
            if per[0] is None:
                continue
            vals += per
        return float("nan") if not vals else 100.0 * sum(vals) / len(vals)
    return {"AP": mean_ap(thrs, (0.0, 1e10)), "AP50": mean_ap([thrs[0]], (0.0, 1e10)), "AP75": mean_ap([thrs[5]], (0.0, 1e10)),
            "APs": mean_ap(thrs, (0.0, 32.0 ** 2)), "APm": mean_ap(thrs, (32.0 ** 2, 96.0 ** 2)), "APl": mean_ap(thrs, (96.0 ** 2, 1e10))}
