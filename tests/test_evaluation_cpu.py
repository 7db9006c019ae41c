"""COCO box AP (aldi_amd/evaluation.py) vs hand-computed cases and vs the loop restatement in oracle/coco_eval.py."""
import copy

import numpy as np
import pytest
import torch


def _ann(img, cat, box, **kw):
    d = dict(image_id=img, category_id=cat, bbox=[float(v) for v in box], area=float(box[2] * box[3]))
    d.update(kw)
    return d


def _det(img, cat, box, score):
    return dict(image_id=img, category_id=cat, bbox=[float(v) for v in box], score=float(score))


def test_perfect_detections_score_100():
    from aldi_amd.evaluation import coco_bbox_metrics
    imgs = [dict(id=i) for i in range(3)]
    anns = [_ann(i, c, (10 + 5 * c, 20, 40 + 30 * c, 50 + 10 * i)) for i in range(3) for c in range(2)]
    dets = [_det(a["image_id"], a["category_id"], a["bbox"], 0.9 - 0.1 * k) for k, a in enumerate(anns)]
    res = coco_bbox_metrics(imgs, anns, dets, [0, 1])
    assert res["AP"] == pytest.approx(100.0) and res["AP50"] == pytest.approx(100.0) and res["AP75"] == pytest.approx(100.0)


def test_hand_computed_tp_fp_tp():
    """two gt; detections tp (0.9), fp (0.8), tp (0.7): recall [.5,.5,1], precision envelope [1, 2/3, 2/3]
    -> AP = (51 * 1 + 50 * 2/3) / 101 at every IoU threshold"""
    from aldi_amd.evaluation import coco_bbox_metrics
    imgs = [dict(id=0)]
    anns = [_ann(0, 0, (0, 0, 50, 50)), _ann(0, 0, (100, 100, 60, 60))]
    dets = [_det(0, 0, (0, 0, 50, 50), 0.9), _det(0, 0, (300, 300, 40, 40), 0.8), _det(0, 0, (100, 100, 60, 60), 0.7)]
    res = coco_bbox_metrics(imgs, anns, dets, [0])
    want = 100.0 * (51 + 50 * 2 / 3) / 101
    assert res["AP"] == pytest.approx(want, rel=1e-9) and res["AP50"] == pytest.approx(want, rel=1e-9)
    assert res["APm"] == pytest.approx(want, rel=1e-9)            # both gt are medium (32^2 .. 96^2); the fp is medium too
    assert np.isnan(res["APs"]) and np.isnan(res["APl"])


def test_iou_threshold_sweep_and_crowd():
    """a detection with IoU 0.6 counts up to the 0.60 threshold only; detections inside a crowd region are neither tp nor fp"""
    from aldi_amd.evaluation import coco_bbox_metrics
    imgs = [dict(id=0)]
    anns = [_ann(0, 0, (0, 0, 100, 100)), _ann(0, 0, (200, 200, 100, 100), iscrowd=1)]
    # IoU of (0,0,100,100) with (0,0,100,60) = 0.6
    dets = [_det(0, 0, (0, 0, 100, 60), 0.9), _det(0, 0, (210, 210, 50, 50), 0.8), _det(0, 0, (220, 220, 50, 50), 0.7)]
    res = coco_bbox_metrics(imgs, anns, dets, [0])
    assert res["AP50"] == pytest.approx(100.0) and res["AP75"] == pytest.approx(0.0)
    assert res["AP"] == pytest.approx(100.0 * 3 / 10)              # thresholds 0.50, 0.55, 0.60 (0.6 >= min(0.6, 1-1e-10))


def test_missing_area_uses_the_reference_formula():
    from aldi_amd.evaluation import maybe_add_optional_annotations
    a = [dict(image_id=0, category_id=0, bbox=[3.0, 7.0, 11.0, 13.0])]
    maybe_add_optional_annotations(a)
    assert a[0]["iscrowd"] == 0 and a[0]["area"] == 7.0 * 11.0     # reference aldi/helpers.py:69-70: bbox[1] * bbox[2]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_scenes_match_loop_oracle(seed):
    from aldi_amd.evaluation import coco_bbox_metrics
    from oracle import coco_eval as oc
    rng = np.random.RandomState(seed)
    imgs = [dict(id=i) for i in range(6)]
    anns, dets = [], []
    for i in range(6):
        for _ in range(rng.randint(0, 6)):
            x, y, w, h = rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(8, 200), rng.uniform(8, 200)
            c = int(rng.randint(0, 3))
            anns.append(_ann(i, c, (x, y, w, h), iscrowd=int(rng.rand() < 0.15)))
            if rng.rand() < 0.8:                                     # a jittered detection of it
                j = rng.uniform(-0.25, 0.25, 4) * np.array([w, h, w, h])
                dets.append(_det(i, c if rng.rand() < 0.9 else int(rng.randint(0, 3)), (x + j[0], y + j[1], max(w + j[2], 2), max(h + j[3], 2)),
                                 round(float(rng.rand()), 2)))        # rounded scores: ties exercise the stable sorts
        for _ in range(rng.randint(0, 4)):                           # clutter
            dets.append(_det(i, int(rng.randint(0, 3)), (rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(5, 150), rng.uniform(5, 150)),
                             round(float(rng.rand()), 2)))
    mine = coco_bbox_metrics(imgs, copy.deepcopy(anns), dets, [0, 1, 2])
    ref = oc.bbox_metrics([im["id"] for im in imgs], copy.deepcopy(anns), dets, [0, 1, 2])
    for k, v in ref.items():
        assert (np.isnan(v) and np.isnan(mine[k])) or mine[k] == pytest.approx(v, abs=1e-9), (k, mine[k], v)


def test_evaluator_adapter_rescales_to_original_size():
    from aldi_amd.evaluation import Detectron2COCOEvaluatorAdapter
    from aldi_amd.structures import Boxes, Instances
    recs = [dict(image_id=7, height=200, width=400, annotations=[dict(bbox=[40, 20, 200, 100], bbox_mode="XYWH_ABS", category_id=1)])]
    ev = Detectron2COCOEvaluatorAdapter("toy_val", recs, num_classes=2, distributed=False)
    inst = Instances((100, 200))                                     # the network saw the image at half size
    inst.pred_boxes = Boxes(torch.tensor([[20.0, 10.0, 120.0, 60.0]]))
    inst.scores = torch.tensor([0.9])
    inst.pred_classes = torch.tensor([1])
    ev.process([dict(image_id=7, height=200, width=400)], [inst])
    res = ev.evaluate()
    assert res["bbox"]["AP50"] == pytest.approx(100.0) and res["bbox"]["AP"] == pytest.approx(100.0)
