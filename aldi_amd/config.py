"""yacs-like config with the reference's key names and defaults.

``add_aldi_config`` mirrors reference aldi/config.py:7-99 key for key (everything
domain-adaptive is off by default); ``get_cfg`` carries the subset of Detectron2 defaults the
hot path reads.  YAML files use the same ``_BASE_`` inheritance and
``KEY.SUBKEY value`` command-line overrides as the reference (tools/train_net.py:49-56).
"""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, List

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    # ---- merging -------------------------------------------------------------------------
    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v, path + k + ".")
            else:
                if isinstance(v, list) and k in self and isinstance(self[k], tuple):
                    v = tuple(v)
                self[k] = v

    @staticmethod
    def load_yaml_with_base(filename: str) -> dict:
        with open(filename) as f:
            raw = f.read()
        cfg = yaml.safe_load(_tuple_safe(raw)) or {}
        if "_BASE_" in cfg:
            base = cfg.pop("_BASE_")
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dicts(merged, cfg)
            return merged
        return cfg

    def merge_from_file(self, filename: str):
        self._merge(CfgNode.load_yaml_with_base(filename))

    def merge_from_list(self, opts: List[Any]):
        assert len(opts) % 2 == 0, "Override list has odd length"
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full_key.split(".")
            for k in keys[:-1]:
                if k not in node:
                    node[k] = CfgNode()
                node = node[k]
            if isinstance(v, str):
                try:
                    v = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            node[keys[-1]] = v


def _tuple_safe(raw: str) -> str:
    """Detectron2 YAMLs write tuples as `(a, b)`; turn them into YAML lists."""
    out = []
    for line in raw.splitlines():
        if ":" in line:
            key, _, val = line.partition(":")
            vs = val.strip()
            if vs.startswith("(") and vs.endswith(")"):
                inner = vs[1:-1].strip().rstrip(",")
                line = f"{key}: [{inner}]"
        out.append(line)
    return "\n".join(out)


def _merge_dicts(a: dict, b: dict):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge_dicts(a[k], v)
        else:
            a[k] = v


CN = CfgNode


def get_cfg() -> CfgNode:
    """Subset of detectron2.config.get_cfg() read by the ALDI hot path."""
    _C = CN()
    _C.VERSION = 2
    _C.MODEL = CN({
        "META_ARCHITECTURE": "GeneralizedRCNN", "DEVICE": "cuda", "WEIGHTS": "", "MASK_ON": False,
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_fpn_backbone", "FREEZE_AT": 2},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res2", "res3", "res4", "res5"], "NORM": "FrozenBN", "STRIDE_IN_1X1": True},
        "FPN": {"IN_FEATURES": ["res2", "res3", "res4", "res5"], "OUT_CHANNELS": 256},
        "ANCHOR_GENERATOR": {"SIZES": [[32], [64], [128], [256], [512]], "ASPECT_RATIOS": [[0.5, 1.0, 2.0]]},
        "RPN": {"IN_FEATURES": ["p2", "p3", "p4", "p5", "p6"], "PRE_NMS_TOPK_TRAIN": 2000, "PRE_NMS_TOPK_TEST": 1000,
                "POST_NMS_TOPK_TRAIN": 1000, "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7, "BATCH_SIZE_PER_IMAGE": 256,
                "POSITIVE_FRACTION": 0.5, "IOU_THRESHOLDS": [0.3, 0.7], "CONV_DIMS": [-1], "BBOX_REG_WEIGHTS": [1.0, 1.0, 1.0, 1.0],
                "SMOOTH_L1_BETA": 0.0, "BBOX_REG_LOSS_TYPE": "smooth_l1", "LOSS_WEIGHT": 1.0, "BBOX_REG_LOSS_WEIGHT": 1.0},
        "ROI_HEADS": {"NAME": "StandardROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["p2", "p3", "p4", "p5"], "BATCH_SIZE_PER_IMAGE": 512,
                      "POSITIVE_FRACTION": 0.25, "IOU_THRESHOLDS": [0.5], "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5,
                      "PROPOSAL_APPEND_GT": True},
        "ROI_BOX_HEAD": {"NAME": "FastRCNNConvFCHead", "NUM_FC": 2, "POOLER_RESOLUTION": 7, "FC_DIM": 1024, "NUM_CONV": 0, "CONV_DIM": 256,
                         "NORM": "", "BBOX_REG_WEIGHTS": [10.0, 10.0, 5.0, 5.0], "SMOOTH_L1_BETA": 0.0, "BBOX_REG_LOSS_TYPE": "smooth_l1",
                         "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "CLS_AGNOSTIC_BBOX_REG": False},
    })
    _C.INPUT = CN({"FORMAT": "BGR", "MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333})
    _C.DATASETS = CN({"TRAIN": (), "TEST": ()})
    _C.DATALOADER = CN({"NUM_WORKERS": 4, "FILTER_EMPTY_ANNOTATIONS": True})
    _C.SOLVER = CN({
        "IMS_PER_BATCH": 16, "BASE_LR": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0001, "GAMMA": 0.1, "STEPS": (30000,),
        "MAX_ITER": 40000, "WARMUP_FACTOR": 1.0 / 1000, "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear",
        "LR_SCHEDULER_NAME": "WarmupMultiStepLR", "CHECKPOINT_PERIOD": 5000, "AMP": {"ENABLED": False},
    })
    _C.TEST = CN({"EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100})
    _C.OUTPUT_DIR = "./output"
    _C.SEED = -1
    _C.VIS_PERIOD = 0
    return _C


def add_aldi_config(cfg: CfgNode):
    """Same keys and defaults as reference aldi/config.py:7-99."""
    _C = cfg
    _C.DATASETS.UNLABELED = tuple()
    _C.DATASETS.BATCH_CONTENTS = ("labeled_weak",)
    _C.DATASETS.BATCH_RATIOS = (1,)

    _C.AUG = CN()
    _C.AUG.WEAK_INCLUDES_MULTISCALE = True
    _C.AUG.LABELED_INCLUDE_RANDOM_ERASING = True
    _C.AUG.UNLABELED_INCLUDE_RANDOM_ERASING = True
    _C.AUG.LABELED_MIC_AUG = False
    _C.AUG.UNLABELED_MIC_AUG = False
    _C.AUG.MIC_RATIO = 0.5
    _C.AUG.MIC_BLOCK_SIZE = 32

    _C.EMA = CN()
    _C.EMA.ENABLED = False
    _C.EMA.ALPHA = 0.9996
    _C.EMA.LOAD_FROM_EMA_ON_START = True
    _C.EMA.START_ITER = 0

    _C.DOMAIN_ADAPT = CN()
    _C.DOMAIN_ADAPT.ALIGN = CN()
    _C.DOMAIN_ADAPT.ALIGN.MIXIN_NAME = "AlignMixin"
    _C.DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED = False
    _C.DOMAIN_ADAPT.ALIGN.IMG_DA_LAYER = "p2"
    _C.DOMAIN_ADAPT.ALIGN.IMG_DA_WEIGHT = 0.01
    _C.DOMAIN_ADAPT.ALIGN.IMG_DA_INPUT_DIM = 256
    _C.DOMAIN_ADAPT.ALIGN.IMG_DA_HIDDEN_DIMS = [256, ]
    _C.DOMAIN_ADAPT.ALIGN.INS_DA_ENABLED = False
    _C.DOMAIN_ADAPT.ALIGN.INS_DA_WEIGHT = 0.01
    _C.DOMAIN_ADAPT.ALIGN.INS_DA_INPUT_DIM = 1024
    _C.DOMAIN_ADAPT.ALIGN.INS_DA_HIDDEN_DIMS = [1024, ]

    _C.DOMAIN_ADAPT.DISTILL = CN()
    _C.DOMAIN_ADAPT.DISTILL.DISTILLER_NAME = "ALDIDistiller"
    _C.DOMAIN_ADAPT.DISTILL.MIXIN_NAME = "DistillMixin"
    _C.DOMAIN_ADAPT.DISTILL.HARD_ROIH_CLS_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.HARD_ROIH_REG_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.HARD_OBJ_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.HARD_RPN_REG_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.ROIH_CLS_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.ROIH_REG_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.OBJ_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.RPN_REG_ENABLED = False
    _C.DOMAIN_ADAPT.DISTILL.CLS_TMP = 1.0
    _C.DOMAIN_ADAPT.DISTILL.OBJ_TMP = 1.0
    _C.DOMAIN_ADAPT.CLS_LOSS_TYPE = "CE"

    _C.DOMAIN_ADAPT.TEACHER = CN()
    _C.DOMAIN_ADAPT.TEACHER.ENABLED = False
    _C.DOMAIN_ADAPT.TEACHER.THRESHOLD = 0.8

    _C.VIT = CN()
    _C.VIT.USE_ACT_CHECKPOINT = True

    _C.SOLVER.IMS_PER_GPU = 2
    _C.SOLVER.BACKWARD_AT_END = True
    _C.SOLVER.OPTIMIZER = "SGD"
    # aldi_amd extensions (not in the reference).  Both are ON by default, so `ALDITrainer(cfg)` built from the reference's own
    # YAML runs the step the benchmark measures; whenever a batch does not fit the fused driver (`_ALDITrainer._can_fuse`: other
    # batch contents, another distiller, a detector without the R-CNN engine) the reference's sequential schedule runs instead.
    # Turn them off with these keys or ALDI_FUSED_STEP=0 / ALDI_STEP_GRAPH=0.
    _C.SOLVER.FUSED_STEP = True           # the step's student passes as one fused launch sequence (numerically the sequential schedule)
    _C.SOLVER.STEP_GRAPH = True           # replay the fused step's two device phases as hipGraphs (aldi_amd/fused_step.py)
    # stem + res2 (frozen) of batch k + 1 inside step k (fused_step: cross-step pipelining; needs the trainer's one-batch look-ahead).  Built, tested
    # (bit-identical prefix, same steps) and OFF: measured 8.12-8.31 ms against 7.97-8.11 in order (profiles/r06_ab_pipeline*.txt) -- the student's
    # prefix already overlaps the teacher's pass at the head of phase A (removing it there saves 0.06 ms), anywhere else it costs its full 0.3-0.4 ms
    _C.SOLVER.PIPELINE_PREFIX = False
    _C.SOLVER.GRAD_PAYLOAD = "fp32"       # data-parallel gradient exchange: "fp32" (exact) or "bf16" (half the bytes per xGMI link; sums in bf16)
    _C.SOLVER.GRAD_EXCHANGE = "auto"        # per bucket: "all_reduce" or "rs_ag" (reduce_scatter_tensor + all_gather_into_tensor, aldi_amd/reduce.py);
                                            # "auto" = rs_ag on the 8 ranks of one fully connected xGMI node, all_reduce otherwise

    # Deformable-DETR (the reference's absent submodule adds these through its own add_deformable_detr_config; configs/Base-DETR.yaml)
    _C.MODEL.DEFORMABLE_DETR = CN()
    D = _C.MODEL.DEFORMABLE_DETR
    D.NUM_CLASSES = 80
    D.BACKBONE = "resnet50"
    D.DILATION = False
    D.POSITION_EMBEDDING = "sine"
    D.POSITION_EMBEDDING_SCALE = 6.283185307179586
    D.NUM_FEATURE_LEVELS = 4
    D.WITH_BOX_REFINE = False
    D.TWO_STAGE = False
    D.FROZEN_WEIGHTS = False
    D.TRANSFORMER = CN()
    D.TRANSFORMER.NUM_QUERIES = 300
    D.TRANSFORMER.ENC_LAYERS = 6
    D.TRANSFORMER.DEC_LAYERS = 6
    D.TRANSFORMER.NHEADS = 8
    D.TRANSFORMER.DIM_FEEDFORWARD = 1024
    D.TRANSFORMER.HIDDEN_DIM = 256
    D.TRANSFORMER.DROPOUT = 0.1
    D.TRANSFORMER.DEC_N_POINTS = 4
    D.TRANSFORMER.ENC_N_POINTS = 4
    D.LOSS = CN()
    D.LOSS.AUX_LOSS = True
    D.LOSS.MASK_LOSS_COEF = 1.0
    D.LOSS.DICE_LOSS_COEF = 1.0
    D.LOSS.CLS_LOSS_COEF = 2.0
    D.LOSS.BBOX_LOSS_COEF = 5.0
    D.LOSS.GIOU_LOSS_COEF = 2.0
    D.LOSS.FOCAL_ALPHA = 0.25
    D.MATCHER = CN()
    D.MATCHER.SET_COST_CLASS = 2
    D.MATCHER.SET_COST_BBOX = 5
    D.MATCHER.SET_COST_GIOU = 2
    if "CLIP_GRADIENTS" not in _C.SOLVER:                     # detectron2 defaults
        _C.SOLVER.CLIP_GRADIENTS = CN({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0})
    if "CROP" not in _C.INPUT:
        _C.INPUT.CROP = CN({"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]})
    _C.SOLVER.BACKBONE_LR_MULTIPLIER = 0.1
    _C.SOLVER.LR_BACKBONE_NAMES = ["backbone.0"]
    _C.SOLVER.LR_LINEAR_PROJ_NAMES = ["reference_points", "sampling_offsets"]
    _C.SOLVER.LR_LINEAR_PROJ_MULTIPLIER = 0.1

    _C.MODEL.CONVNEXT = CN()
    _C.MODEL.CONVNEXT.DEPTHS = [3, 3, 9, 3]
    _C.MODEL.CONVNEXT.DIMS = [96, 192, 384, 768]
    _C.MODEL.CONVNEXT.DROP_PATH_RATE = 0.2
    _C.MODEL.CONVNEXT.LAYER_SCALE_INIT_VALUE = 1e-6
    _C.MODEL.CONVNEXT.OUT_FEATURES = [0, 1, 2, 3]
    _C.SOLVER.WEIGHT_DECAY_RATE = 0.95
