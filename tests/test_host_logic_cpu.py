"""Host-side logic that mirrors the reference's plugin API, against the golden traces (CPU only)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch


def test_config_keys_and_defaults_match_reference():
    from aldi_amd.config import add_aldi_config, get_cfg
    cfg = get_cfg()
    add_aldi_config(cfg)
    assert cfg.DATASETS.BATCH_CONTENTS == ("labeled_weak",) and cfg.DATASETS.BATCH_RATIOS == (1,)
    assert cfg.EMA.ENABLED is False and cfg.EMA.ALPHA == 0.9996 and cfg.EMA.START_ITER == 0 and cfg.EMA.LOAD_FROM_EMA_ON_START is True
    A = cfg.DOMAIN_ADAPT.ALIGN
    assert (A.MIXIN_NAME, A.IMG_DA_ENABLED, A.IMG_DA_LAYER, A.IMG_DA_WEIGHT, A.IMG_DA_INPUT_DIM, A.IMG_DA_HIDDEN_DIMS) == ("AlignMixin", False, "p2", 0.01, 256, [256])
    assert (A.INS_DA_ENABLED, A.INS_DA_WEIGHT, A.INS_DA_INPUT_DIM, A.INS_DA_HIDDEN_DIMS) == (False, 0.01, 1024, [1024])
    D = cfg.DOMAIN_ADAPT.DISTILL
    assert D.DISTILLER_NAME == "ALDIDistiller" and D.MIXIN_NAME == "DistillMixin" and D.CLS_TMP == 1.0 and D.OBJ_TMP == 1.0
    for k in ("HARD_ROIH_CLS_ENABLED", "HARD_ROIH_REG_ENABLED", "HARD_OBJ_ENABLED", "HARD_RPN_REG_ENABLED", "ROIH_CLS_ENABLED",
              "ROIH_REG_ENABLED", "OBJ_ENABLED", "RPN_REG_ENABLED"):
        assert D[k] is False
    assert cfg.DOMAIN_ADAPT.CLS_LOSS_TYPE == "CE" and cfg.DOMAIN_ADAPT.TEACHER.THRESHOLD == 0.8
    assert cfg.SOLVER.IMS_PER_GPU == 2 and cfg.SOLVER.BACKWARD_AT_END is True and cfg.SOLVER.OPTIMIZER == "SGD"


def test_yaml_base_inheritance_and_overrides():
    from aldi_amd.config import add_aldi_config, get_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(root, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 8 and cfg.SOLVER.IMS_PER_GPU == 2 and cfg.SOLVER.BASE_LR == 0.06
    assert cfg.DATASETS.BATCH_CONTENTS == ("labeled_strong", "unlabeled_strong") and cfg.DATASETS.BATCH_RATIOS == (1, 1)
    assert cfg.EMA.ENABLED and cfg.DOMAIN_ADAPT.DISTILL.OBJ_ENABLED and not cfg.DOMAIN_ADAPT.DISTILL.HARD_OBJ_ENABLED
    assert cfg.SOLVER.BACKWARD_AT_END is False and cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN == 2000
    cfg.merge_from_list(["DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED", "True", "SOLVER.BASE_LR", "0.01"])
    assert cfg.DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED is True and cfg.SOLVER.BASE_LR == 0.01
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 1.0


def test_fast_path_is_the_default_and_the_reference_yaml_selects_it():
    """SOLVER.FUSED_STEP / STEP_GRAPH are on unless switched off; (in the build container, where /root/reference exists) the reference's own
    ALDI-Best-Cityscapes.yaml loads unmodified, and into the same model / solver / adaptation / EMA configuration as the repository's copy"""
    from aldi_amd.config import add_aldi_config, get_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(path):
        cfg = get_cfg()
        add_aldi_config(cfg)
        cfg.merge_from_file(path)
        return cfg

    def flat(node, pre=""):
        out = {}
        for k, v in node.items():
            if hasattr(v, "items"):
                out.update(flat(v, pre + k + "."))
            else:
                out[pre + k] = v
        return out
    mine = load(os.path.join(root, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    assert mine.SOLVER.FUSED_STEP is True and mine.SOLVER.STEP_GRAPH is True
    ref_path = "/root/reference/configs/cityscapes/ALDI-Best-Cityscapes.yaml"
    if not os.path.exists(ref_path):
        return
    a, b = flat(mine), flat(load(ref_path))
    # everything that configures the STEP is identical; what differs names data that does not exist here (dataset names, input sizes of
    # the host-side resize, augmentation switches, evaluation / checkpoint periods, output directory, the initial checkpoint file)
    step_keys = [k for k in b if k.split(".")[0] in ("MODEL", "SOLVER", "DOMAIN_ADAPT", "EMA", "VIT") and not k.startswith("MODEL.ROI_MASK_HEAD")]
    assert len(step_keys) > 100
    differ = sorted(k for k in step_keys if a.get(k, "<missing>") != b[k])
    assert differ == ["MODEL.WEIGHTS", "SOLVER.CHECKPOINT_PERIOD"], differ


def test_step_driver_matches_reference_trace(golden_dir):
    """aldi_amd.trainer.run_model_labeled_unlabeled vs the call/loss trace recorded from reference aldi/trainer.py:28-117."""
    from aldi_amd.dataloader import unpack_data_weak_strong
    from aldi_amd.trainer import run_model_labeled_unlabeled
    traces = json.load(open(os.path.join(golden_dir, "g6_step_trace.json")))
    for name, tr in traces.items():
        log = []

        class Model:
            def __init__(self):
                self.img_align = object() if tr["align"] else None
                self.ins_align = object() if tr["align"] else None
                self.n = 0

            def __call__(self, data, **kw):
                self.n += 1
                log.append(["model", [d["id"] for d in data], dict(sorted(kw.items()))])
                base = float(self.n)
                out = {"loss_cls": torch.tensor(base, requires_grad=True) * 1.0, "loss_rpn_cls": torch.tensor(base + 0.25, requires_grad=True) * 1.0}
                if kw.get("do_align"):
                    out["loss_da_img"] = torch.tensor(base + 0.5, requires_grad=True) * 1.0
                elif self.img_align is not None:
                    out["_da"] = torch.tensor(0.0, requires_grad=True) * 1.0
                return out

        class Dist:
            def distill_enabled(self):
                return tr["distill"]

            def __call__(self, t, s):
                log.append(["distiller", [d["id"] for d in t], [d["id"] for d in s]])
                return {"loss_cls": torch.tensor(7.0, requires_grad=True) * 0.0, "loss_cls_ce": torch.tensor(3.0, requires_grad=True) * 1.0}

        class T:
            pass
        t = T()
        t.model, t.distiller, t.backward_at_end, t.model_batch_size = Model(), Dist(), tr["backward_at_end"], tr["ims_per_gpu"]
        t.do_backward = lambda losses, override=False: log.append(["backward", round(float(losses), 6)])
        n = tr["nper"]
        lab = [{"id": f"L{i}", "img_weak": f"w{i}"} for i in range(n)]
        unl = [{"id": f"U{i}", "img_weak": f"w{i}"} for i in range(n)]
        has_unl = any(c.startswith("unlabeled") for c in tr["contents"])
        data = unpack_data_weak_strong(lab, unl if has_unl else None, batch_contents=tuple(tr["contents"]))
        ld = run_model_labeled_unlabeled(t, *data)
        assert log == tr["log"], name
        assert {k: round(float(v), 6) for k, v in ld.items()} == tr["loss_dict"], name
        assert list(ld.keys()) == list(tr["loss_dict"].keys()) or set(ld) == set(tr["loss_dict"])
        assert {k: bool(getattr(v, "requires_grad", False)) for k, v in ld.items()} == tr["requires_grad"], name


def test_unpack_data_weak_strong_matches_reference(golden_dir):
    from aldi_amd.dataloader import unpack_data_weak_strong
    res = json.load(open(os.path.join(golden_dir, "g7_unpack.json")))
    lab = [{"image": "Ls0", "img_weak": "Lw0"}, {"image": "Ls1", "img_weak": "Lw1"}]
    unl = [{"image": "Us0", "img_weak": "Uw0"}]
    for key, exp in res.items():
        if key == "labeled_none":
            out = unpack_data_weak_strong(None, unl, batch_contents=("labeled_strong", "unlabeled_strong"))
        else:
            out = unpack_data_weak_strong(lab, unl, batch_contents=tuple(key.split("|")))
        assert [None if o is None else [d["image"] for d in o] for o in out] == exp, key
    assert lab[0]["image"] == "Ls0"      # the strong dicts are not modified (weak = deepcopy)


def test_process_bbox_matches_reference(golden_dir):
    from aldi_amd.pseudolabeler import process_bbox
    from aldi_amd.structures import Boxes, Instances
    g = np.load(os.path.join(golden_dir, "g4_process_bbox.npz"))
    for tag in ("thr08", "empty"):
        inst = Instances((100, 200))
        inst.scores = torch.from_numpy(g[f"{tag}_in_scores"])
        inst.pred_boxes = Boxes(torch.from_numpy(g[f"{tag}_in_boxes"]))
        inst.pred_classes = torch.from_numpy(g[f"{tag}_in_classes"])
        r = process_bbox(inst, thres=0.8)
        assert np.array_equal(r.gt_boxes.tensor.numpy(), g[f"{tag}_gt_boxes"])
        assert np.array_equal(r.gt_classes.numpy(), g[f"{tag}_gt_classes"])
        assert np.array_equal(r.scores.numpy(), g[f"{tag}_scores"])


def test_hard_loss_mask_keys(golden_dir):
    """ALDIDistiller.__call__ loss key order / masking (reference aldi/distill.py:170-191), host part only."""
    from aldi_amd.distill import ALDIDistiller, Distiller
    res = json.load(open(os.path.join(golden_dir, "g8_hard_mask.json")))
    for key in ("0000", "1010", "1111"):
        f = [c == "1" for c in key]
        d = object.__new__(ALDIDistiller)
        d.do_hard_cls, d.do_hard_obj, d.do_hard_rpn_reg, d.do_hard_roi_reg = f
        d.do_cls_dst = d.do_obj_dst = d.do_rpn_reg_dst = d.do_roih_reg_dst = True
        d._distill_forward = lambda t, s: OrderedDict(loss_cls=torch.tensor(1.5), loss_box_reg=torch.tensor(2.5), loss_rpn_cls=torch.tensor(3.5),
                                                      loss_rpn_loc=torch.tensor(4.5))
        d._soft = OrderedDict(loss_obj_bce=torch.tensor(0.1), loss_rpn_l1=torch.tensor(0.2), loss_cls_ce=torch.tensor(0.3), loss_roih_l1=torch.tensor(0.4))
        L = d([], [])
        assert list(L.keys()) == res[key]["keys"]
        assert {k: round(float(v), 6) for k, v in L.items()} == res[key]["values"]
        assert d.distill_enabled() == res[key]["enabled"]
    base = Distiller(None, None)
    assert base.distill_enabled() is False and base([], []) == {}


def test_param_layout_roundtrip_and_d2_key_names():
    from aldi_amd import synthetic as syn
    from aldi_amd.arch import ParamLayout
    lay = ParamLayout(8, img_da=True, ins_da=True)
    sd = syn.init_state_dict(8, seed=3, img_da=True, ins_da=True)
    keys = lay.state_dict_keys()
    assert keys[:6] == ["backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.stem.conv1.norm.weight", "backbone.bottom_up.stem.conv1.norm.bias",
                        "backbone.bottom_up.stem.conv1.norm.running_mean", "backbone.bottom_up.stem.conv1.norm.running_var",
                        "backbone.bottom_up.res2.0.shortcut.weight"]
    assert "roi_heads.box_predictor.bbox_pred.bias" in keys and "img_align.model.4.weight" in keys and "ins_align.model.3.bias" in keys
    assert set(keys) == set(sd.keys())
    flat = lay.pack(sd)
    back = lay.unpack(flat)
    for k in keys:
        assert torch.equal(back[k], sd[k]), k
    n_train = sum(v.numel() for k, v in sd.items() if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2"))
    assert n_train == 41_108_536 + 590_337 + 1_050_625      # R50-FPN trainable (K=8) + the two discriminators (golden g1 counts)
    assert lay.n_train >= n_train


def test_lr_schedule_is_detectron2_warmup_multistep():
    from aldi_amd.trainer import WarmupMultiStepLR

    class Opt:
        param_groups = [{"lr": 0.0}]
    s = WarmupMultiStepLR(Opt(), 0.06, (1600,), 0.1, 0.01, 100)
    assert abs(s.lr_at(0) - 0.06 * 0.01) < 1e-12
    assert abs(s.lr_at(50) - 0.06 * (0.01 * 0.5 + 0.5)) < 1e-12
    assert abs(s.lr_at(100) - 0.06) < 1e-12 and abs(s.lr_at(1600) - 0.006) < 1e-12


def test_registries_expose_reference_names():
    from aldi_amd.align import ALIGN_MIXIN_REGISTRY
    from aldi_amd.distill import DISTILL_MIXIN_REGISTRY, DISTILLER_REGISTRY
    from aldi_amd.model import META_ARCH_REGISTRY
    from aldi_amd import trainer
    assert ALIGN_MIXIN_REGISTRY.get("AlignMixin") and DISTILL_MIXIN_REGISTRY.get("DistillMixin") and META_ARCH_REGISTRY.get("GeneralizedRCNN")
    for n in ("Distiller", "HardDistiller", "ALDIDistiller"):
        assert DISTILLER_REGISTRY.get(n)
    assert trainer.Trainer is trainer.ALDITrainer
    with pytest.raises(KeyError):
        DISTILLER_REGISTRY.get("nope")


def test_vit_layerwise_lr_decay_groups():
    """ViTDet-B AdamW param groups: detectron2 get_vit_lr_decay_rate(num_layers=12, lr_decay_rate=0.7) as the
    reference enables it for build_vitdet_b_backbone (aldi/backbone.py:73-79, aldi/trainer.py:204)."""
    from aldi_amd.vit import VitConfig, VitParams
    cfg = VitConfig(sfp=True, num_classes=8)
    P = VitParams(cfg, "cpu")
    pre = cfg.prefix
    assert abs(P.lr_factor(pre + "blocks.0.attn.qkv.weight", 0.7, 12) - 0.7 ** 12) < 1e-15
    assert abs(P.lr_factor(pre + "blocks.11.mlp.fc2.bias", 0.7, 12) - 0.7) < 1e-15
    assert abs(P.lr_factor(pre + "pos_embed", 0.7, 12) - 0.7 ** 13) < 1e-15
    assert abs(P.lr_factor(pre + "patch_embed.proj.weight", 0.7, 12) - 0.7 ** 13) < 1e-15
    assert P.lr_factor("backbone.simfp_2.0.weight", 0.7, 12) == 1.0 and P.lr_factor("roi_heads.box_predictor.cls_score.weight", 0.7, 12) == 1.0
    assert len(P.lr_groups(None, 12)) == 2                                  # decay off: [decayed | norms + pos_embed]
    g = P.lr_groups(0.7, 12)
    assert g[0][0] == 0 and g[-1][1] == P.n and all(a[1] == b[0] for a, b in zip(g[:-1], g[1:]))   # a partition of the flat state
    assert len(g) <= 30
    for name in P.spec:
        lo = P.off[name]
        (piece,) = [x for x in g if x[0] <= lo < x[1]]
        assert piece[2] == (lo < P.n_decay) and abs(piece[3] - P.lr_factor(name, 0.7, 12)) < 1e-15, name


def test_d2params_from_cfg_reads_and_validates():
    """engine.D2Params.from_cfg: the Detectron2 keys of the R50 engine come from the config; what the kernels do not implement raises"""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.engine import D2Params
    cfg = get_cfg()
    add_aldi_config(cfg)
    assert D2Params.from_cfg(cfg) == D2Params()                    # detectron2's defaults are the engine's defaults
    cfg.merge_from_list(["MODEL.PIXEL_MEAN", [1.0, 2.0, 3.0], "MODEL.PIXEL_STD", [4.0, 5.0, 6.0], "MODEL.RPN.NMS_THRESH", 0.6,
                         "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 1500, "MODEL.ROI_HEADS.POSITIVE_FRACTION", 0.5, "MODEL.RPN.IOU_THRESHOLDS", [0.2, 0.8],
                         "MODEL.ROI_HEADS.IOU_THRESHOLDS", [0.6], "MODEL.ROI_HEADS.NMS_THRESH_TEST", 0.4, "TEST.DETECTIONS_PER_IMAGE", 50,
                         "MODEL.ANCHOR_GENERATOR.SIZES", [[16], [32], [64], [128], [256]], "MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS", [5.0, 5.0, 2.0, 2.0]])
    p = D2Params.from_cfg(cfg)
    assert p.pixel_mean == (1.0, 2.0, 3.0) and p.pixel_std == (4.0, 5.0, 6.0) and p.rpn_nms == 0.6 and p.rpn_pre == (1500, 1000)
    assert p.roi_pos_frac == 0.5 and p.rpn_iou == (0.2, 0.8) and p.roi_iou == 0.6 and p.nms_test == 0.4 and p.dets == 50
    assert p.anchor_sizes == (16.0, 32.0, 64.0, 128.0, 256.0) and p.roi_weights == (5.0, 5.0, 2.0, 2.0)
    import pytest
    for bad in (["MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS", [[1.0]]], ["MODEL.RPN.PRE_NMS_TOPK_TRAIN", 4000], ["TEST.DETECTIONS_PER_IMAGE", 500],
                ["MODEL.RPN.SMOOTH_L1_BETA", 0.1], ["MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 14], ["MODEL.RPN.BBOX_REG_WEIGHTS", [2.0, 2.0, 1.0, 1.0]]):
        c2 = get_cfg()
        add_aldi_config(c2)
        c2.merge_from_list(bad)
        with pytest.raises(ValueError):
            D2Params.from_cfg(c2)


def test_build_train_loader_batch_shares_follow_the_reference_loop():
    """aldi/trainer.py:211-222: one batch size per entry of BATCH_CONTENTS (a LIST: a content named twice counts twice), the sizes must add up
    to IMS_PER_BATCH, and a domain's loader serves the largest of its parts"""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    from aldi_amd.reduce import resolve_exchange

    def loaders(contents, ratios, total):
        cfg = get_cfg()
        add_aldi_config(cfg)
        cfg.merge_from_list(["DATASETS.BATCH_CONTENTS", contents, "DATASETS.BATCH_RATIOS", ratios, "SOLVER.IMS_PER_BATCH", total,
                             "SYNTHETIC.HEIGHT", 64, "SYNTHETIC.WIDTH", 96])
        dl = ALDITrainer.build_train_loader(cfg)
        return dl.labeled_loader, dl.unlabeled_loader

    lab, unl = loaders(("labeled_strong", "labeled_strong"), (1, 1), 4)            # duplicates: 2 + 2 = 4, the labeled loader serves 2
    assert lab is not None and unl is None and lab.bs == 2
    lab, unl = loaders(("labeled_weak", "labeled_strong", "unlabeled_strong"), (1, 1, 2), 8)
    assert lab.bs == 2 and unl.bs == 4
    with pytest.raises(AssertionError):
        loaders(("labeled_strong", "unlabeled_strong"), (1, 2), 4)                 # int(4/3) + int(8/3) = 3 != 4
    # SOLVER.GRAD_EXCHANGE "auto" outside a process group is the plain all-reduce; explicit names pass through
    assert resolve_exchange("auto") == "all_reduce" and resolve_exchange("rs_ag") == "rs_ag" and resolve_exchange("all_reduce") == "all_reduce"
