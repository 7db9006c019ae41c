#!/usr/bin/env python
"""Generate golden input/output vectors from the REFERENCE's own ALDI code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (justinkay/aldi) imports ``detectron2`` / ``fvcore`` at module top
level; neither is installed nor installable here.  This script fabricates those
module *names* at import time (a ``sys.meta_path`` finder) with minimal real
behaviour for the handful of helpers the ALDI-owned arithmetic touches
(``Registry``, ``configurable``, ``Boxes``, ``Instances``, ``cat``,
``cross_entropy``, ``smooth_l1_loss``, ``comm.get_world_size``), imports the
reference's ``aldi.*`` modules from /root/reference, runs the ALDI-owned
functions on seeded inputs and stores inputs + outputs as ``.npz`` fixtures next
to this file.  Only the fixtures travel; no reference source is copied.

CAVEAT recorded with every fixture: ``cat`` (= ``torch.cat``), ``cross_entropy``
(= ``F.cross_entropy`` with empty guard) and ``smooth_l1_loss`` (beta=0 -> |x-y|,
mean of empty = 0*sum) are stubbed from the published Detectron2/fvcore
behaviour -- those three third-party helpers are themselves unpinned.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import random
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ----------------------------------------------------------------------------
# stubs for the absent third-party names
# ----------------------------------------------------------------------------
class Registry:
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


def configurable(init_func=None, *, from_config=None):
    if init_func is not None:
        return init_func
    return lambda f: f


class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self._fields[k] = v

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, k, v):
        if k.startswith("_"):
            object.__setattr__(self, k, v)
        else:
            self._fields[k] = v

    def __getattr__(self, k):
        if k == "_fields" or k not in self._fields:
            raise AttributeError(k)
        return self._fields[k]

    def to(self, device):
        return self

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0


def _cat(tensors, dim=0):
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def _cross_entropy(input, target, *, reduction="mean", **kw):
    if target.numel() == 0 and reduction == "mean":
        return input.sum() * 0.0
    return F.cross_entropy(input, target, reduction=reduction, **kw)


def _smooth_l1_loss(input, target, beta, reduction="none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


class _AnyBase(torch.nn.Module):
    """stands in for GeneralizedRCNN & friends (never instantiated here)."""


class _Comm(types.ModuleType):
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def is_main_process():
        return True


class _Transform:
    """fvcore.transforms.transform.Transform: only what aldi/aug.py's transforms touch (constructor + _set_attributes)"""
    def __init__(self):
        pass

    def _set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self" and not k.startswith("_"):
                    setattr(self, k, v)


class _NoOpTransform(_Transform):
    def apply_image(self, img):
        return img


def _cv2_resize(src, dsize, interpolation=0):
    """cv2.resize(..., INTER_NEAREST) restated (cv2 is absent): x_ofs = min(floor(x * src_w / dst_w), src_w - 1)"""
    W, H = dsize
    sh, sw = src.shape[:2]
    ys = np.minimum(np.floor(np.arange(H) * (sh / H)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(W) * (sw / W)).astype(np.int64), sw - 1)
    return src[ys][:, xs]


REAL = {
    "fvcore.transforms.transform": {"Transform": _Transform, "NoOpTransform": _NoOpTransform},
    "cv2": {"resize": _cv2_resize, "INTER_NEAREST": 0},
    "detectron2.utils.registry": {"Registry": Registry},
    "detectron2.config": {"configurable": configurable, "CfgNode": dict},
    "detectron2.structures": {"Boxes": Boxes, "Instances": Instances},
    "detectron2.structures.boxes": {"Boxes": Boxes},
    "detectron2.structures.instances": {"Instances": Instances},
    "detectron2.layers": {"cat": _cat},
    "detectron2.layers.wrappers": {"cross_entropy": _cross_entropy},
    "detectron2.modeling": {"GeneralizedRCNN": _AnyBase},
    "detectron2.modeling.meta_arch.rcnn": {"GeneralizedRCNN": _AnyBase},
    "detectron2.modeling.meta_arch.build": {"META_ARCH_REGISTRY": Registry("META_ARCH")},
    "fvcore.nn": {"smooth_l1_loss": _smooth_l1_loss},
    "detectron2.modeling.backbone": {"Backbone": torch.nn.Module},        # aldi/backbone.py ConvNeXt(Backbone) is a plain nn.Module
}


class _Fake(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.endswith("REGISTRY"):
            obj = Registry(name)
        else:
            obj = type(name, (), {})      # any other imported name becomes an empty class
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("detectron2", "fvcore", "cv2", "scipy")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS and fullname.split(".")[0] != "scipy":
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name == "detectron2.utils.comm":
            m = _Comm(spec.name)
        else:
            m = _Fake(spec.name)
        m.__path__ = []
        for k, v in REAL.get(spec.name, {}).items():
            setattr(m, k, v)
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    mods = {}
    for n in ("helpers", "align", "ema", "pseudolabeler", "distill", "dataloader", "trainer", "aug", "backbone"):
        mods[n] = importlib.import_module("aldi." + n)
    return mods


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


# ----------------------------------------------------------------------------
# G1  discriminators + gradient reversal + domain BCE  (aldi/align.py:76-135, aldi/helpers.py:51-63)
# ----------------------------------------------------------------------------
def g1(m):
    out = {}
    torch.manual_seed(11)
    # small instances keep the fixture small; the default-size parameter counts are pinned separately below
    conv = m["align"].ConvDiscriminator(32, hidden_dims=[32])
    fc = m["align"].FCDiscriminator(64, hidden_dims=[64])
    for tag, net, x in (("conv", conv, torch.randn(2, 32, 9, 11)), ("fc", fc, torch.randn(16, 64))):
        for labeled in (1, 0):
            xi = x.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            feats = m["helpers"].grad_reverse(xi)
            preds = net(feats)
            loss = 0.01 * F.binary_cross_entropy_with_logits(
                preds, torch.FloatTensor(preds.data.size()).fill_(labeled))
            loss.backward()
            out[f"{tag}_x"] = x
            out[f"{tag}_preds"] = preds
            out[f"{tag}_loss_l{labeled}"] = loss
            out[f"{tag}_dx_l{labeled}"] = xi.grad
            for k, p in net.named_parameters():
                out[f"{tag}_grad_l{labeled}.{k}"] = p.grad
        for k, v in net.state_dict().items():
            out[f"{tag}_sd.{k}"] = v
    out["conv_nparams_default"] = np.int64(sum(p.numel() for p in m["align"].ConvDiscriminator(256, hidden_dims=[256]).parameters()))
    out["fc_nparams_default"] = np.int64(sum(p.numel() for p in m["align"].FCDiscriminator(1024, hidden_dims=[1024]).parameters()))
    out["conv_keys"] = np.array(list(conv.state_dict().keys()))
    out["fc_keys"] = np.array(list(fc.state_dict().keys()))
    save("g1_discriminators", **out)


class _IO:
    def __init__(self, output=None):
        self.output = output


def _bare_distiller(m, **attrs):
    d = object.__new__(m["distill"].ALDIDistiller)      # skip __init__ (hooks need a real D2 model)
    base = dict(do_hard_cls=False, do_hard_obj=False, do_hard_rpn_reg=False, do_hard_roi_reg=False,
                do_cls_dst=True, do_obj_dst=True, do_rpn_reg_dst=True, do_roih_reg_dst=True,
                cls_temperature=1.0, obj_temperature=1.0, cls_loss_type="CE", pseudo_label_threshold=0.8)
    base.update(attrs)
    for k, v in base.items():
        setattr(d, k, v)
    return d


# ----------------------------------------------------------------------------
# G2  get_roih_losses  (aldi/distill.py:231-278)
# ----------------------------------------------------------------------------
def g2(m):
    out = {}
    gen = torch.Generator().manual_seed(22)
    cases = [("r256k8", 256, 8, False), ("r64k80", 64, 80, False), ("nofg", 32, 8, True)]
    for tag, R, K, nofg in cases:
        sl = torch.randn(R, K + 1, generator=gen) * 2
        tl = torch.randn(R, K + 1, generator=gen) * 2
        if nofg:
            tl[:, K] = 50.0                       # teacher argmax is background everywhere
        sdl = torch.randn(R, 4 * K, generator=gen)
        tdl = torch.randn(R, 4 * K, generator=gen)
        out.update({f"{tag}_s_logits": sl, f"{tag}_t_logits": tl, f"{tag}_s_deltas": sdl, f"{tag}_t_deltas": tdl})
        for lt in ("CE", "KL"):
            for T in (1.0, 0.5):
                d = _bare_distiller(m, cls_loss_type=lt, cls_temperature=T)
                s1 = sl.clone().requires_grad_(True)
                s2 = sdl.clone().requires_grad_(True)
                d.student_boxpred_io = _IO((s1, s2))
                d.teacher_boxpred_io = _IO((tl, tdl))
                L = d.get_roih_losses()
                tot = L["loss_cls_ce"] + L["loss_roih_l1"]
                tot.backward()
                key = f"{tag}_{lt}_T{T}"
                out[key + "_loss_cls_ce"] = L["loss_cls_ce"]
                out[key + "_loss_roih_l1"] = L["loss_roih_l1"]
                out[key + "_dlogits"] = s1.grad
                out[key + "_ddeltas"] = s2.grad
    save("g2_roih_losses", **out)


# ----------------------------------------------------------------------------
# G3  get_rpn_losses incl. the index-order quirk  (aldi/distill.py:193-229)
# ----------------------------------------------------------------------------
def g3(m):
    out = {}
    gen = torch.Generator().manual_seed(33)
    N, A = 2, 3
    shapes = [(8, 12), (4, 6), (2, 3), (1, 2), (1, 1)]
    sumA = sum(h * w * A for h, w in shapes)
    out["shapes"] = np.array(shapes, dtype=np.int64)

    def heads():
        return ([torch.randn(N, A, h, w, generator=gen) for h, w in shapes],
                [torch.randn(N, 4 * A, h, w, generator=gen) for h, w in shapes])
    for tag, zero_fg in (("mix", False), ("zerofg", True)):
        s_lo, s_de = heads()
        t_lo, t_de = heads()
        labels = torch.full((N, sumA), -1, dtype=torch.int8)
        perm = torch.randperm(N * sumA, generator=gen)
        flat = labels.view(-1)
        flat[perm[:40]] = 0
        if not zero_fg:
            flat[perm[40:60]] = 1
        fixed = [labels[i].clone() for i in range(N)]

        class _RPN:
            def label_and_sample_anchors(self, anchors, gt_instances):
                return [x.clone() for x in fixed], None

        class _Teacher:
            device = torch.device("cpu")
            proposal_generator = _RPN()
        for T in (1.0, 0.5):
            d = _bare_distiller(m, obj_temperature=T)
            d.teacher = _Teacher()
            s_lo_g = [t.clone().requires_grad_(True) for t in s_lo]
            s_de_g = [t.clone().requires_grad_(True) for t in s_de]
            d.student_rpn_head_io = _IO((s_lo_g, s_de_g))
            d.teacher_rpn_head_io = _IO((t_lo, t_de))
            d.teacher_anchor_io = _IO(None)
            insts = [{"instances": Instances((1, 1))} for _ in range(N)]
            L = d.get_rpn_losses(insts)
            (L["loss_obj_bce"] + L["loss_rpn_l1"]).backward()
            key = f"{tag}_T{T}"
            out[key + "_loss_obj_bce"] = L["loss_obj_bce"]
            out[key + "_loss_rpn_l1"] = L["loss_rpn_l1"]
            for l in range(len(shapes)):
                out[f"{key}_dlogits{l}"] = s_lo_g[l].grad if s_lo_g[l].grad is not None else torch.zeros_like(s_lo[l])
                out[f"{key}_ddeltas{l}"] = s_de_g[l].grad if s_de_g[l].grad is not None else torch.zeros_like(s_de[l])
        out[f"{tag}_labels"] = labels
        for l in range(len(shapes)):
            out[f"{tag}_s_logits{l}"], out[f"{tag}_s_deltas{l}"] = s_lo[l], s_de[l]
            out[f"{tag}_t_logits{l}"], out[f"{tag}_t_deltas{l}"] = t_lo[l], t_de[l]
    save("g3_rpn_losses", **out)


# ----------------------------------------------------------------------------
# G4  process_bbox threshold filter  (aldi/pseudolabeler.py:51-67)
# ----------------------------------------------------------------------------
def g4(m):
    out = {}
    scores = torch.tensor([0.99, 0.81, 0.8, 0.8000001, 0.79, 0.5, 0.05], dtype=torch.float32)
    boxes = torch.arange(28, dtype=torch.float32).reshape(7, 4)
    classes = torch.tensor([3, 1, 4, 1, 5, 2, 6])
    for tag, thr, sc in (("thr08", 0.8, scores), ("empty", 0.8, scores * 0.5)):
        inst = Instances((100, 200))
        inst.scores = sc
        inst.pred_boxes = Boxes(boxes)
        inst.pred_classes = classes
        r = m["pseudolabeler"].process_bbox(inst, thres=thr)
        out[f"{tag}_in_scores"], out[f"{tag}_in_boxes"], out[f"{tag}_in_classes"] = sc, boxes, classes
        out[f"{tag}_gt_boxes"] = r.gt_boxes.tensor
        out[f"{tag}_gt_classes"] = r.gt_classes
        out[f"{tag}_scores"] = r.scores
        out[f"{tag}_image_size"] = np.array(r.image_size)
    save("g4_process_bbox", **out)


# ----------------------------------------------------------------------------
# G5  EMA  (aldi/ema.py:8-57)
# ----------------------------------------------------------------------------
def g5(m):
    out = {}

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 4, 3)
            self.query_embed = torch.nn.Embedding(5, 6)
            self.register_buffer("running_var", torch.ones(4))

        @property
        def device(self):
            return self.conv.weight.device
    torch.manual_seed(55)
    teacher_src = Tiny()
    ema = m["ema"].EMA(teacher_src, alpha=0.9996, start_iter=2)
    student = Tiny()
    with torch.no_grad():
        student.running_var.add_(torch.rand(4))
    for k, v in ema.model.state_dict().items():
        out["t0." + k] = v.clone()
    for k, v in student.state_dict().items():
        out["s." + k] = v.clone()
    ema.update_weights(student, 3)                 # iter > start -> EMA
    for k, v in ema.model.state_dict().items():
        out["t_after_ema." + k] = v.clone()
    ema.update_weights(student, 2)                 # iter <= start -> copy
    for k, v in ema.model.state_dict().items():
        out["t_after_copy." + k] = v.clone()
    out["alpha"] = np.float64(0.9996)
    save("g5_ema", **out)


# ----------------------------------------------------------------------------
# G6  step trace of run_model_labeled_unlabeled  (aldi/trainer.py:28-117)
# ----------------------------------------------------------------------------
def g6(m):
    import json
    T = m["trainer"]
    traces = {}

    class FakeModel:
        def __init__(self, img, ins, log):
            self.img_align = object() if img else None
            self.ins_align = object() if ins else None
            self.log = log
            self.n = 0

        def __call__(self, data, **kw):
            self.n += 1
            self.log.append(["model", [d["id"] for d in data], dict(sorted(kw.items()))])
            base = float(self.n)
            out = {"loss_cls": torch.tensor(base, requires_grad=True) * 1.0,
                   "loss_rpn_cls": torch.tensor(base + 0.25, requires_grad=True) * 1.0}
            if kw.get("do_align"):
                out["loss_da_img"] = torch.tensor(base + 0.5, requires_grad=True) * 1.0
            elif self.img_align is not None:
                out["_da"] = torch.tensor(0.0, requires_grad=True) * 1.0
            return out

    class FakeDistiller:
        def __init__(self, on, log):
            self.on, self.log = on, log

        def distill_enabled(self):
            return self.on

        def __call__(self, t, s):
            self.log.append(["distiller", [d["id"] for d in t], [d["id"] for d in s]])
            return {"loss_cls": torch.tensor(7.0, requires_grad=True) * 0.0,
                    "loss_cls_ce": torch.tensor(3.0, requires_grad=True) * 1.0}

    class FakeTrainer:
        pass

    def run(name, contents, align, distill, bae, nper=2, ims_per_gpu=2):
        log = []
        tr = FakeTrainer()
        tr.model = FakeModel(align, align, log)
        tr.distiller = FakeDistiller(distill, log)
        tr.backward_at_end = bae
        tr.model_batch_size = ims_per_gpu

        def do_backward(losses, override=False):
            log.append(["backward", round(float(losses), 6)])
        tr.do_backward = do_backward
        lab = [{"id": f"L{i}", "img_weak": f"w{i}"} for i in range(nper)]
        unl = [{"id": f"U{i}", "img_weak": f"w{i}"} for i in range(nper)]
        has_unl = any(c.startswith("unlabeled") for c in contents)
        data = m["dataloader"].unpack_data_weak_strong(lab, unl if has_unl else None, batch_contents=contents)
        ld = T.run_model_labeled_unlabeled(tr, *data)
        traces[name] = {"contents": list(contents), "align": align, "distill": distill, "backward_at_end": bae,
                        "nper": nper, "ims_per_gpu": ims_per_gpu, "log": log,
                        "loss_dict": {k: round(float(v), 6) for k, v in ld.items()},
                        "requires_grad": {k: bool(getattr(v, "requires_grad", False)) for k, v in ld.items()}}
    for bae in (False, True):
        s = "_bae" if bae else ""
        run("source_only" + s, ("labeled_strong",), False, False, bae)
        run("aldi_best" + s, ("labeled_strong", "unlabeled_strong"), False, True, bae)
        run("aldi_align" + s, ("labeled_strong", "unlabeled_strong"), True, True, bae)
        run("weak_too" + s, ("labeled_weak", "labeled_strong", "unlabeled_strong"), True, True, bae)
    run("chunks4", ("labeled_strong", "unlabeled_strong"), False, True, False, nper=4, ims_per_gpu=2)
    with open(os.path.join(HERE, "g6_step_trace.json"), "w") as f:
        json.dump(traces, f, indent=1, sort_keys=True)
    print("wrote g6_step_trace.json")


# ----------------------------------------------------------------------------
# G7  unpack_data_weak_strong  (aldi/dataloader.py:57-80)
# ----------------------------------------------------------------------------
def g7(m):
    import json
    res = {}
    lab = [{"image": "Ls0", "img_weak": "Lw0"}, {"image": "Ls1", "img_weak": "Lw1"}]
    unl = [{"image": "Us0", "img_weak": "Uw0"}]
    for contents in (("labeled_weak",), ("labeled_strong",), ("labeled_strong", "unlabeled_strong"),
                     ("labeled_weak", "labeled_strong", "unlabeled_weak"), ("unlabeled_weak",)):
        out = m["dataloader"].unpack_data_weak_strong(lab, unl, batch_contents=contents)
        res["|".join(contents)] = [None if o is None else [d["image"] for d in o] for o in out]
    res["labeled_none"] = [None if o is None else [d["image"] for d in o]
                           for o in m["dataloader"].unpack_data_weak_strong(None, unl, batch_contents=("labeled_strong", "unlabeled_strong"))]
    with open(os.path.join(HERE, "g7_unpack.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote g7_unpack.json")


# ----------------------------------------------------------------------------
# G8  ALDIDistiller.__call__ hard-loss masking  (aldi/distill.py:170-191)
# ----------------------------------------------------------------------------
def g8(m):
    import json
    res = {}
    for flags in ((False, False, False, False), (True, False, True, False), (True, True, True, True)):
        d = _bare_distiller(m, do_hard_cls=flags[0], do_hard_obj=flags[1], do_hard_rpn_reg=flags[2], do_hard_roi_reg=flags[3])
        d._distill_forward = lambda t, s: {"loss_cls": torch.tensor(1.5), "loss_box_reg": torch.tensor(2.5),
                                           "loss_rpn_cls": torch.tensor(3.5), "loss_rpn_loc": torch.tensor(4.5)}
        d.get_rpn_losses = lambda t: {"loss_obj_bce": torch.tensor(0.1), "loss_rpn_l1": torch.tensor(0.2)}
        d.get_roih_losses = lambda: {"loss_cls_ce": torch.tensor(0.3), "loss_roih_l1": torch.tensor(0.4)}
        L = d([], [])
        res["".join("1" if f else "0" for f in flags)] = {"keys": list(L.keys()), "values": {k: round(float(v), 6) for k, v in L.items()},
                                                           "enabled": bool(d.distill_enabled())}
    d = _bare_distiller(m, do_cls_dst=False, do_obj_dst=False, do_rpn_reg_dst=False, do_roih_reg_dst=False)
    res["all_off_enabled"] = bool(d.distill_enabled())
    base = m["distill"].Distiller(None, None)
    res["base_distiller"] = {"enabled": bool(base.distill_enabled()), "call": base([], [])}
    with open(os.path.join(HERE, "g8_hard_mask.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote g8_hard_mask.json")


# ----------------------------------------------------------------------------
# G9  strong augmentation, ALDI-owned transforms (aldi/aug.py:80-171): blur (real scipy), random erase, MIC block mask
#     (cv2.resize NEAREST stubbed above).  RNG seeds are stored so that the drawn parameters can be re-derived.
# ----------------------------------------------------------------------------
def g9(m):
    aug = m["aug"]
    rng = np.random.default_rng(2024)
    out = {}
    for tag, (H, W) in (("a", (40, 56)), ("b", (33, 47))):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 3] = (img[: H // 3].astype(np.int32) // 8 + 220).clip(0, 255).astype(np.uint8)     # a bright band: exercises the clip
        out[f"img_{tag}"] = img
        for i, seed in enumerate((3, 17, 99)):
            random.seed(seed)
            out[f"blur_{tag}{i}"] = aug.RandomBlurTransform((0.1, 2.0)).apply_image(img.copy())
            out[f"blur_{tag}{i}_seed"] = np.int64(seed)
        for i, (seed, (sl, sh, r1, r2)) in enumerate(((5, (0.05, 0.2, 0.3, 3.3)), (6, (0.02, 0.2, 0.1, 6)), (7, (0.02, 0.2, 0.05, 8)))):
            random.seed(seed)
            np.random.seed(seed)
            out[f"erase_{tag}{i}"] = aug.RandomEraseTransform(sl=sl, sh=sh, r1=r1, r2=r2, value="random").apply_image(img.copy())
            out[f"erase_{tag}{i}_cfg"] = np.array([seed, sl, sh, r1, r2], dtype=np.float64)
        for i, (seed, ratio, block) in enumerate(((11, 0.5, 8), (12, 0.3, 5))):
            np.random.seed(seed)
            out[f"mic_{tag}{i}"] = aug.MICTransform(ratio, block).apply_image(img.copy())
            out[f"mic_{tag}{i}_cfg"] = np.array([seed, ratio, block], dtype=np.float64)
    save("g9_aug", **out)


# ----------------------------------------------------------------------------
# G10  ConvNeXt trunk (aldi/backbone.py:189-352): the reference's own class on a small configuration -- state_dict, input,
#      the four normalised stage outputs, and every parameter / input gradient of a weighted sum of the outputs
# ----------------------------------------------------------------------------
def g10(m):
    torch.manual_seed(21)
    B = m["backbone"]
    net = B.ConvNeXt(in_chans=3, depths=[1, 1, 2, 1], dims=[32, 64, 96, 128], drop_path_rate=0.0, layer_scale_init_value=1e-6,
                     out_features=[0, 1, 2, 3]).float()
    with torch.no_grad():           # the stock init leaves gamma at 1e-6 and every bias at 0: give each parameter a value that matters
        for n, p in net.named_parameters():
            if n.endswith("gamma"):
                p.copy_(0.5 + 0.2 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
            elif "norm" in n or n.endswith("downsample_layers.0.1.weight") or (".0.weight" in n and "downsample_layers." in n and p.dim() == 1):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "dwconv.weight" in n:
                p.copy_(torch.randn_like(p) * 0.15)
            else:
                p.copy_(torch.randn_like(p) * (1.5 / max(p[0].numel(), 1)) ** 0.5)
            p.copy_(p.bfloat16().float())                         # both sides use bf16-representable weights
    img = torch.randint(0, 256, (2, 3, 64, 96), dtype=torch.uint8)
    mean = torch.tensor([103.530, 116.280, 123.675]).view(1, 3, 1, 1)          # detectron2 defaults, kept by Base-RCNN-ConvNeXt-FPN.yaml
    x = (img.float() - mean).requires_grad_(True)
    outs = net(x)
    gs = [torch.randn_like(outs[i]).bfloat16().float() for i in range(4)]
    torch.autograd.backward([outs[i] for i in range(4)], gs)
    out = {"img": img}
    for i in range(4):
        out[f"out{i}"] = outs[i].detach()
        out[f"gout{i}"] = gs[i]
    for k, v in net.state_dict().items():
        out["sd." + k] = v
    for k, p in net.named_parameters():
        out["grad." + k] = p.grad
    out["keys"] = np.array(list(net.state_dict().keys()))
    save("g10_convnext", **out)


# ----------------------------------------------------------------------------
# G11  discriminators with other hidden_dims lists (aldi/align.py:103-135 build ANY list: two hidden layers, none at all)
# ----------------------------------------------------------------------------
def g11(m):
    out = {}
    torch.manual_seed(23)
    nets = {"conv2": (m["align"].ConvDiscriminator(32, hidden_dims=[16, 32]), torch.randn(2, 32, 9, 11)),
            "conv0": (m["align"].ConvDiscriminator(32, hidden_dims=[]), torch.randn(2, 32, 5, 6)),
            "fc2": (m["align"].FCDiscriminator(64, hidden_dims=[32, 16]), torch.randn(16, 64)),
            "fc0": (m["align"].FCDiscriminator(64, hidden_dims=[]), torch.randn(16, 64))}
    for tag, (net, x) in nets.items():
        for labeled in (1, 0):
            xi = x.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            preds = net(m["helpers"].grad_reverse(xi))
            loss = 0.01 * F.binary_cross_entropy_with_logits(preds, torch.FloatTensor(preds.data.size()).fill_(labeled))
            loss.backward()
            out[f"{tag}_x"] = x
            out[f"{tag}_preds"] = preds
            out[f"{tag}_loss_l{labeled}"] = loss
            out[f"{tag}_dx_l{labeled}"] = xi.grad
            for k, p in net.named_parameters():
                out[f"{tag}_grad_l{labeled}.{k}"] = p.grad
        for k, v in net.state_dict().items():
            out[f"{tag}_sd.{k}"] = v
        out[f"{tag}_keys"] = np.array(list(net.state_dict().keys()))
    save("g11_discriminators_deep", **out)


if __name__ == "__main__":
    random.seed(0)
    mods = import_reference()
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11):
        if not only or fn.__name__ in only:
            fn(mods)
