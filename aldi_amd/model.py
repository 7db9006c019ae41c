"""``build_aldi`` and the GeneralizedRCNN meta-architecture, same plugin surface as the reference
(aldi/model.py:12-34): ``class ALDI(align_mixin, distill_mixin, base_cls)`` composed from the three
registries, ``model(batched_inputs, labeled=True, do_align=False) -> dict`` in training and
``model.inference(batched_inputs, do_postprocess=False) -> list[Instances]``.

The model is not an nn.Module tree: it owns an engine (aldi_amd.engine.RCNN) that runs the HIP
kernels.  Losses come back as 0-d device tensors wired into torch autograd through a small
bridge, so the reference's driver code (`sum(losses)/n`, `v * 0`, `.backward()`,
aldi/trainer.py:61-79) works unchanged: autograd delivers d(total)/d(loss_k) to the bridge, which
launches ONE engine backward per forward with those coefficients.
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from typing import Dict, List

import torch

from . import synthetic
from .arch import ParamLayout
from .engine import RCNN, Ctx, D2Params, Weights
from .helpers import HookPoint
from .registry import Registry
from .structures import Boxes, Instances, as_record

META_ARCH_REGISTRY = Registry("META_ARCH")


class _Holder:
    """Everything one forward leaves behind for its (single) backward."""
    def __init__(self, model, ctx: Ctx):
        self.model, self.ctx = model, ctx
        self.scales: Dict[str, float] = {}
        self.root = None
        self.done = False


class _Root(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, holder):
        ctx.holder = holder
        return anchor.new_zeros(())

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        if not h.done:                      # all _Losses.backward of this forward have run by now
            h.done = True
            h.model.engine.backward(h.ctx, h.scales)
        return None, None


class _Losses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, root, holder, names, *values):
        ctx.holder, ctx.names = holder, names
        return tuple(v.detach().clone() for v in values)

    @staticmethod
    def backward(ctx, *grads):
        dev = ctx.holder.model.device
        vals = torch.stack([g.to(torch.float32).reshape(()) if g is not None else torch.zeros((), device=dev) for g in grads]).tolist()
        for n, v in zip(ctx.names, vals):
            ctx.holder.scales[n] = ctx.holder.scales.get(n, 0.0) + v
        return (torch.zeros((), device=dev), None, None) + (None,) * len(grads)


def wire_losses(holder: _Holder, losses: "OrderedDict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
    names = tuple(losses.keys())
    outs = _Losses.apply(holder.root, holder, names, *losses.values())
    return OrderedDict(zip(names, outs))


class DevicePseudoLabels(Instances):
    """Pseudo-label `Instances` that live on the GPU; the reference moves them to the CPU
    (aldi/pseudolabeler.py:63-65) -- here the CPU view is materialised only if someone reads it."""
    def __init__(self, image_size, dev: dict, index: int):
        super().__init__(image_size)
        object.__setattr__(self, "_dev", dev)
        object.__setattr__(self, "_index", index)

    def _materialise(self):
        if not self._fields:
            n = int(self._dev["count"][self._index])
            self.set("gt_boxes", Boxes(self._dev["boxes"][self._index, :n].cpu()))
            self.set("gt_classes", self._dev["classes"][self._index, :n].to(torch.int64).cpu())
            self.set("scores", self._dev["scores"][self._index, :n].cpu())

    def __getattr__(self, name):
        if name in ("gt_boxes", "gt_classes", "scores"):
            self._materialise()
        return Instances.__getattr__(self, name)

    def __len__(self):
        return int(self._dev["count"][self._index])


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN:
    """R50-FPN Faster R-CNN on the HIP engine (detectron2 GeneralizedRCNN's role; configs/detectron2/Base-RCNN-FPN.yaml)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        da = cfg.get("DOMAIN_ADAPT", {}).get("ALIGN", {}) if hasattr(cfg, "get") else {}
        # False, or the discriminator's shape (any FPN level / hidden_dims list: aldi/align.py:22-52, aldi/config.py:41-49)
        self._img_da = dict(layer=da.get("IMG_DA_LAYER", "p2"), input_dim=da.get("IMG_DA_INPUT_DIM", 256),
                            hidden_dims=list(da.get("IMG_DA_HIDDEN_DIMS", [256]))) if da.get("IMG_DA_ENABLED", False) else False
        self._ins_da = dict(input_dim=da.get("INS_DA_INPUT_DIM", 1024),
                            hidden_dims=list(da.get("INS_DA_HIDDEN_DIMS", [1024]))) if da.get("INS_DA_ENABLED", False) else False
        self.device = torch.device(cfg.MODEL.DEVICE)
        if self.device.type != "cuda":
            raise RuntimeError("aldi_amd runs on the MI355X HIP path only (MODEL.DEVICE must be cuda); there is no CPU fallback")
        self.dtype = torch.bfloat16 if cfg.SOLVER.AMP.ENABLED else torch.float32
        self.vitdet = str(cfg.MODEL.BACKBONE.NAME).startswith("build_vitdet")
        seed = cfg.SEED if cfg.SEED is not None and cfg.SEED >= 0 else 1
        self.convnext = str(cfg.MODEL.BACKBONE.NAME) == "build_convnext_fpn_backbone"
        self.adamw = self.vitdet or self.convnext          # flat-container models are trained with the HIP AdamW
        if self.vitdet:
            self._build_vitdet(seed)
        elif self.convnext:
            self._build_convnext(seed)
        else:
            self.layout = ParamLayout(self.num_classes, self._img_da, self._ins_da)
            self.weights = Weights(self.layout, self.device, self.dtype, trainable=True)
            self.engine = RCNN(self.weights, self.num_classes, D2Params.from_cfg(cfg))
        self.training = True
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self._last: _Holder = None
        # hook points with the reference's module paths (aldi/distill.py:122-138, aldi/align.py:46-52)
        self.backbone = HookPoint(self, "backbone")
        self.proposal_generator = HookPoint(self, "proposal_generator")
        self.proposal_generator.rpn_head = HookPoint(self, "rpn_head")
        self.proposal_generator.anchor_generator = HookPoint(self, "anchor_generator")
        self.roi_heads = HookPoint(self, "roi_heads")
        self.roi_heads.box_predictor = HookPoint(self, "box_predictor")
        self.roi_heads.box_head = HookPoint(self, "box_head")
        # Construction initialises; cfg.MODEL.WEIGHTS is read by the checkpointer (`Trainer.resume_or_load`, reference
        # tools/train_net.py:84 / detectron2 build_model), with its key matching and `ema` handling -- not here.
        if self.vitdet or self.convnext:
            self.weights.init_random(seed)
        else:
            self.load_state_dict(synthetic.init_state_dict(self.num_classes, seed=seed, img_da=self._img_da, ins_da=self._ins_da))

    def vit_config(self):
        """reference aldi/backbone.py:36-64 (build_vitdet_b_backbone / build_vitdet_l_backbone over detectron2's model_zoo
        common/models/mask_rcnn_vitdet.py) + configs/Base-RCNN-VitDetB.yaml:7-19"""
        from .vit import VitConfig
        cfg = self.cfg
        M = cfg.MODEL
        if M.ROI_BOX_HEAD.NORM != "LN" or M.ROI_BOX_HEAD.NUM_FC != 1 or list(M.RPN.CONV_DIMS) != [-1, -1]:
            raise ValueError("the ViTDet engine implements the head layout of configs/Base-RCNN-VitDetB.yaml (NORM LN, NUM_FC 1, RPN.CONV_DIMS [-1, -1])")
        large = "vitdet_l" in M.BACKBONE.NAME
        kw = dict(embed=1024, depth=24, heads=16, drop_path_rate=0.4,
                  global_blocks=(5, 11, 17, 23)) if large else dict(embed=768, depth=12, heads=12, drop_path_rate=0.1, global_blocks=(2, 5, 8, 11))
        syn = cfg.get("SYNTHETIC", {})
        return VitConfig(sfp=True, num_classes=self.num_classes, box_convs=M.ROI_BOX_HEAD.NUM_CONV, fc_dim=M.ROI_BOX_HEAD.FC_DIM,
                         pixel_mean=tuple(M.PIXEL_MEAN), pixel_std=tuple(M.PIXEL_STD), **{**kw, **dict(syn.get("VIT", {}))})

    def _build_convnext(self, seed: int):
        """reference aldi/backbone.py:354-392 (build_convnext_fpn_backbone) + configs/Base-RCNN-ConvNeXt-FPN.yaml"""
        from .convnext import ConvNeXtConfig, ConvNeXtRCNN
        from .vit import VitParams
        if self._img_da or self._ins_da:
            raise ValueError("adversarial alignment is not wired for the ConvNeXt trunk")
        if self.dtype != torch.bfloat16:
            raise ValueError("the ConvNeXt trunk runs in bf16 (SOLVER.AMP.ENABLED True)")
        M = self.cfg.MODEL
        ccfg = ConvNeXtConfig(depths=tuple(M.CONVNEXT.DEPTHS), dims=tuple(M.CONVNEXT.DIMS), drop_path_rate=float(M.CONVNEXT.DROP_PATH_RATE),
                              layer_scale_init_value=float(M.CONVNEXT.LAYER_SCALE_INIT_VALUE), num_classes=self.num_classes,
                              fc_dim=M.ROI_BOX_HEAD.FC_DIM, anchor_sizes=tuple(int(s_[0]) for s_ in M.ANCHOR_GENERATOR.SIZES),
                              pixel_mean=tuple(M.PIXEL_MEAN), pixel_std=tuple(M.PIXEL_STD))
        self.weights = VitParams(ccfg, self.device)
        self.layout = self.weights
        self.engine = ConvNeXtRCNN(self.weights, self.num_classes, seed=seed)

    def _build_vitdet(self, seed: int):
        from .vit import VitParams
        from .vitdet import VitDetRCNN
        if self._img_da or self._ins_da:
            raise ValueError("adversarial alignment is not wired for the ViTDet trunk")
        if self.dtype != torch.bfloat16:
            raise ValueError("the ViTDet trunk runs in bf16 (SOLVER.AMP.ENABLED True): its attention kernels are bf16 MFMA")
        self.weights = VitParams(self.vit_config(), self.device)
        self.layout = self.weights
        self.engine = VitDetRCNN(self.weights, self.num_classes, seed=seed)

    # ---- nn.Module-like surface -------------------------------------------------------------
    def to(self, device):
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        return self.weights.state_dict()

    def load_state_dict(self, sd, strict: bool = True):
        self.weights.load_state_dict(sd)

    def parameters(self):
        return iter([self.weights.master[: self.layout.n_train]])

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        for k, v in self.__dict__.items():
            if k in ("weights", "engine", "_anchor", "_last", "backbone", "proposal_generator", "roi_heads"):
                continue
            setattr(new, k, copy.deepcopy(v, memo) if k not in ("cfg", "layout", "device", "dtype") else v)
        if self.vitdet:
            new._build_vitdet(1)
        elif self.convnext:
            new._build_convnext(1)
        else:
            new.weights = Weights(self.layout, self.device, self.dtype, trainable=True)
            new.engine = RCNN(new.weights, self.num_classes, self.engine.p)
        new.weights.master.copy_(self.weights.master)
        new.weights.refresh()
        new._anchor = torch.zeros((), device=self.device, requires_grad=True)
        new._last = None
        new.backbone = HookPoint(new, "backbone")
        new.proposal_generator = HookPoint(new, "proposal_generator")
        new.proposal_generator.rpn_head = HookPoint(new, "rpn_head")
        new.proposal_generator.anchor_generator = HookPoint(new, "anchor_generator")
        new.roi_heads = HookPoint(new, "roi_heads")
        new.roi_heads.box_predictor = HookPoint(new, "box_predictor")
        new.roi_heads.box_head = HookPoint(new, "box_head")
        return new

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    # ---- forward ----------------------------------------------------------------------------
    def _split_inputs(self, batched_inputs):
        images = [b["image"] for b in batched_inputs]
        insts = [b.get("instances") for b in batched_inputs]
        return images, insts

    def _gt(self, insts):
        """device GT for the engine: reuse device-resident pseudo labels when all images carry them."""
        if insts and all(isinstance(i, DevicePseudoLabels) for i in insts) and len({id(i._dev) for i in insts}) == 1 \
                and [i._index for i in insts] == list(range(len(insts))):
            return insts[0]._dev, None
        return None, [as_record(i) for i in insts]

    def forward(self, batched_inputs: List[Dict], do_align: bool = False, labeled: bool = True):
        if not self.training:
            return self.inference(batched_inputs)
        images, insts = self._split_inputs(batched_inputs)
        gt_dev, recs = self._gt(insts)
        da = self.cfg.DOMAIN_ADAPT.ALIGN if "DOMAIN_ADAPT" in self.cfg else None
        c = self.engine.forward_train(images, recs, pre_roi_hook=lambda: self.roi_heads.fire_pre(), gt_dev=gt_dev,
                                      do_align=do_align and (self._img_da or self._ins_da), labeled=labeled,
                                      da_weights=(da.IMG_DA_WEIGHT, da.INS_DA_WEIGHT) if da is not None else (0.0, 0.0))
        h = _Holder(self, c)
        h.root = _Root.apply(self._anchor, h)
        self._last = h
        # what the reference's forward hooks capture
        self.backbone.fire(None, OrderedDict(zip(("p2", "p3", "p4", "p5", "p6"), c.P)))
        self.proposal_generator.rpn_head.fire(None, c.head)
        self.proposal_generator.fire(None, ((c.props, c.prop_count), None))
        self.roi_heads.box_head.fire(None, c.fc2)
        self.roi_heads.box_predictor.fire(None, c.pred)
        return wire_losses(h, self.engine.loss_dict(c))

    def inference(self, batched_inputs: List[Dict], do_postprocess: bool = False, pl_thresh: float = 2.0):
        """-> list[Instances] with pred_boxes / scores / pred_classes (network-input pixel space)."""
        assert not do_postprocess, "aldi_amd keeps detections in network-input space (the hot path calls do_postprocess=False)"
        with torch.no_grad():
            images, _ = self._split_inputs(batched_inputs)
            self.roi_heads.fire_pre()                       # ManualSeed fires on every roi_heads forward (aldi/helpers.py:25-26)
            c = self.engine.inference(images, pl_thresh)
            self._last_inference = c
            out = []
            cnt = c.det.count.tolist()
            for i, n in enumerate(cnt):
                inst = Instances(c.sizes[i])
                inst.pred_boxes = Boxes(c.det.boxes[i, :n])
                inst.scores = c.det.scores[i, :n]
                inst.pred_classes = c.det.classes[i, :n].to(torch.int64)
                out.append(inst)
            return out


def build_aldi(cfg):
    """Add Align and Distill capabilities to any Meta Architecture dynamically (reference aldi/model.py:12-34)."""
    from .align import ALIGN_MIXIN_REGISTRY
    from .distill import DISTILL_MIXIN_REGISTRY
    if cfg.MODEL.META_ARCHITECTURE == "DeformableDETR":
        from . import detr  # noqa: F401  (registers the detector and its mixins, as importing aldi.detr does in the reference)
    base_cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    align_mixin = ALIGN_MIXIN_REGISTRY.get(cfg.DOMAIN_ADAPT.ALIGN.MIXIN_NAME)
    distill_mixin = DISTILL_MIXIN_REGISTRY.get(cfg.DOMAIN_ADAPT.DISTILL.MIXIN_NAME)

    class ALDI(align_mixin, distill_mixin, base_cls):
        def __init__(self, cfg):
            super(ALDI, self).__init__(cfg)

        def forward(self, batched_inputs: List[Dict[str, torch.Tensor]], labeled: bool = True, do_align: bool = False):
            return super(ALDI, self).forward(batched_inputs, do_align=do_align, labeled=labeled)

    model = ALDI(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
