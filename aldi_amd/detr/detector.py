"""META_ARCHITECTURE "DeformableDETR" (configs/Base-DETR.yaml:2) on the HIP library: the R50 trunk of the Faster R-CNN engine (stem + res2
frozen, C3..C5 out) + the deformable transformer of aldi_amd/detr/model.py + the set criterion of aldi_amd/detr/criterion.py, behind the
same model surface as `GeneralizedRCNN` (aldi_amd/model.py): `model(batched_inputs)` -> dict of weighted losses in training (wired to one
engine backward through `wire_losses`), `model.inference(...)` -> Instances / device-resident pseudo labels, `model.weights` = ONE flat
fp32 state for the optimizer (AdamW + full-model gradient clip), the EMA teacher and the checkpointer.

The reference registers its detector as `DETRDistillMixin(DeformableDETR)` / `DETRAlignMixin(DeformableDETR)` (aldi/detr/distill.py:6-7,
aldi/detr/align.py:6-7) from an ABSENT submodule; the algorithm is restated from the published model and pinned through
oracle/deformable_detr.py.  Differences, stated: the backbone is this repository's R50 (stride on the bottleneck's 1x1 conv, as in the
reference's FPN models) rather than torchvision's (stride on the 3x3 conv); dropout draws from its own stateless generator."""
from __future__ import annotations

import copy
import math
from collections import OrderedDict
from typing import Dict, List

import torch

from .. import ops, synthetic
from ..arch import ParamLayout
from ..engine import RCNN, Ctx, D2Params, Weights
from ..helpers import HookPoint
from ..model import META_ARCH_REGISTRY, DevicePseudoLabels, _Holder, _Root, wire_losses
from ..structures import Boxes, Instances, as_record
from .criterion import SetCriterion
from .model import DeformableTransformer, FlatParams, param_spec

GMAX_PL = 128          # pseudo-label slots per image (top-100 detections at most)


def _init_transformer(P: FlatParams, *, n_heads: int, num_levels: int, enc_points: int, dec_points: int, seed: int):
    """the authors' initialisation: Xavier-uniform matrices, zero biases, sampling offsets pointing at a ring of directions scaled by the
    point index, uniform attention weights, class bias = logit(0.01), the box head's last layer zero with (w, h) bias -2"""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in P.spec.items():
        if k.endswith("sampling_offsets.weight") or k.endswith("attention_weights.weight") or k.endswith("attention_weights.bias"):
            sd[k] = torch.zeros(shp)
        elif k.endswith("sampling_offsets.bias"):
            pts = dec_points if ".decoder." in k else enc_points
            th = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
            grid = torch.stack([th.cos(), th.sin()], -1)
            grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(n_heads, 1, 1, 2).repeat(1, num_levels, pts, 1)
            for i in range(pts):
                grid[:, :, i, :] *= i + 1
            sd[k] = grid.reshape(-1)
        elif k == "query_embed.weight" or k == "transformer.level_embed":
            sd[k] = torch.randn(shp, generator=g)
        elif k == "class_embed.bias":
            sd[k] = torch.full(shp, -math.log((1 - 0.01) / 0.01))
        elif k == "bbox_embed.layers.2.weight":
            sd[k] = torch.zeros(shp)
        elif k == "bbox_embed.layers.2.bias":
            sd[k] = torch.tensor([0.0, 0.0, -2.0, -2.0])
        elif len(shp) >= 2:                                       # Xavier uniform (convs: fan = channels x taps)
            fan_out, fan_in = shp[0] * int(math.prod(shp[2:])), shp[1] * int(math.prod(shp[2:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * a
        elif k.endswith(".weight"):                               # norms
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.zeros(shp)
    P.load(sd)


class DetrWeights:
    """the whole detector's state in ONE fp32 buffer: [R50 state (aldi_amd.engine.Weights layout) | transformer (FlatParams)]; both
    containers keep working on views of it.  What the trainer / EMA / checkpointer need of `model.weights`."""
    def __init__(self, backbone: Weights, tr: FlatParams):
        self.backbone, self.tr = backbone, tr
        self.nb, self.nt = backbone.master.numel(), tr.master.numel()
        self.master = torch.cat([backbone.master, tr.master])
        backbone.master = self.master[:self.nb]
        if backbone.dtype == torch.float32:
            backbone.compute = backbone.master
        tr.master = self.master[self.nb:]
        self.nbg = backbone.layout.n_train
        self.grad = torch.zeros(self.nbg + self.nt, dtype=torch.float32, device=self.master.device)
        backbone._grad = self.grad[:self.nbg]
        tr.grad = self.grad[self.nbg:]
        self.m = self.v = None
        self.step = 0
        self._gscale = 1.0
        self.layout = self
        self.n_train = self.nbg + self.nt
        self.trunk_names = [k for k in backbone.layout.state_dict_keys() if k.startswith("backbone.bottom_up.")]

    # ---- layout surface (EMA / checkpointer) ------------------------------------------------
    def state_dict_keys(self) -> List[str]:
        return self.trunk_names + list(self.tr.spec.keys())

    def ranges(self, names):
        """(lo, hi) in the MASTER buffer"""
        out = []
        for k in names:
            if k not in self.tr.spec:
                raise KeyError(f"{k}: only the transformer's tensors are addressable by state-dict name")
            a, b = self.tr.ranges([k])[0]
            out.append((self.nb + a, self.nb + b))
        return out

    def state_dict(self):
        sd = OrderedDict((k, v) for k, v in self.backbone.state_dict().items() if k.startswith("backbone.bottom_up."))
        sd.update(self.tr.state_dict())
        return sd

    def load_state_dict(self, sd, strict: bool = True):
        """strict: every key of this detector must be present (KeyError lists the missing ones).  strict=False loads the intersection --
        e.g. a backbone-only R50 pretrain, the usual Deformable-DETR initialisation -- and returns (missing, unexpected) like torch does"""
        full = self.backbone.state_dict()
        mine = [k for k in full if k.startswith("backbone.bottom_up.")] + list(self.tr.spec)
        missing = [k for k in mine if k not in sd]
        unexpected = [k for k in sd if k not in full and k not in self.tr.spec]
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        full.update({k: v for k, v in sd.items() if k in full})
        self.backbone.load_state_dict(full)
        cur = self.tr.state_dict() if missing else {}
        self.tr.load({k: sd[k] if k in sd else cur[k] for k in self.tr.spec})
        self.refresh()
        if missing:
            import logging
            logging.getLogger(__name__).warning("DetrWeights.load_state_dict: %d keys kept at their current values (e.g. %s)", len(missing), missing[0])
        return missing, unexpected

    def refresh(self, cast: bool = True):
        self.backbone.refresh()

    def zero_grad(self):
        self.grad.zero_()
        self._gscale = 1.0

    def scale_grad(self, f: float):
        self._gscale *= f

    def ema_from(self, student: "DetrWeights", alpha: float, copy_only: bool = False):
        ops.ema_update(self.master, student.master, None, self.master.numel(), alpha, copy_only, torch.float32)
        self.refresh()

    # ---- optimizer --------------------------------------------------------------------------
    def grad_norm(self) -> float:
        return float(torch.linalg.vector_norm(self.grad))          # (one reduction + the step's second host read)

    def adamw_step(self, lr: float, *, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, clip: float = 0.0, backbone_mult: float = 0.1,
                   proj_mult: float = 0.1, proj_names=("reference_points", "sampling_offsets")):
        """torch.optim.AdamW with the reference's parameter groups (configs/Base-DETR.yaml:59-69): backbone x BACKBONE_LR_MULTIPLIER,
        reference_points / sampling_offsets x LR_LINEAR_PROJ_MULTIPLIER; full-model gradient-norm clip (CLIP_VALUE, L2)"""
        from .. import vit_ops as V
        gs = self._gscale
        if clip and clip > 0:
            norm = self.grad_norm() * gs
            gs *= min(1.0, clip / (norm + 1e-6))
        if self.m is None:
            self.m, self.v = torch.zeros_like(self.grad), torch.zeros_like(self.grad)
        self.step += 1
        nbg = self.nbg
        # backbone: its trainable prefix (res3..res5; what the detector does not use of the R50 layout never receives a gradient and has
        # no weight decay applied to it either: only the trunk's ranges are stepped)
        for a, b in self.backbone.layout.ranges([k for k in self.backbone.layout.t if k.startswith(("backbone.bottom_up.res3", "backbone.bottom_up.res4", "backbone.bottom_up.res5"))]):
            if b <= nbg and b > a:
                V.adamw_step(self.master[a:b], self.grad[a:b], self.m[a:b], self.v[a:b], None, lr=lr * backbone_mult, betas=betas, eps=eps,
                             weight_decay=weight_decay, step=self.step, grad_scale=gs)
        segs, at = [], 0
        for k in self.tr.spec:
            if any(s_ in k for s_ in proj_names):
                a, b = self.tr.ranges([k])[0]
                if a > at:
                    segs.append((at, a, 1.0))
                segs.append((a, b, proj_mult))
                at = b
        segs.append((at, self.nt, 1.0))
        for a, b, f in segs:
            if b > a:
                V.adamw_step(self.master[self.nb + a:self.nb + b], self.grad[nbg + a:nbg + b], self.m[nbg + a:nbg + b], self.v[nbg + a:nbg + b], None,
                             lr=lr * f, betas=betas, eps=eps, weight_decay=weight_decay, step=self.step, grad_scale=gs)
        self._gscale = 1.0
        self.refresh()


class _DetrEngine:
    """what `wire_losses` / `_Root.backward` call: engine.backward(ctx, scales)"""
    def __init__(self, model: "DeformableDETR"):
        self.m = model

    def backward(self, c: Ctx, scales: Dict[str, float]):
        m = self.m
        if m._last is None or m._last.ctx is not c:
            # the detector keeps ONE tape (transformer.tape, the trunk's saved activations): only the most recent forward can be differentiated
            raise RuntimeError("DeformableDETR: backward of an earlier forward (SOLVER.BACKWARD_AT_END True with several micro-steps); "
                               "set SOLVER.BACKWARD_AT_END False as configs/Base-DETR.yaml does")
        per = {}
        for k, v in scales.items():                               # one upstream factor per loss TYPE (the layers' copies share it)
            base = "_".join(k.split("_")[:2])
            if base in per and abs(per[base] - v) > 1e-12:
                raise NotImplementedError("different gradient scales for the decoder layers' copies of one loss")
            per[base] = v
        s = [per.get("loss_ce", 0.0), per.get("loss_bbox", 0.0), per.get("loss_giou", 0.0)]
        crit = m.criterion
        if s == [1.0, 1.0, 1.0]:
            gl, gb = c.g_logits, c.g_boxes
        else:                                                     # re-run the loss kernel with the scaled coefficients (same matching)
            saved = crit.coef
            crit.coef = tuple(a * b for a, b in zip(saved, s))
            try:
                _, gl, gb = crit(c.logits, c.boxes, c.targets, num_boxes=c.num_boxes, match=c.match)
            finally:
                crit.coef = saved
        gfeats = m.transformer.backward(gl, gb)
        m.bengine.trunk_backward(c.trunk, {3: gfeats[0], 4: gfeats[1], 5: gfeats[2]})


@META_ARCH_REGISTRY.register()
class DeformableDETR:
    def __init__(self, cfg):
        self.cfg = cfg
        dd = cfg.MODEL.DEFORMABLE_DETR
        T = dd.TRANSFORMER
        self.device = torch.device(cfg.MODEL.DEVICE)
        if self.device.type != "cuda":
            raise RuntimeError("aldi_amd runs on the MI355X HIP path only (MODEL.DEVICE must be cuda); there is no CPU fallback")
        if cfg.SOLVER.AMP.ENABLED:
            raise ValueError("DeformableDETR runs in fp32 (SOLVER.AMP.ENABLED False, as configs/Base-DETR.yaml:56-58)")
        if dd.WITH_BOX_REFINE or dd.TWO_STAGE or dd.BACKBONE != "resnet50" or dd.DILATION or dd.POSITION_EMBEDDING != "sine":
            raise ValueError("DeformableDETR: only the plain variant (ResNet-50, sine embedding, no box refinement / two-stage / dilation)")
        self.dtype = torch.float32
        self.num_classes = int(dd.NUM_CLASSES)
        self.adamw = True
        self.detr = True
        self.vitdet = self.convnext = False
        seed = cfg.SEED if cfg.SEED is not None and cfg.SEED >= 0 else 1
        self._build(seed)
        L_ = dd.LOSS
        Mt = dd.MATCHER
        self.criterion = SetCriterion(cls_coef=L_.CLS_LOSS_COEF, bbox_coef=L_.BBOX_LOSS_COEF, giou_coef=L_.GIOU_LOSS_COEF, cost_class=Mt.SET_COST_CLASS,
                                      cost_bbox=Mt.SET_COST_BBOX, cost_giou=Mt.SET_COST_GIOU, focal_alpha=L_.FOCAL_ALPHA)
        self.aux_loss = bool(L_.AUX_LOSS)
        self.training = True
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self._last = None
        self.backbone = HookPoint(self, "backbone")

    def _dims(self):
        dd = self.cfg.MODEL.DEFORMABLE_DETR
        T = dd.TRANSFORMER
        return dict(d_model=int(T.HIDDEN_DIM), num_levels=int(dd.NUM_FEATURE_LEVELS), enc_layers=int(T.ENC_LAYERS), dec_layers=int(T.DEC_LAYERS), n_heads=int(T.NHEADS),
                    enc_points=int(T.ENC_N_POINTS), dec_points=int(T.DEC_N_POINTS))

    def _build(self, seed: int):
        dd = self.cfg.MODEL.DEFORMABLE_DETR
        T = dd.TRANSFORMER
        lay = ParamLayout(self.num_classes, False, False)
        bw = Weights(lay, self.device, torch.float32, trainable=True)
        std = float(sum(self.cfg.MODEL.PIXEL_STD) / 3.0)
        bw.load_state_dict(synthetic.init_state_dict(self.num_classes, seed=seed, input_rms=75.0 / max(std, 1e-6)))
        dims = self._dims()
        P = FlatParams(param_spec(ffn=int(T.DIM_FEEDFORWARD), num_queries=int(T.NUM_QUERIES), num_classes=self.num_classes, **dims), self.device, True)
        _init_transformer(P, n_heads=dims["n_heads"], num_levels=dims["num_levels"], enc_points=dims["enc_points"], dec_points=dims["dec_points"], seed=seed)
        self.weights = DetrWeights(bw, P)
        self.weights.refresh()
        self.layout = self.weights
        self.bengine = RCNN(bw, self.num_classes, D2Params.from_cfg(self.cfg))
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.transformer = DeformableTransformer(P, device=self.device, dropout=float(T.DROPOUT), seed=seed + 7919 * rank, **dims)    # (dropout masks differ by rank)
        self.engine = _DetrEngine(self)

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
        return iter([self.weights.master])

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        for k, v in self.__dict__.items():
            if k in ("weights", "layout", "bengine", "transformer", "engine", "_anchor", "_last", "backbone", "criterion"):
                continue
            setattr(new, k, copy.deepcopy(v, memo) if k not in ("cfg", "device", "dtype") else v)
        new._build(1)
        new.weights.master.copy_(self.weights.master)
        new.weights.refresh()
        new.criterion = self.criterion
        new._anchor = torch.zeros((), device=self.device, requires_grad=True)
        new._last = None
        new.backbone = HookPoint(new, "backbone")
        return new

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    # ---- forward ----------------------------------------------------------------------------
    def _trunk(self, images, save: bool):
        E = self.bengine
        st, sizes, hw = E.stage_images(images)
        c = E._drive(E.trunk_steps(st, sizes, save, fpn=False))
        mask = torch.ones((len(images), st.shape[2], st.shape[3]), dtype=torch.bool)
        for i, (h, w) in enumerate(sizes):
            mask[i, :h, :w] = False
        return c, sizes, mask

    @staticmethod
    def _targets(insts, sizes):
        """ground truth in absolute xyxy pixels -> the criterion's (cx, cy, w, h) / image size"""
        out = []
        for inst, (h, w) in zip(insts, sizes):
            r = as_record(inst)
            b = r["gt_boxes"].to(torch.float32).cpu().view(-1, 4)
            cxcywh = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1) / torch.tensor([w, h, w, h], dtype=torch.float32)
            out.append({"labels": r["gt_classes"].to(torch.long).cpu().view(-1), "boxes": cxcywh})
        return out

    def _check_labels(self, targets):
        for t in targets:
            if len(t["labels"]) and (int(t["labels"].min()) < 0 or int(t["labels"].max()) >= self.num_classes):
                raise ValueError(f"ground-truth class outside [0, {self.num_classes}) for MODEL.DEFORMABLE_DETR.NUM_CLASSES = {self.num_classes}")

    def forward(self, batched_inputs: List[Dict], do_align: bool = False, labeled: bool = True):
        if not self.training:
            return self.inference(batched_inputs)
        if do_align:
            raise NotImplementedError("adversarial alignment is not implemented for DeformableDETR")
        images = [b["image"] for b in batched_inputs]
        ctr, sizes, mask = self._trunk(images, save=True)
        feats = [ctr.cs[1], ctr.cs[2], ctr.cs[3]]
        logits, boxes = self.transformer.forward(feats, mask, record=True, feats_need_grad=True)
        if not self.aux_loss:
            logits, boxes = logits[-1:], boxes[-1:]
        targets = self._targets([b["instances"] for b in batched_inputs], sizes)
        self._check_labels(targets)
        losses, gl, gb = self.criterion(logits, boxes, targets)
        if not self.aux_loss:                                     # gradients for the full stack of layers: zeros for the unused ones
            full_l, full_b = self.transformer._out[0].new_zeros(self.transformer._out[0].shape), self.transformer._out[1].new_zeros(self.transformer._out[1].shape)
            n = gl.numel()
            full_l.view(-1)[-n:] = gl.view(-1)
            full_b.view(-1)[-gb.numel():] = gb.view(-1)
            gl, gb = full_l, full_b
        c = Ctx()
        c.trunk, c.logits, c.boxes, c.targets, c.g_logits, c.g_boxes = ctr, logits, boxes, targets, gl, gb
        # the normaliser and the assignment the forward's losses were computed with: a re-run of the loss kernel in the backward (gradient
        # accumulation / masked losses: a scale != 1) must divide by the SAME count -- the world mean under data parallelism, not this
        # rank's -- and needs no second collective
        c.num_boxes, c.match = self.criterion.last_num_boxes, self.criterion.last_match
        h = _Holder(self, c)
        h.root = _Root.apply(self._anchor, h)
        self._last = h
        return wire_losses(h, losses)

    # ---- the fused student pass --------------------------------------------------------------
    def forward_fused(self, parts: List[List[Dict]]):
        """the micro-batches of one iteration (the sequential driver's IMS_PER_GPU-sized chunks: source, pseudo-labelled target) through ONE
        trunk + transformer pass; the set criterion runs PER CHUNK on its slice of the outputs -- its own Hungarian assignment and its own
        normaliser (the chunk's target count, world mean), exactly what `model(chunk)` computes -- so every kernel of the trunk and the
        transformer sees twice the rows and the step issues half the launches.  Batch elements are independent in this detector except
        through the padded canvas (GroupNorm statistics of the input projections run over it): the trainer fuses only chunks whose canvases
        coincide.  -> (ctx, [weighted loss dict per chunk]); `backward_fused(ctx, scales)` differentiates sum_i scales[i] * sum(losses[i]).
        Dropout draws come from the same stateless per-site generator, over the fused batch's shapes (another stream than two half-size passes).
        Two halves: `forward_fused_begin` reads the images only (the labels -- the teacher's pseudo labels, produced meanwhile on another
        stream -- are first read by `forward_fused_finish`)."""
        return self.forward_fused_finish(self.forward_fused_begin(parts), parts)

    def forward_fused_begin(self, parts: List[List[Dict]]) -> Ctx:
        assert self.training
        images = [b["image"] for part in parts for b in part]
        ctr, sizes, mask = self._trunk(images, save=True)
        logits, boxes = self.transformer.forward([ctr.cs[1], ctr.cs[2], ctr.cs[3]], mask, record=True, feats_need_grad=True)
        if not self.aux_loss:
            logits, boxes = logits[-1:], boxes[-1:]
        c = Ctx()
        c.trunk, c.sizes, c.logits, c.boxes = ctr, sizes, logits, boxes
        self._last = None                                         # (the tape belongs to this pass: an older holder's backward must raise)
        self._last_fused = c
        return c

    def forward_fused_finish(self, c: Ctx, parts: List[List[Dict]]):
        per, lo = [], 0
        for part in parts:
            hi = lo + len(part)
            targets = self._targets([b["instances"] for b in part], c.sizes[lo:hi])
            self._check_labels(targets)
            losses, gl, gb = self.criterion(c.logits[:, lo:hi].contiguous(), c.boxes[:, lo:hi].contiguous(), targets)
            per.append(dict(lo=lo, hi=hi, losses=losses, gl=gl, gb=gb))
            lo = hi
        c.parts = per
        return c, [p_["losses"] for p_ in per]

    def backward_fused(self, c: Ctx, scales: List[float]):
        if self.__dict__.get("_last_fused") is not c:
            raise RuntimeError("DeformableDETR: backward of an earlier fused forward")
        T = self.transformer
        gl = T._out[0].new_zeros(T._out[0].shape).view(T.nd, -1, *c.parts[0]["gl"].shape[2:])
        gb = T._out[1].new_zeros(T._out[1].shape).view(T.nd, -1, *c.parts[0]["gb"].shape[2:])
        for p_, s in zip(c.parts, scales):                        # (AUX_LOSS off: only the last layer's slot is filled)
            ld = p_["gl"].shape[0]
            gl[T.nd - ld:, p_["lo"]:p_["hi"]] = p_["gl"] * float(s)
            gb[T.nd - ld:, p_["lo"]:p_["hi"]] = p_["gb"] * float(s)
        self._last_fused = None
        gfeats = T.backward(gl, gb)
        self.bengine.trunk_backward(c.trunk, {3: gfeats[0], 4: gfeats[1], 5: gfeats[2]})

    def inference(self, batched_inputs: List[Dict], do_postprocess: bool = False, pl_thresh: float = 2.0):
        """-> list[Instances] (top-100 detections of the last decoder layer, absolute xyxy in network-input pixels); the detections scoring
        above pl_thresh are also left on the device as this inference's pseudo labels (`_last_inference.pseudo`)"""
        assert not do_postprocess, "aldi_amd keeps detections in network-input space (the hot path calls do_postprocess=False)"
        with torch.no_grad():
            images = [b["image"] for b in batched_inputs]
            ctr, sizes, mask = self._trunk(images, save=False)
            logits, boxes = self.transformer.forward([ctr.cs[1], ctr.cs[2], ctr.cs[3]], mask, record=False)
            lg, bx = logits[-1], boxes[-1]
            B, Nq, K = lg.shape
            prob = torch.sigmoid(lg).view(B, -1)                   # (B x Nq x K scores: glue on a few thousand numbers)
            k = min(100, Nq * K)
            scores, idx = torch.topk(prob, k, dim=1)
            q, labels = torch.div(idx, K, rounding_mode="floor"), idx % K
            b = torch.gather(bx, 1, q[..., None].expand(-1, -1, 4))
            scale = torch.tensor([[w, h, w, h] for (h, w) in sizes], dtype=torch.float32, device=self.device)[:, None]
            xyxy = torch.cat([b[..., :2] - 0.5 * b[..., 2:], b[..., :2] + 0.5 * b[..., 2:]], -1) * scale
            keep = scores > pl_thresh
            cnt = keep.sum(1).to(torch.int32)
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)      # kept ones first, score order preserved
            pb = torch.zeros((B, GMAX_PL, 4), dtype=torch.float32, device=self.device)
            pc = torch.zeros((B, GMAX_PL), dtype=torch.int32, device=self.device)
            ps = torch.zeros((B, GMAX_PL), dtype=torch.float32, device=self.device)
            pb[:, :k] = torch.gather(xyxy, 1, order[..., None].expand(-1, -1, 4))
            pc[:, :k] = torch.gather(labels, 1, order).to(torch.int32)
            ps[:, :k] = torch.gather(scores, 1, order)
            c = Ctx()
            c.sizes, c.pseudo = sizes, {"boxes": pb, "classes": pc, "scores": ps, "count": cnt}
            self._last_inference = c
            out = []
            for i in range(B):
                inst = Instances(sizes[i])
                inst.pred_boxes = Boxes(xyxy[i])
                inst.scores = scores[i]
                inst.pred_classes = labels[i].to(torch.int64)
                out.append(inst)
            return out
