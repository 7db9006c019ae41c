"""Architecture table of the R50-FPN Faster R-CNN the ALDI step runs on, and the mapping between
Detectron2-style state_dict keys (what the reference's checkpoints / EMA / optimizer see:
``backbone.bottom_up.res2.0.conv1.weight`` ...; reference aldi/checkpoint.py:22-24, aldi/ema.py:19-50)
and the engine's flat parameter buffer (NHWC-kernel layout, packed heads).

Hyper-parameters follow reference configs/detectron2/Base-RCNN-FPN.yaml:1-31 and
configs/Base-RCNN-FPN.yaml:1-25 plus the Detectron2 defaults listed in SURVEY.md Appendix A.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_MID = (64, 128, 256, 512)
STAGE_OUT = (256, 512, 1024, 2048)
FPN_C = 256
NUM_ANCHORS = 3
POOL = 7
FC_DIM = 1024


def pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class Conv:
    name: str            # D2 module path (without .weight)
    cin: int
    cout: int
    k: int               # 0 = Linear
    stride: int = 1
    pad: int = 0
    bn: bool = False
    bias: bool = False
    trainable: bool = True


def d2_convs(num_classes: int) -> "OrderedDict[str, Conv]":
    """Every conv / linear of GeneralizedRCNN(R50-FPN) in state_dict order."""
    s: "OrderedDict[str, Conv]" = OrderedDict()
    bu = "backbone.bottom_up."
    s[bu + "stem.conv1"] = Conv(bu + "stem.conv1", 3, 64, 7, 2, 3, bn=True, trainable=False)
    cin = 64
    for si, (nb, mid, out) in enumerate(zip(STAGE_BLOCKS, STAGE_MID, STAGE_OUT)):
        for b in range(nb):
            stride = 2 if (b == 0 and si > 0) else 1
            p = f"{bu}res{si + 2}.{b}."
            tr = si > 0                                   # FREEZE_AT=2: stem + res2 frozen
            if b == 0:
                s[p + "shortcut"] = Conv(p + "shortcut", cin, out, 1, stride, 0, bn=True, trainable=tr)
            s[p + "conv1"] = Conv(p + "conv1", cin, mid, 1, stride, 0, bn=True, trainable=tr)   # STRIDE_IN_1X1
            s[p + "conv2"] = Conv(p + "conv2", mid, mid, 3, 1, 1, bn=True, trainable=tr)
            s[p + "conv3"] = Conv(p + "conv3", mid, out, 1, 1, 0, bn=True, trainable=tr)
            cin = out
    for lvl, c in zip((2, 3, 4, 5), STAGE_OUT):
        s[f"backbone.fpn_lateral{lvl}"] = Conv(f"backbone.fpn_lateral{lvl}", c, FPN_C, 1, bias=True)
        s[f"backbone.fpn_output{lvl}"] = Conv(f"backbone.fpn_output{lvl}", FPN_C, FPN_C, 3, 1, 1, bias=True)
    rp = "proposal_generator.rpn_head."
    s[rp + "conv"] = Conv(rp + "conv", FPN_C, FPN_C, 3, 1, 1, bias=True)
    s[rp + "objectness_logits"] = Conv(rp + "objectness_logits", FPN_C, NUM_ANCHORS, 1, bias=True)
    s[rp + "anchor_deltas"] = Conv(rp + "anchor_deltas", FPN_C, 4 * NUM_ANCHORS, 1, bias=True)
    s["roi_heads.box_head.fc1"] = Conv("roi_heads.box_head.fc1", FPN_C * POOL * POOL, FC_DIM, 0, bias=True)
    s["roi_heads.box_head.fc2"] = Conv("roi_heads.box_head.fc2", FC_DIM, FC_DIM, 0, bias=True)
    s["roi_heads.box_predictor.cls_score"] = Conv("roi_heads.box_predictor.cls_score", FC_DIM, num_classes + 1, 0, bias=True)
    s["roi_heads.box_predictor.bbox_pred"] = Conv("roi_heads.box_predictor.bbox_pred", FC_DIM, 4 * num_classes, 0, bias=True)
    return s


FPN_LEVELS = ("p2", "p3", "p4", "p5", "p6")


def da_spec(img, ins):
    """Normalised discriminator specs.  `img` / `ins`: False | True (the reference defaults, aldi/config.py:41-49) | dict with
    the DOMAIN_ADAPT.ALIGN keys in lower case: img {layer, input_dim, hidden_dims}, ins {input_dim, hidden_dims}."""
    def norm(v, dflt):
        if not v:
            return None
        d = dict(dflt)
        if isinstance(v, dict):
            d.update(v)
        d["hidden_dims"] = [int(x) for x in d["hidden_dims"]]
        return d
    i = norm(img, dict(layer="p2", input_dim=256, hidden_dims=[256]))
    if i is not None and i["layer"] not in FPN_LEVELS:
        raise ValueError(f"DOMAIN_ADAPT.ALIGN.IMG_DA_LAYER must be one of {FPN_LEVELS}, got {i['layer']}")
    return i, norm(ins, dict(input_dim=1024, hidden_dims=[1024]))


def disc_convs(img, ins) -> "OrderedDict[str, Conv]":
    """Discriminator modules added by AlignMixin (reference aldi/align.py:41-42,103-135) for ANY hidden_dims list, under the
    nn.Sequential indices of the reference: ConvDiscriminator = [Conv2d(k=3, no padding), ReLU] per hidden dim, then
    AdaptiveAvgPool2d, Flatten, Linear(., 1)  ->  convs at model.0, model.2, ..., the Linear at model.(2n+2);
    FCDiscriminator = Flatten, [Linear, ReLU] per hidden dim, Linear(., 1)  ->  model.1, model.3, ..., model.(2n+1)."""
    img, ins = da_spec(img, ins)
    s: "OrderedDict[str, Conv]" = OrderedDict()
    if img:
        prev = img["input_dim"]
        for i, d in enumerate(img["hidden_dims"]):
            s[f"img_align.model.{2 * i}"] = Conv(f"img_align.model.{2 * i}", prev, d, 3, 1, 0, bias=True)
            prev = d
        n = len(img["hidden_dims"])
        s[f"img_align.model.{2 * n + 2}"] = Conv(f"img_align.model.{2 * n + 2}", prev, 1, 0, bias=True)
    if ins:
        prev = ins["input_dim"]
        for i, d in enumerate(ins["hidden_dims"]):
            s[f"ins_align.model.{2 * i + 1}"] = Conv(f"ins_align.model.{2 * i + 1}", prev, d, 0, bias=True)
            prev = d
        n = len(ins["hidden_dims"])
        s[f"ins_align.model.{2 * n + 1}"] = Conv(f"ins_align.model.{2 * n + 1}", prev, 1, 0, bias=True)
    return s


@dataclass
class Packed:
    """One engine weight tensor [rows][KH][KW][Cin] (+ bias [rows]) built from 1..n D2 modules (rows concatenated, zero padded)."""
    name: str
    sources: List[str]
    rows: int
    k: int
    cin: int
    stride: int
    pad: int
    bn: Optional[str]          # D2 module carrying the FrozenBN buffers
    bias: bool
    trainable: bool
    fc1_permute: bool = False  # D2 flattens (C,7,7); the engine's ROIAlign output is (7,7,C)
    w_off: int = -1
    b_off: int = -1
    bn_off: int = -1           # channel offset inside the BN section

    @property
    def kk(self):
        return max(self.k, 1)

    @property
    def wshape(self):
        return (self.rows, self.kk, self.kk, self.cin)


def engine_tensors(num_classes: int, img_da=False, ins_da=False) -> "OrderedDict[str, Packed]":
    d2 = d2_convs(num_classes)
    out: "OrderedDict[str, Packed]" = OrderedDict()
    for name, c in d2.items():
        if name.endswith("objectness_logits") or name.endswith("anchor_deltas") or name.endswith("cls_score") or name.endswith("bbox_pred"):
            continue
        out[name] = Packed(name, [name], c.cout, c.k, c.cin, c.stride, c.pad, name if c.bn else None, c.bias, c.trainable,
                           fc1_permute=name.endswith("box_head.fc1"))
    rp = "proposal_generator.rpn_head."
    out["rpn_head_out"] = Packed("rpn_head_out", [rp + "objectness_logits", rp + "anchor_deltas"], pad_to(5 * NUM_ANCHORS, 16), 1, FPN_C,
                                 1, 0, None, True, True)
    bp = "roi_heads.box_predictor."
    out["box_pred"] = Packed("box_pred", [bp + "cls_score", bp + "bbox_pred"], pad_to(5 * num_classes + 1, 16), 0, FC_DIM, 1, 0, None, True, True)
    for name, c in disc_convs(img_da, ins_da).items():
        rows = c.cout if c.cout > 1 else 8
        out[name] = Packed(name, [name], rows, c.k, c.cin, c.stride, c.pad, None, True, True)
    return out


class ParamLayout:
    """Flat fp32 layout: [trainable weights+biases | frozen weights | bn_w | bn_b | bn_mean | bn_var]."""

    def __init__(self, num_classes: int, img_da=False, ins_da=False):
        self.num_classes = num_classes
        self.img_da, self.ins_da = da_spec(img_da, ins_da)
        self.d2 = d2_convs(num_classes)
        self.d2.update(disc_convs(img_da, ins_da))
        self.t = engine_tensors(num_classes, img_da, ins_da)
        off = 0
        for trainable in (True, False):
            for p in self.t.values():
                if p.trainable != trainable:
                    continue
                n = p.rows * p.kk * p.kk * p.cin
                p.w_off = off
                off += pad_to(n, 64)
                if p.bias:
                    p.b_off = off
                    off += pad_to(p.rows, 64)
            if trainable:
                self.n_train = off
        self.n_weights = off
        ch = 0
        for p in self.t.values():
            if p.bn:
                p.bn_off = ch
                ch += pad_to(p.rows, 64)
        self.bn_channels = ch
        self.bn_base = off
        self.n_total = off + 4 * ch

    # ---- state_dict <-> flat (CPU tensors) ------------------------------------------------
    def pack(self, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        flat = torch.zeros(self.n_total, dtype=torch.float32)
        for p in self.t.values():
            rows = []
            for src in p.sources:
                w = sd[src + ".weight"].detach().to(torch.float32).cpu()
                c = self.d2[src]
                if c.k == 0:
                    if p.fc1_permute:
                        w = w.view(c.cout, FPN_C, POOL, POOL).permute(0, 2, 3, 1).reshape(c.cout, 1, 1, -1)
                    else:
                        w = w.view(c.cout, 1, 1, c.cin)
                else:
                    w = w.permute(0, 2, 3, 1)
                rows.append(w.reshape(c.cout, -1))
            w = torch.cat(rows, 0)
            blk = torch.zeros(p.rows, w.shape[1])
            blk[: w.shape[0]] = w
            flat[p.w_off: p.w_off + blk.numel()] = blk.reshape(-1)
            if p.bias:
                b = torch.cat([sd[src + ".bias"].detach().to(torch.float32).cpu() for src in p.sources])
                flat[p.b_off: p.b_off + b.numel()] = b
            if p.bn:
                for i, f in enumerate(("weight", "bias", "running_mean", "running_var")):
                    v = sd[f"{p.bn}.norm.{f}"].detach().to(torch.float32).cpu()
                    o = self.bn_base + i * self.bn_channels + p.bn_off
                    flat[o: o + v.numel()] = v
                    if f == "running_var" and p.rows < pad_to(p.rows, 64):
                        flat[o + v.numel(): o + pad_to(p.rows, 64)] = 1.0
        return flat

    def unpack(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        flat = flat.detach().cpu()
        sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        tmp: Dict[str, torch.Tensor] = {}
        for p in self.t.values():
            w = flat[p.w_off: p.w_off + p.rows * p.kk * p.kk * p.cin].view(p.rows, -1)
            b = flat[p.b_off: p.b_off + p.rows] if p.bias else None
            r0 = 0
            for src in p.sources:
                c = self.d2[src]
                ws = w[r0: r0 + c.cout]
                if c.k == 0:
                    if p.fc1_permute:
                        ws = ws.view(c.cout, POOL, POOL, FPN_C).permute(0, 3, 1, 2).reshape(c.cout, -1)
                    else:
                        ws = ws.reshape(c.cout, c.cin)
                else:
                    ws = ws.view(c.cout, c.k, c.k, c.cin).permute(0, 3, 1, 2)
                tmp[src + ".weight"] = ws.contiguous().clone()
                if b is not None:
                    tmp[src + ".bias"] = b[r0: r0 + c.cout].clone()
                r0 += c.cout
            if p.bn:
                for i, f in enumerate(("weight", "bias", "running_mean", "running_var")):
                    o = self.bn_base + i * self.bn_channels + p.bn_off
                    tmp[f"{p.bn}.norm.{f}"] = flat[o: o + p.rows].clone()
        for name, c in self.d2.items():          # D2 key order
            sd[name + ".weight"] = tmp[name + ".weight"]
            if c.bias:
                sd[name + ".bias"] = tmp[name + ".bias"]
            if c.bn:
                for f in ("weight", "bias", "running_mean", "running_var"):
                    sd[f"{name}.norm.{f}"] = tmp[f"{name}.norm.{f}"]
        return sd

    def ranges(self, names) -> List[Tuple[int, int]]:
        """flat-buffer element ranges (weights + biases) of engine tensors, padded extents: the layout rounds every tensor up to 64
        elements and the padding never receives gradient, so neighbouring layers merge into ONE contiguous range and nothing is left
        between them for the gradient exchange to reduce piecemeal"""
        out = []
        for n in names:
            p = self.t[n]
            out.append((p.w_off, p.w_off + pad_to(p.rows * p.kk * p.kk * p.cin, 64)))
            if p.bias:
                out.append((p.b_off, p.b_off + pad_to(p.rows, 64)))
        return out

    def state_dict_keys(self) -> List[str]:
        keys = []
        for name, c in self.d2.items():
            keys.append(name + ".weight")
            if c.bias:
                keys.append(name + ".bias")
            if c.bn:
                keys += [f"{name}.norm.{f}" for f in ("weight", "bias", "running_mean", "running_var")]
        return keys
