"""CPU restatement (plain torch) of the ConvNeXt-FPN trunk the reference trains in configs/Base-RCNN-ConvNeXt-FPN.yaml /
cityscapes/ALDI-Best-ConvNeXt-Cityscapes.yaml: `ConvNeXt.forward_features` (aldi/backbone.py:226-331) + detectron2 FPN
(fuse "sum", no norm, LastLevelMaxPool; aldi/backbone.py:373-392).

TEST INFRASTRUCTURE ONLY.  The ConvNeXt part is **pinned** against the reference's own class through golden g10
(tests/test_convnext_oracle_cpu.py); the FPN is the same three lines as in oracle/d2_rcnn.py (detectron2, unpinned)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn.functional as F

from .d2_vitdet import chan_ln


def block(x, sd, p, scale: Optional[torch.Tensor] = None, eps: float = 1e-6):
    """ConvNextBlock (aldi/backbone.py:189-224): depthwise 7x7, LayerNorm over channels, Linear 4x, GELU, Linear, layer scale gamma,
    per-sample stochastic-depth multiplier, residual"""
    C = x.shape[1]
    y = F.conv2d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps)
    y = F.linear(F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])), sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    y = (sd[p + "gamma"] * y).permute(0, 3, 1, 2)
    if scale is not None:
        y = y * scale.view(-1, 1, 1, 1).to(y.dtype)
    return x + y


def convnext_forward(cc: dict, sd, x, drop_scales: Optional[torch.Tensor] = None, prefix: str = "backbone.bottom_up."):
    """normalised images (N, 3, H, W) -> [stage 0..3 outputs after norm{i}] (strides 4, 8, 16, 32)"""
    p = prefix
    outs, k = [], 0
    for i in range(4):
        d = f"{p}downsample_layers.{i}."
        if i == 0:
            x = chan_ln(F.conv2d(x, sd[d + "0.weight"], sd[d + "0.bias"], stride=4), sd[d + "1.weight"], sd[d + "1.bias"])
        else:
            x = F.conv2d(chan_ln(x, sd[d + "0.weight"], sd[d + "0.bias"]), sd[d + "1.weight"], sd[d + "1.bias"], stride=2)
        for j in range(cc["depths"][i]):
            x = block(x, sd, f"{p}stages.{i}.{j}.", None if drop_scales is None else drop_scales[k])
            k += 1
        outs.append(chan_ln(x, sd[f"{p}norm{i}.weight"], sd[f"{p}norm{i}.bias"]))
    return outs


def fpn(sd, cs) -> "OrderedDict[str, torch.Tensor]":
    feats = OrderedDict()
    prev = F.conv2d(cs[3], sd["backbone.fpn_lateral5.weight"], sd["backbone.fpn_lateral5.bias"])
    outs = {5: F.conv2d(prev, sd["backbone.fpn_output5.weight"], sd["backbone.fpn_output5.bias"], 1, 1)}
    for lvl in (4, 3, 2):
        top = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = F.conv2d(cs[lvl - 2], sd[f"backbone.fpn_lateral{lvl}.weight"], sd[f"backbone.fpn_lateral{lvl}.bias"]) + top
        outs[lvl] = F.conv2d(prev, sd[f"backbone.fpn_output{lvl}.weight"], sd[f"backbone.fpn_output{lvl}.bias"], 1, 1)
    for lvl in (2, 3, 4, 5):
        feats[f"p{lvl}"] = outs[lvl]
    feats["p6"] = F.max_pool2d(outs[5], kernel_size=1, stride=2, padding=0)
    return feats


def arch(cc: dict, drop_scales: Optional[torch.Tensor] = None) -> dict:
    """callables for d2_rcnn.forward_train(arch=...): only the backbone differs from the R50 detector (standard RPN / box heads)"""
    def backbone(cfg, sd, x):
        return fpn(sd, convnext_forward(cc, sd, x, drop_scales))
    return dict(backbone=backbone)
