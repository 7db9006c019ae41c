"""CPU restatement (plain torch, fp32/fp64) of the ViTDet-B detector the reference trains in BASELINE cfg 4
(configs/Base-RCNN-VitDetB.yaml; aldi/backbone.py:21-43 builds it from detectron2's model_zoo `mask_rcnn_vitdet.py`).

TEST INFRASTRUCTURE ONLY -- like the rest of oracle/: imported by tests/, never by the product path.

detectron2 is not vendored in /root/reference, so the modules below restate its published algorithm:
  detectron2/modeling/backbone/vit.py   (PatchEmbed, Block, Attention, ViT, SimpleFeaturePyramid)
  detectron2/modeling/backbone/utils.py (window_partition / unpartition, get_rel_pos, add_decomposed_rel_pos, get_abs_pos)
  detectron2/modeling/proposal_generator/rpn.py  StandardRPNHead with conv_dims = [-1, -1]
  detectron2/modeling/roi_heads/box_head.py      FastRCNNConvFCHead (NUM_CONV 4, NORM "LN", NUM_FC 1)
Pinned: tests/test_vit_oracle_cpu.py checks `vit_forward` against transformers' VitDetModel (an independent implementation
of the same trunk, present in the image) on random weights; the SimpleFeaturePyramid / heads are a few torch.nn.functional
calls each and stay "parity unpinned" against detectron2 itself.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Optional

import torch
import torch.nn.functional as F

VIT_B = dict(embed=768, depth=12, heads=12, patch=16, window=14, global_blocks=(2, 5, 8, 11), pretrain_grid=14, rel_input=64, ln_eps=1e-6)


def abs_pos(pos_embed: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """pos_embed (1, G*G+1, E) with a cls slot -> (1, gh, gw, E); bicubic when the grid differs [utils.get_abs_pos]"""
    grid = pos_embed[:, 1:]
    G = int(math.isqrt(grid.shape[1]))
    assert G * G == grid.shape[1]
    if (G, G) == (gh, gw):
        return grid.reshape(1, gh, gw, -1)
    r = F.interpolate(grid.reshape(1, G, G, -1).permute(0, 3, 1, 2), size=(gh, gw), mode="bicubic", align_corners=False)
    return r.permute(0, 2, 3, 1)


def rel_table(table: torch.Tensor, n: int) -> torch.Tensor:
    """(L, hd) learned table -> (n, n, hd) with entry [q, k] = table'[q - k + n - 1], table' = linear resize to 2n-1 [utils.get_rel_pos,
    q_size == k_size]"""
    need = 2 * n - 1
    if table.shape[0] != need:
        table = F.interpolate(table.t()[None], size=need, mode="linear")[0].t()
    idx = torch.arange(n)[:, None] - torch.arange(n)[None, :] + (n - 1)
    return table[idx]


def attention(x, sd, p, heads: int):
    """x (B, H, W, E) -> (B, H, W, E): qkv, scaled dot product + decomposed relative position bias, softmax, proj"""
    B, H, W, E = x.shape
    hd = E // heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(1, 2)
    Rh, Rw = rel_table(sd[p + "rel_pos_h"], H), rel_table(sd[p + "rel_pos_w"], W)          # (H, H, hd), (W, W, hd)
    q4 = q.reshape(B * heads, H, W, hd)
    bias_h = torch.einsum("bhwc,hkc->bhwk", q4, Rh)                                        # unscaled q, as in add_decomposed_rel_pos
    bias_w = torch.einsum("bhwc,wkc->bhwk", q4, Rw)
    attn = (attn.view(-1, H, W, H, W) + bias_h[..., :, None] + bias_w[..., None, :]).view(-1, H * W, H * W)
    out = (attn.softmax(-1) @ v).view(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, E)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def to_windows(x, ws: int):
    B, H, W, E = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, E).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, E)
    return x, (Hp, Wp)


def from_windows(w, ws: int, padded, size):
    Hp, Wp = padded
    H, W = size
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W]


def block(x, sd, p, vc, windowed: bool, scales=None):
    """pre-norm transformer block; window attention pads AFTER norm1 (zeros take part as keys); `scales` = the two per-sample
    stochastic-depth multipliers (None in eval)"""
    E = x.shape[-1]
    y = F.layer_norm(x, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], vc["ln_eps"])
    if windowed:
        H, W = y.shape[1:3]
        y, padded = to_windows(y, vc["window"])
    y = attention(y, sd, p + "attn.", vc["heads"])
    if windowed:
        y = from_windows(y, vc["window"], padded, (H, W))
    if scales is not None:
        y = y * scales[0].view(-1, 1, 1, 1).to(y.dtype)
    x = x + y
    y = F.layer_norm(x, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], vc["ln_eps"])
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if scales is not None:
        y = y * scales[1].view(-1, 1, 1, 1).to(y.dtype)
    return x + y


def vit_forward(vc: dict, sd, x, drop_scales: Optional[torch.Tensor] = None, prefix: str = "backbone.net."):
    """normalised images (N, 3, H, W) -> last feature map (N, E, H/16, W/16)  [ViT.forward / aldi/backbone.py:21-34]"""
    p = prefix
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=vc["patch"]).permute(0, 2, 3, 1)
    t = t + abs_pos(sd[p + "pos_embed"], t.shape[1], t.shape[2])
    for i in range(vc["depth"]):
        t = block(t, sd, f"{p}blocks.{i}.", vc, windowed=i not in vc["global_blocks"], scales=None if drop_scales is None else drop_scales[i])
    return t.permute(0, 3, 1, 2)


def chan_ln(x, w, b, eps=1e-6):
    """detectron2 layers/batch_norm.py LayerNorm: over the channel dim of an NCHW map"""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def simple_feature_pyramid(sd, x) -> "OrderedDict[str, torch.Tensor]":
    """SimpleFeaturePyramid(scale_factors=(4, 2, 1, 0.5), norm="LN") + LastLevelMaxPool"""
    q = "backbone."

    def conv_ln(t, name, k):
        return chan_ln(F.conv2d(t, sd[name + ".weight"], padding=k // 2), sd[name + ".norm.weight"], sd[name + ".norm.bias"])
    t = F.conv_transpose2d(x, sd[q + "simfp_2.0.weight"], sd[q + "simfp_2.0.bias"], stride=2)
    t = F.gelu(chan_ln(t, sd[q + "simfp_2.1.weight"], sd[q + "simfp_2.1.bias"]))
    t = F.conv_transpose2d(t, sd[q + "simfp_2.3.weight"], sd[q + "simfp_2.3.bias"], stride=2)
    out = OrderedDict()
    out["p2"] = conv_ln(conv_ln(t, q + "simfp_2.4", 1), q + "simfp_2.5", 3)
    t = F.conv_transpose2d(x, sd[q + "simfp_3.0.weight"], sd[q + "simfp_3.0.bias"], stride=2)
    out["p3"] = conv_ln(conv_ln(t, q + "simfp_3.1", 1), q + "simfp_3.2", 3)
    out["p4"] = conv_ln(conv_ln(x, q + "simfp_4.0", 1), q + "simfp_4.1", 3)
    out["p5"] = conv_ln(conv_ln(F.max_pool2d(x, 2, 2), q + "simfp_5.1", 1), q + "simfp_5.2", 3)
    out["p6"] = F.max_pool2d(out["p5"], kernel_size=1, stride=2)
    return out


def make_backbone(vc: dict, drop_scales: Optional[torch.Tensor] = None):
    def backbone(cfg, sd, x):
        return simple_feature_pyramid(sd, vit_forward(vc, sd, x, drop_scales))
    return backbone


def _same(t):
    return t


def rpn_head(cfg, sd, feats: List[torch.Tensor], rnd=_same):
    """StandardRPNHead with conv_dims [-1, -1]: two 3x3 conv + ReLU, then the 1x1 objectness / delta predictors.
    `rnd` (tests only) is applied to every stored activation: passing a bf16 round trip reproduces the storage precision of the
    HIP path, so that ReLU masks agree and gradients can be compared element-wise."""
    p = "proposal_generator.rpn_head."
    logits, deltas = [], []
    for x in feats:
        t = rnd(F.relu(F.conv2d(x, sd[p + "conv.conv0.weight"], sd[p + "conv.conv0.bias"], 1, 1)))
        t = rnd(F.relu(F.conv2d(t, sd[p + "conv.conv1.weight"], sd[p + "conv.conv1.bias"], 1, 1)))
        logits.append(F.conv2d(t, sd[p + "objectness_logits.weight"], sd[p + "objectness_logits.bias"]))
        deltas.append(F.conv2d(t, sd[p + "anchor_deltas.weight"], sd[p + "anchor_deltas.bias"]))
    return logits, deltas


def box_head(sd, pooled, num_conv: int = 4, rnd=_same):
    """FastRCNNConvFCHead: num_conv x (conv3x3 no bias, LN, ReLU), flatten (C, 7, 7), fc1 + ReLU  (`rnd`: see rpn_head)"""
    p = "roi_heads.box_head."
    x = pooled
    for i in range(1, num_conv + 1):
        y = rnd(F.conv2d(x, sd[f"{p}conv{i}.weight"], padding=1))
        x = rnd(F.relu(chan_ln(y, sd[f"{p}conv{i}.norm.weight"], sd[f"{p}conv{i}.norm.bias"])))
    return rnd(F.relu(F.linear(x.flatten(1), sd[p + "fc1.weight"], sd[p + "fc1.bias"])))


def arch(vc: dict, drop_scales: Optional[torch.Tensor] = None) -> dict:
    """the callables d2_rcnn.forward_train / inference swap in for the ViTDet detector"""
    return dict(backbone=make_backbone(vc, drop_scales), rpn_head=rpn_head, box_head=box_head)


def adamw_step(params, grads, m, v, step: int, lr: float, wd: float, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.AdamW, one tensor: returns nothing, updates in place (restated for the optimizer parity test)"""
    b1, b2 = betas
    params.mul_(1 - lr * wd)
    m.mul_(b1).add_(grads, alpha=1 - b1)
    v.mul_(b2).addcmul_(grads, grads, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
    params.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
