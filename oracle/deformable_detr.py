"""CPU restatement (plain torch, fp32 / fp64) of the Deformable-DETR detector behind BASELINE configs[4]
(`configs/Base-DETR.yaml`: META_ARCHITECTURE "DeformableDETR", 4 feature levels, 300 queries, 6 + 6 layers, 8 heads x 4 points,
no box refinement, not two-stage, focal / L1 / GIoU set loss with auxiliary decoder losses).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference reaches this model through the ABSENT submodule
`aldi/detr/libs/DeformableDETRDetectron2` (`.gitmodules:4-6`, empty directory; `aldi/detr/distill.py:3`, `aldi/detr/align.py:3`), so
the algorithm is restated from its published description (Zhu et al., "Deformable DETR", ICLR 2021, and the authors' public
implementation's semantics: parameter names below are that implementation's).  PINNED against transformers'
`DeformableDetrForObjectDetection` (present in the image): `tests/golden/make_detr_golden.py` runs it on seeded inputs with random
weights and stores inputs, weights and outputs; `tests/test_oracle_detr_cpu.py` holds this file to those vectors (and, when
transformers is importable, to a live run).  What transformers does not pin is said where it applies.

Everything after the backbone is here: input projections (conv + GroupNorm(32)), sine position embedding, the deformable encoder,
the decoder (self attention + deformable cross attention), the heads, Hungarian matching and the losses.  The multi-scale deformable
attention sampling itself is `oracle/ms_deform_attn.py`.

Parameters: a dict name -> tensor with the original implementation's names:
  input_proj.{l}.0.{weight,bias} (conv), input_proj.{l}.1.{weight,bias} (GroupNorm), transformer.level_embed [L, d],
  transformer.encoder.layers.{i}.self_attn.{sampling_offsets,attention_weights,value_proj,output_proj}.{weight,bias},
  transformer.encoder.layers.{i}.{norm1,norm2}.{weight,bias}, transformer.encoder.layers.{i}.{linear1,linear2}.{weight,bias},
  transformer.decoder.layers.{i}.self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias} (nn.MultiheadAttention),
  transformer.decoder.layers.{i}.cross_attn.* (as the encoder's self_attn), transformer.decoder.layers.{i}.{norm1,norm2,norm3}.*,
  transformer.decoder.layers.{i}.{linear1,linear2}.*, transformer.reference_points.{weight,bias}, query_embed.weight [Nq, 2 d],
  class_embed.{weight,bias}, bbox_embed.layers.{0,1,2}.{weight,bias}."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .ms_deform_attn import ms_deform_attn


# ------------------------------------------------------------------------------------------------------------ pieces
def sine_position_embedding(mask: torch.Tensor, d_model: int, temperature: float = 10000.0, scale: float = 2 * math.pi) -> torch.Tensor:
    """mask (B, H, W) bool, True = padding -> (B, d_model, H, W).  normalize=True form (POSITION_EMBEDDING 'sine',
    POSITION_EMBEDDING_SCALE 2 pi, configs/Base-DETR.yaml:11-12): cumulative counts of valid pixels, normalised by the last one."""
    npf = d_model // 2
    not_mask = (~mask).to(torch.float32)
    y_embed = not_mask.cumsum(1)
    x_embed = not_mask.cumsum(2)
    eps = 1e-6
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def _linear(p, name, x):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _layer_norm(p, name, x):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], 1e-5)


def deformable_attention(p: Dict[str, torch.Tensor], pre: str, query, reference_points, value_in, shapes: Sequence[Tuple[int, int]], padding_mask,
                         n_heads: int, n_points: int):
    """MSDeformAttn.forward: query (B, Q, d), reference_points (B, Q, L, 2) in [0, 1] of the VALID area of each level, value_in
    (B, S, d) the flattened maps, padding_mask (B, S) True = padding."""
    B, Q, d = query.shape
    L = len(shapes)
    value = _linear(p, pre + ".value_proj", value_in)
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(B, -1, n_heads, d // n_heads)
    off = _linear(p, pre + ".sampling_offsets", query).view(B, Q, n_heads, L, n_points, 2)
    aw = _linear(p, pre + ".attention_weights", query).view(B, Q, n_heads, L * n_points)
    aw = F.softmax(aw, -1).view(B, Q, n_heads, L, n_points)
    normalizer = torch.tensor([[w, h] for (h, w) in shapes], dtype=query.dtype)                   # (L, 2): offsets are in pixels of their level
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = ms_deform_attn(value, list(shapes), loc, aw)
    return _linear(p, pre + ".output_proj", out)


def _keep(drop, name, t):
    """dropout hook of the restatement: `drop(site name, tensor) -> tensor` (None = evaluation mode / p = 0).  Sites carry the authors'
    module names: <layer>.dropout1..4 and <layer>.self_attn.attn (the attention probabilities inside nn.MultiheadAttention)."""
    return t if drop is None else drop(name, t)


def multihead_self_attention(p, pre: str, q_in, k_in, v_in, n_heads: int, drop=None):
    """nn.MultiheadAttention (batch-first here) with its packed in_proj: q, k from (target + query position), v from target"""
    B, Q, d = q_in.shape
    w, b = p[pre + ".in_proj_weight"], p[pre + ".in_proj_bias"]
    q = F.linear(q_in, w[:d], b[:d]).view(B, Q, n_heads, d // n_heads).transpose(1, 2)
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d]).view(B, Q, n_heads, d // n_heads).transpose(1, 2)
    v = F.linear(v_in, w[2 * d:], b[2 * d:]).view(B, Q, n_heads, d // n_heads).transpose(1, 2)
    att = _keep(drop, pre + ".attn", F.softmax((q * (d // n_heads) ** -0.5) @ k.transpose(-1, -2), -1))
    out = (att @ v).transpose(1, 2).reshape(B, Q, d)
    return F.linear(out, p[pre + ".out_proj.weight"], p[pre + ".out_proj.bias"])


# ------------------------------------------------------------------------------------------------------------ the model after the backbone
def prepare_levels(p, feats: List[torch.Tensor], image_mask: torch.Tensor, d_model: int, num_levels: int):
    """input_proj on the backbone maps (C3..C5; further levels: a stride-2 3x3 conv on the LAST backbone map, then on the previous extra
    level), GroupNorm(32), position embeddings + level embedding, flattening.  image_mask (B, H, W) bool, True = padding of the batched
    images; every level's mask is its nearest-neighbour resampling (as the backbone wrapper and the extra levels do it).
    -> src (B, S, d), pos (B, S, d), mask (B, S), shapes, valid_ratios (B, L, 2) = (valid width, valid height) fractions"""
    srcs, ms = [], []
    for l in range(num_levels):
        if l < len(feats):
            x = F.conv2d(feats[l], p[f"input_proj.{l}.0.weight"], p[f"input_proj.{l}.0.bias"])
        else:
            x = F.conv2d(feats[-1] if l == len(feats) else srcs[-1], p[f"input_proj.{l}.0.weight"], p[f"input_proj.{l}.0.bias"], stride=2, padding=1)
        x = F.group_norm(x, 32, p[f"input_proj.{l}.1.weight"], p[f"input_proj.{l}.1.bias"], 1e-5)
        srcs.append(x)
        ms.append(F.interpolate(image_mask[None].float(), size=x.shape[-2:]).to(torch.bool)[0])
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    pos = [sine_position_embedding(m, d_model).to(srcs[0].dtype) + p["transformer.level_embed"][l].view(1, -1, 1, 1) for l, m in enumerate(ms)]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([q.flatten(2).transpose(1, 2) for q in pos], 1)
    mask = torch.cat([m.flatten(1) for m in ms], 1)
    vr = []
    for m in ms:
        H, W = m.shape[1:]
        vh = (~m[:, :, 0]).sum(1).to(src.dtype) / H
        vw = (~m[:, 0, :]).sum(1).to(src.dtype) / W
        vr.append(torch.stack([vw, vh], -1))
    return src, pos, mask, shapes, torch.stack(vr, 1)


def encoder_reference_points(shapes, valid_ratios):
    """pixel centres of every level, in fractions of that level's VALID area, re-expressed for every level: (B, S, L, 2)"""
    ref = []
    for l, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=valid_ratios.dtype), torch.linspace(0.5, W - 0.5, W, dtype=valid_ratios.dtype), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, l, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, l, 0] * W)
        ref.append(torch.stack((rx, ry), -1))
    ref = torch.cat(ref, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def encoder(p, src, pos, mask, shapes, valid_ratios, n_layers: int, n_heads: int, n_points: int, drop=None):
    ref = encoder_reference_points(shapes, valid_ratios)
    x = src
    for i in range(n_layers):
        pre = f"transformer.encoder.layers.{i}"
        x = _layer_norm(p, pre + ".norm1", x + _keep(drop, pre + ".dropout1", deformable_attention(p, pre + ".self_attn", x + pos, ref, x, shapes, mask, n_heads, n_points)))
        h = _keep(drop, pre + ".dropout2", F.relu(_linear(p, pre + ".linear1", x)))
        x = _layer_norm(p, pre + ".norm2", x + _keep(drop, pre + ".dropout3", _linear(p, pre + ".linear2", h)))
    return x


def decoder(p, memory, mask, shapes, valid_ratios, n_layers: int, n_heads: int, n_points: int, drop=None):
    """-> hidden states of every layer (n_layers, B, Nq, d) and the (fixed: no box refinement) reference points (B, Nq, 2)"""
    B = memory.shape[0]
    qe = p["query_embed.weight"]
    d = qe.shape[1] // 2
    query_pos, tgt = qe[:, :d][None].expand(B, -1, -1), qe[:, d:][None].expand(B, -1, -1)
    reference = torch.sigmoid(_linear(p, "transformer.reference_points", query_pos))
    ref_in = reference[:, :, None] * valid_ratios[:, None]
    hs = []
    for i in range(n_layers):
        pre = f"transformer.decoder.layers.{i}"
        q = tgt + query_pos
        tgt = _layer_norm(p, pre + ".norm2", tgt + _keep(drop, pre + ".dropout2", multihead_self_attention(p, pre + ".self_attn", q, q, tgt, n_heads, drop)))
        tgt = _layer_norm(p, pre + ".norm1", tgt + _keep(drop, pre + ".dropout1", deformable_attention(p, pre + ".cross_attn", tgt + query_pos, ref_in, memory, shapes, mask, n_heads, n_points)))
        h = _keep(drop, pre + ".dropout3", F.relu(_linear(p, pre + ".linear1", tgt)))
        tgt = _layer_norm(p, pre + ".norm3", tgt + _keep(drop, pre + ".dropout4", _linear(p, pre + ".linear2", h)))
        hs.append(tgt)
    return torch.stack(hs), reference


def heads(p, hs, reference):
    """class logits and boxes (cx, cy, w, h in [0, 1]) of every decoder layer; the box head's first two outputs are offsets from the
    reference point in logit space"""
    logits = _linear(p, "class_embed", hs)
    t = _linear(p, "bbox_embed.layers.2", F.relu(_linear(p, "bbox_embed.layers.1", F.relu(_linear(p, "bbox_embed.layers.0", hs)))))
    t = torch.cat([t[..., :2] + inverse_sigmoid(reference)[None], t[..., 2:]], -1)
    return logits, torch.sigmoid(t)


def forward(p, feats, image_mask, *, d_model=256, num_levels=4, enc_layers=6, dec_layers=6, n_heads=8, enc_points=4, dec_points=4, drop=None):
    """feats: the backbone's C3..C5 maps (B, C_l, H_l, W_l); image_mask (B, H, W) True = padding -> (logits (L, B, Nq, K), boxes (L, B, Nq, 4))"""
    src, pos, mask, shapes, vr = prepare_levels(p, feats, image_mask, d_model, num_levels)
    memory = encoder(p, src, pos, mask, shapes, vr, enc_layers, n_heads, enc_points, drop)
    hs, reference = decoder(p, memory, mask, shapes, vr, dec_layers, n_heads, dec_points, drop)
    return heads(p, hs, reference)


# ------------------------------------------------------------------------------------------------------------ set loss
def box_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def generalized_box_iou(a, b):
    """(Na, 4), (Nb, 4) xyxy -> (Na, Nb)"""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt, rb = torch.max(a[:, None, :2], b[None, :, :2]), torch.min(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    union = area_a[:, None] + area_b[None] - inter
    iou = inter / union
    lt2, rb2 = torch.min(a[:, None, :2], b[None, :, :2]), torch.max(a[:, None, 2:], b[None, :, 2:])
    hull = (rb2 - lt2).clamp(min=0).prod(-1)
    return iou - (hull - union) / hull


def hungarian_match(logits, boxes, targets, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, alpha=0.25, gamma=2.0):
    """logits (B, Nq, K), boxes (B, Nq, 4), targets [{"labels": (n,), "boxes": (n, 4)}] -> [(query idx, target idx)] per image
    (MATCHER.SET_COST_* of configs/Base-DETR.yaml:36-39; focal-style class cost)"""
    from scipy.optimize import linear_sum_assignment
    out = []
    for b, t in enumerate(targets):
        if len(t["labels"]) == 0:
            out.append((torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)))
            continue
        prob = torch.sigmoid(logits[b])
        neg = (1 - alpha) * prob ** gamma * -(1 - prob + 1e-8).log()
        pos = alpha * (1 - prob) ** gamma * -(prob + 1e-8).log()
        c_cls = pos[:, t["labels"]] - neg[:, t["labels"]]
        c_box = torch.cdist(boxes[b], t["boxes"], p=1)
        c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes[b]), box_cxcywh_to_xyxy(t["boxes"]))
        C = cost_bbox * c_box + cost_class * c_cls + cost_giou * c_giou
        i, j = linear_sum_assignment(C.detach().cpu().numpy())
        out.append((torch.as_tensor(i, dtype=torch.long), torch.as_tensor(j, dtype=torch.long)))
    return out


def sigmoid_focal_loss(logits, onehot, num_boxes, alpha=0.25, gamma=2.0):
    prob = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    p_t = prob * onehot + (1 - prob) * (1 - onehot)
    loss = ce * (1 - p_t) ** gamma
    loss = (alpha * onehot + (1 - alpha) * (1 - onehot)) * loss
    return loss.mean(1).sum() / num_boxes


def set_losses(logits, boxes, targets, indices=None, num_boxes=None, alpha=0.25):
    """one decoder layer's losses: focal classification (x Nq, as the authors do), L1 and GIoU on the matched pairs, all / num_boxes"""
    if indices is None:
        indices = hungarian_match(logits, boxes, targets, alpha=alpha)
    if num_boxes is None:
        num_boxes = max(float(sum(len(t["labels"]) for t in targets)), 1.0)
    B, Nq, K = logits.shape
    onehot = torch.zeros_like(logits)
    src_b, tgt_b = [], []
    for b, (i, j) in enumerate(indices):
        onehot[b, i, targets[b]["labels"][j]] = 1.0
        src_b.append(boxes[b, i])
        tgt_b.append(targets[b]["boxes"][j])
    src_b, tgt_b = torch.cat(src_b), torch.cat(tgt_b)
    loss_ce = sigmoid_focal_loss(logits, onehot, num_boxes, alpha=alpha) * Nq
    loss_bbox = (src_b - tgt_b).abs().sum() / num_boxes
    loss_giou = (1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src_b), box_cxcywh_to_xyxy(tgt_b)))).sum() / num_boxes if len(src_b) else src_b.sum()
    return {"loss_ce": loss_ce, "loss_bbox": loss_bbox, "loss_giou": loss_giou}, indices


def criterion(all_logits, all_boxes, targets, weights=(2.0, 5.0, 2.0), alpha=0.25):
    """LOSS section of configs/Base-DETR.yaml:27-35: the last layer's losses + the same for every earlier decoder layer (AUX_LOSS), each
    with its own matching; -> (dict, weighted total)"""
    out, total = {}, 0.0
    L = all_logits.shape[0]
    for l in range(L):
        d, _ = set_losses(all_logits[l], all_boxes[l], targets, alpha=alpha)
        suffix = "" if l == L - 1 else f"_{l}"
        for (k, v), w in zip(d.items(), weights):
            out[k + suffix] = v
            total = total + w * v
    return out, total


def post_process(logits, boxes, image_sizes, topk=100):
    """the detector's inference output (what the teacher's pseudo-labels are thresholded from): top-k over all (query, class) sigmoid
    scores of the LAST decoder layer, boxes in absolute xyxy pixels of each image"""
    B, Nq, K = logits.shape
    prob = torch.sigmoid(logits).view(B, -1)
    scores, idx = torch.topk(prob, min(topk, Nq * K), dim=1)
    q, labels = torch.div(idx, K, rounding_mode="floor"), idx % K
    b = box_cxcywh_to_xyxy(torch.gather(boxes, 1, q[..., None].expand(-1, -1, 4)))
    scale = torch.tensor([[w, h, w, h] for (h, w) in image_sizes], dtype=boxes.dtype)[:, None]
    return scores, labels, b * scale
