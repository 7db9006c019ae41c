"""The Deformable-DETR detector AFTER its backbone on the HIP library (forward pass; BASELINE configs[4]): input
projections + GroupNorm, deformable encoder, decoder (self attention + deformable cross attention), class / box heads -- every map a
C-ABI launch (linear layers = aldi_conv_igemm with H = W = 1 in fp32, aldi_group_norm_forward, aldi_layernorm_forward,
aldi_msda_prepare + aldi_ms_deform_attn_forward, aldi_mha_small_forward, aldi_detr_box_finish).  fp32: the reference runs this detector
with AMP off (configs/Base-DETR.yaml:56-58).

The reference's own detector is an absent submodule (`aldi/detr/libs/DeformableDETRDetectron2`, .gitmodules:4-6; registered as
`DETRDistillMixin` / `DETRAlignMixin` in aldi/detr/distill.py:6-7, aldi/detr/align.py:6-7); parameter names are the authors'
(oracle/deformable_detr.py lists them) and the arithmetic is held to that oracle in tests/test_detr_gpu.py.

NOT here yet: the backward pass, the set loss on the device, the training step."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch

from .. import _lib as L
from .. import ops
from .. import vit_ops as V
from ..ops import _p, stream_ptr


def _linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, relu: bool = False, res: torch.Tensor = None) -> torch.Tensor:
    """x [T, Cin] fp32, w [Cout, Cin] -> [T, Cout] (+ res, ReLU in the epilogue)"""
    T, Cin = x.shape
    if w.shape[0] % 4:                      # (the kernels write 4 output channels at a time: the 2-output reference-point layer is padded)
        pad = 4 - w.shape[0] % 4
        y = _linear(x, torch.cat([w, w.new_zeros(pad, Cin)]), torch.cat([b, b.new_zeros(pad)]), relu=relu)
        return y[:, :w.shape[0]].contiguous() + (0 if res is None else res)
    y = ops.conv2d(x.view(T, 1, 1, Cin), w.view(w.shape[0], 1, 1, Cin), shift=b, relu=relu,
                   res=None if res is None else res.view(T, 1, 1, -1), res_mode=0 if res is None else 1)
    return y.view(T, w.shape[0])


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5) -> torch.Tensor:
    """x [N, H, W, C] fp32 -> GroupNorm(groups) over (H, W, C / groups)"""
    N, H, W_, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((N, groups), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = torch.empty(max(int(L.lib.aldi_group_norm_workspace(N, H * W_, groups)), 4), dtype=torch.uint8, device=x.device)
    L.call("aldi_group_norm_forward", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(ws), N, H * W_, C, groups, float(eps), stream_ptr())
    return y


def sine_position_embedding(mask: torch.Tensor, d_model: int, temperature: float = 10000.0, scale: float = 2 * math.pi) -> torch.Tensor:
    """(B, H, W) bool padding mask -> (B, H, W, d): an input-independent table (host arithmetic, cached by the caller)"""
    npf = d_model // 2
    nm = (~mask).to(torch.float32)
    y, x = nm.cumsum(1), nm.cumsum(2)
    y = (y - 0.5) / (y[:, -1:, :] + 1e-6) * scale
    x = (x - 0.5) / (x[:, :, -1:] + 1e-6) * scale
    dim_t = temperature ** (2 * torch.div(torch.arange(npf, dtype=torch.float32), 2, rounding_mode="floor") / npf)
    px, py = x[..., None] / dim_t, y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3)


class DeformableTransformer:
    def __init__(self, params: Dict[str, torch.Tensor], *, d_model=256, num_levels=4, enc_layers=6, dec_layers=6, n_heads=8, enc_points=4, dec_points=4,
                 device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("the Deformable-DETR path runs on the MI355X HIP library only; there is no CPU fallback")
        self.d, self.L, self.ne, self.nd, self.M, self.pe, self.pd = d_model, num_levels, enc_layers, dec_layers, n_heads, enc_points, dec_points
        self.dev = torch.device(device)
        self.p = {k: v.detach().to(self.dev, torch.float32).contiguous() for k, v in params.items()}
        p = self.p
        for l in range(num_levels):                                   # conv weights [Cout, Cin, KH, KW] -> [Cout, KH, KW, Cin]
            p[f"input_proj.{l}.0.weight"] = p[f"input_proj.{l}.0.weight"].permute(0, 2, 3, 1).contiguous()
        for pre in [f"transformer.encoder.layers.{i}.self_attn" for i in range(enc_layers)] + [f"transformer.decoder.layers.{i}.cross_attn" for i in range(dec_layers)]:
            p[pre + ".so_aw.weight"] = torch.cat([p[pre + ".sampling_offsets.weight"], p[pre + ".attention_weights.weight"]]).contiguous()
            p[pre + ".so_aw.bias"] = torch.cat([p[pre + ".sampling_offsets.bias"], p[pre + ".attention_weights.bias"]]).contiguous()
        qe = p["query_embed.weight"]
        self.query_pos, self.tgt0 = qe[:, :d_model].contiguous(), qe[:, d_model:].contiguous()
        # the decoder's reference points depend on the parameters only
        self.reference = torch.sigmoid(_linear(self.query_pos, p["transformer.reference_points.weight"], p["transformer.reference_points.bias"])).contiguous()
        self._tables = {}

    # ------------------------------------------------------------------------------------------------ input-independent tables
    def tables(self, image_mask: torch.Tensor, shapes: Sequence[Tuple[int, int]]):
        """per (padding mask, level shapes): level masks, position + level embeddings, valid ratios, encoder / decoder reference points"""
        key = (tuple(shapes), tuple(image_mask.shape), hash(image_mask.cpu().numpy().tobytes()))
        t = self._tables.get(key)
        if t is not None:
            return t
        m = image_mask.cpu()
        B = m.shape[0]
        ms = [torch.nn.functional.interpolate(m[None].float(), size=s).to(torch.bool)[0] for s in shapes]
        lvl = self.p["transformer.level_embed"].cpu()
        pos = torch.cat([(sine_position_embedding(mm, self.d) + lvl[l].view(1, 1, 1, -1)).flatten(1, 2) for l, mm in enumerate(ms)], 1)
        keep = torch.cat([(~mm).flatten(1) for mm in ms], 1).to(torch.uint8)
        vr = torch.stack([torch.stack([(~mm[:, 0, :]).sum(1).float() / mm.shape[2], (~mm[:, :, 0]).sum(1).float() / mm.shape[1]], -1) for mm in ms], 1)   # (B, L, 2) = (w, h)
        ref = []
        for l, (H, W_) in enumerate(shapes):
            ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
            ref.append(torch.stack((rx.reshape(-1)[None] / (vr[:, None, l, 0] * W_), ry.reshape(-1)[None] / (vr[:, None, l, 1] * H)), -1))
        ref_enc = (torch.cat(ref, 1)[:, :, None] * vr[:, None]).contiguous()                       # (B, S, L, 2)
        ref_dec = (self.reference.cpu()[None, :, None] * vr[:, None]).contiguous()                 # (B, Nq, L, 2)
        sh = torch.tensor([[h, w] for h, w in shapes], dtype=torch.int32)
        ls = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1]), dtype=torch.int32)
        t = dict(pos=pos.to(self.dev).contiguous(), keep=keep.to(self.dev).contiguous(), ref_enc=ref_enc.to(self.dev), ref_dec=ref_dec.to(self.dev),
                 shapes=sh.to(self.dev), lstart=ls.to(self.dev), S=int(pos.shape[1]))
        self._tables[key] = t
        return t

    # ------------------------------------------------------------------------------------------------ pieces
    def _deform_attn(self, pre: str, query: torch.Tensor, ref: torch.Tensor, value_in: torch.Tensor, res: torch.Tensor, t: dict, B: int, points: int):
        """query [B*Q, d] (already + position), ref [B, Q, L, 2], value_in [B*S, d]; -> output_proj(attention) + res"""
        p, M, Lv, d = self.p, self.M, self.L, self.d
        T = query.shape[0]
        raw = _linear(query, p[pre + ".so_aw.weight"], p[pre + ".so_aw.bias"])
        value = _linear(value_in, p[pre + ".value_proj.weight"], p[pre + ".value_proj.bias"])
        L.call("aldi_mask_rows", _p(value), _p(t["keep"]), value.shape[0], d, stream_ptr())
        loc = torch.empty((T, M, Lv, points, 2), dtype=torch.float32, device=self.dev)
        aw = torch.empty((T, M, Lv, points), dtype=torch.float32, device=self.dev)
        L.call("aldi_msda_prepare", _p(raw), _p(ref), _p(t["shapes"]), _p(loc), _p(aw), T, M, Lv, points, stream_ptr())
        out = torch.empty((T, d), dtype=torch.float32, device=self.dev)
        L.call("aldi_ms_deform_attn_forward", _p(value), _p(t["shapes"]), _p(t["lstart"]), _p(loc), _p(aw), _p(out), B, t["S"], M, d // M, T // B, Lv, points,
               stream_ptr())
        return _linear(out, p[pre + ".output_proj.weight"], p[pre + ".output_proj.bias"], res=res)

    def _ln(self, name: str, x: torch.Tensor) -> torch.Tensor:
        return V.layernorm_forward(x, self.p[name + ".weight"], self.p[name + ".bias"], eps=1e-5)[0]

    def _add(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return ops.add_f32(a, b, torch.empty_like(a))

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, feats: List[torch.Tensor], image_mask: torch.Tensor):
        """feats: the backbone's maps, NHWC fp32 on the device (C3, C4, C5); image_mask (B, H, W) bool, True = padding.
        -> (logits [dec_layers, B, Nq, K], boxes [dec_layers, B, Nq, 4] as (cx, cy, w, h) in [0, 1])"""
        p, d = self.p, self.d
        B = feats[0].shape[0]
        srcs = []
        for l in range(self.L):
            w, b = p[f"input_proj.{l}.0.weight"], p[f"input_proj.{l}.0.bias"]
            if l < len(feats):
                x = ops.conv2d(feats[l], w, shift=b)
            else:
                x = ops.conv2d(feats[-1] if l == len(feats) else srcs[-1], w, shift=b, stride=2, pad=1)
            srcs.append(group_norm(x, p[f"input_proj.{l}.1.weight"], p[f"input_proj.{l}.1.bias"]))
        shapes = [(s.shape[1], s.shape[2]) for s in srcs]
        t = self.tables(image_mask, shapes)
        S = t["S"]
        x = torch.cat([s.view(B, -1, d) for s in srcs], 1).view(B * S, d).contiguous()
        pos = t["pos"].view(B * S, d)
        for i in range(self.ne):
            pre = f"transformer.encoder.layers.{i}"
            x = self._ln(pre + ".norm1", self._deform_attn(pre + ".self_attn", self._add(x, pos), t["ref_enc"], x, x, t, B, self.pe))
            h = _linear(x, p[pre + ".linear1.weight"], p[pre + ".linear1.bias"], relu=True)
            x = self._ln(pre + ".norm2", _linear(h, p[pre + ".linear2.weight"], p[pre + ".linear2.bias"], res=x))
        memory = x
        Nq = self.query_pos.shape[0]
        qpos = self.query_pos[None].expand(B, -1, -1).reshape(B * Nq, d).contiguous()
        tgt = self.tgt0[None].expand(B, -1, -1).reshape(B * Nq, d).contiguous()
        hs = []
        M, dh = self.M, d // self.M
        for i in range(self.nd):
            pre = f"transformer.decoder.layers.{i}"
            w, b = p[pre + ".self_attn.in_proj_weight"], p[pre + ".self_attn.in_proj_bias"]
            qk = _linear(self._add(tgt, qpos), w[:2 * d], b[:2 * d])
            v = _linear(tgt, w[2 * d:], b[2 * d:])
            att = torch.empty((B * Nq, d), dtype=torch.float32, device=self.dev)
            L.call("aldi_mha_small_forward", _p(qk), qk.data_ptr() + 4 * d, _p(v), _p(att), None, B, Nq, M, dh, 2 * d, 2 * d, d, float(dh) ** -0.5, stream_ptr())
            tgt = self._ln(pre + ".norm2", _linear(att, p[pre + ".self_attn.out_proj.weight"], p[pre + ".self_attn.out_proj.bias"], res=tgt))
            tgt = self._ln(pre + ".norm1", self._deform_attn(pre + ".cross_attn", self._add(tgt, qpos), t["ref_dec"], memory, tgt, t, B, self.pd))
            h = _linear(tgt, p[pre + ".linear1.weight"], p[pre + ".linear1.bias"], relu=True)
            tgt = self._ln(pre + ".norm3", _linear(h, p[pre + ".linear2.weight"], p[pre + ".linear2.bias"], res=tgt))
            hs.append(tgt)
        hs = torch.cat(hs)                                                     # [dec_layers * B * Nq, d]
        logits = _linear(hs, p["class_embed.weight"], p["class_embed.bias"])
        h = _linear(hs, p["bbox_embed.layers.0.weight"], p["bbox_embed.layers.0.bias"], relu=True)
        h = _linear(h, p["bbox_embed.layers.1.weight"], p["bbox_embed.layers.1.bias"], relu=True)
        tb = _linear(h, p["bbox_embed.layers.2.weight"], p["bbox_embed.layers.2.bias"])
        boxes = torch.empty_like(tb)
        L.call("aldi_detr_box_finish", _p(tb), _p(self.reference), _p(boxes), tb.shape[0], Nq, stream_ptr())
        K = logits.shape[1]
        return logits.view(self.nd, B, Nq, K), boxes.view(self.nd, B, Nq, 4)
