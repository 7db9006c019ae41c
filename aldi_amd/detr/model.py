"""The Deformable-DETR detector AFTER its backbone on the HIP library (BASELINE configs[4]): input projections +
GroupNorm, deformable encoder, decoder (self attention + deformable cross attention), class / box heads, forward AND backward -- every map
a C-ABI launch (linear layers = aldi_conv_igemm / aldi_conv_wgrad with H = W = 1 in fp32, aldi_group_norm_*, aldi_layernorm_*,
aldi_msda_prepare* + aldi_ms_deform_attn_*, aldi_mha_small_*, aldi_detr_box_finish*).  fp32: the reference runs this detector with AMP
off (configs/Base-DETR.yaml:56-58).

The reference's own detector is an absent submodule (`aldi/detr/libs/DeformableDETRDetectron2`, .gitmodules:4-6; registered as
`DETRDistillMixin` / `DETRAlignMixin` in aldi/detr/distill.py:6-7, aldi/detr/align.py:6-7); parameter names are the authors'
(oracle/deformable_detr.py lists them) and the arithmetic -- outputs and every parameter gradient -- is held to that oracle (itself
pinned against transformers' implementation) in tests/test_detr_gpu.py.

No autograd: the forward records a tape of the launches it made and `backward` walks it in reverse (the ViT path's way, aldi_amd/vit.py).
Dropout (TRANSFORMER.DROPOUT, 0.1 in the reference's config) acts in recorded (training) forwards at the authors' sites -- after each
attention / feed-forward block, inside the feed-forward block and on the decoder self attention's probabilities -- through
aldi_dropout_add / aldi_mha_small_*: stateless keep decisions from (seed, element), recomputed by the backward; its own generator, not
torch's random stream."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L
from .. import ops
from .. import vit_ops as V
from ..ops import _p, stream_ptr


# ---------------------------------------------------------------------------------------------------------------- parameters
def param_spec(*, d_model=256, num_levels=4, enc_layers=6, dec_layers=6, n_heads=8, enc_points=4, dec_points=4, ffn=1024, num_queries=300, num_classes=80,
               backbone_channels=(512, 1024, 2048)) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape in the authors' state-dict layout (conv weights [Cout, Cin, KH, KW]).  Order matters: a deformable-attention layer's
    sampling_offsets / attention_weights tensors are adjacent so that the two linear maps run as one launch on a view of the flat buffer."""
    d, s = d_model, OrderedDict()
    for l in range(num_levels):
        cin = backbone_channels[l] if l < len(backbone_channels) else (backbone_channels[-1] if l == len(backbone_channels) else d)
        k = 1 if l < len(backbone_channels) else 3
        s[f"input_proj.{l}.0.weight"], s[f"input_proj.{l}.0.bias"] = (d, cin, k, k), (d,)
        s[f"input_proj.{l}.1.weight"], s[f"input_proj.{l}.1.bias"] = (d,), (d,)
    s["transformer.level_embed"] = (num_levels, d)

    def attn(pre, pts):
        n = n_heads * num_levels * pts
        s[pre + ".sampling_offsets.weight"], s[pre + ".attention_weights.weight"] = (2 * n, d), (n, d)
        s[pre + ".sampling_offsets.bias"], s[pre + ".attention_weights.bias"] = (2 * n,), (n,)
        for nm in ("value_proj", "output_proj"):
            s[pre + f".{nm}.weight"], s[pre + f".{nm}.bias"] = (d, d), (d,)

    def ffn_norms(pre, norms):
        s[pre + ".linear1.weight"], s[pre + ".linear1.bias"] = (ffn, d), (ffn,)
        s[pre + ".linear2.weight"], s[pre + ".linear2.bias"] = (d, ffn), (d,)
        for nm in norms:
            s[pre + f".{nm}.weight"], s[pre + f".{nm}.bias"] = (d,), (d,)
    for i in range(enc_layers):
        attn(f"transformer.encoder.layers.{i}.self_attn", enc_points)
        ffn_norms(f"transformer.encoder.layers.{i}", ("norm1", "norm2"))
    for i in range(dec_layers):
        pre = f"transformer.decoder.layers.{i}"
        attn(pre + ".cross_attn", dec_points)
        s[pre + ".self_attn.in_proj_weight"], s[pre + ".self_attn.in_proj_bias"] = (3 * d, d), (3 * d,)
        s[pre + ".self_attn.out_proj.weight"], s[pre + ".self_attn.out_proj.bias"] = (d, d), (d,)
        ffn_norms(pre, ("norm1", "norm2", "norm3"))
    s["transformer.reference_points.weight"], s["transformer.reference_points.bias"] = (2, d), (2,)
    s["query_embed.weight"] = (num_queries, 2 * d)
    s["class_embed.weight"], s["class_embed.bias"] = (num_classes, d), (num_classes,)
    for j, (o, i) in enumerate(((d, d), (d, d), (4, d))):
        s[f"bbox_embed.layers.{j}.weight"], s[f"bbox_embed.layers.{j}.bias"] = (o, i), (o,)
    return s


class FlatParams:
    """All parameters in ONE fp32 buffer (+ gradient, + AdamW moments): the optimizer, the EMA and the gradient-norm clip are single
    launches over it.  Conv weights are kept in the kernels' [Cout, KH, KW, Cin] order (converted at load / state_dict time)."""
    def __init__(self, spec: "OrderedDict[str, Tuple[int, ...]]", device="cuda", trainable: bool = True):
        self.spec, self.dev = spec, torch.device(device)
        self.off, n = {}, 0
        for k, shp in spec.items():
            self.off[k] = n
            n += (int(math.prod(shp)) + 3) // 4 * 4                     # 16-byte aligned tensors
        self.n = n
        self.master = torch.zeros(n, dtype=torch.float32, device=self.dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=self.dev) if trainable else None
        self.m = self.v = None
        self.step = 0

    def _shape(self, k):
        shp = self.spec[k]
        return (shp[0], shp[2], shp[3], shp[1]) if len(shp) == 4 else shp

    def p(self, k: str) -> torch.Tensor:
        return self.master[self.off[k]: self.off[k] + int(math.prod(self.spec[k]))].view(self._shape(k))

    def g(self, k: str) -> torch.Tensor:
        return self.grad[self.off[k]: self.off[k] + int(math.prod(self.spec[k]))].view(self._shape(k))

    def span(self, buf: torch.Tensor, first: str, last: str, shape) -> torch.Tensor:
        """a view over ADJACENT tensors first .. last (sampling_offsets + attention_weights as one linear map)"""
        a, b = self.off[first], self.off[last] + int(math.prod(self.spec[last]))
        assert b - a == int(math.prod(shape)), (first, last, shape)
        return buf[a:b].view(shape)

    def ranges(self, names) -> List[Tuple[int, int]]:
        return [(self.off[k], self.off[k] + int(math.prod(self.spec[k]))) for k in names]

    def load(self, sd: Dict[str, torch.Tensor]):
        for k in self.spec:
            t = sd[k].detach().to(torch.float32)
            if t.dim() == 4:
                t = t.permute(0, 2, 3, 1)
            self.p(k).copy_(t.contiguous().to(self.dev))

    def state_dict(self, buf: Optional[torch.Tensor] = None) -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        for k, shp in self.spec.items():
            t = (self.master if buf is None else buf)[self.off[k]: self.off[k] + int(math.prod(shp))].view(self._shape(k))
            out[k] = (t.permute(0, 3, 1, 2) if len(shp) == 4 else t).contiguous().clone()
        return out

    def zero_grad(self):
        self.grad.zero_()

    def adamw_step(self, lr: float, *, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, grad_scale: float = 1.0, lr_scale: Optional[Dict[str, float]] = None):
        """torch.optim.AdamW over the flat buffer; lr_scale: {substring of a parameter name: factor} (LR_LINEAR_PROJ_NAMES / MULTIPLIER of
        configs/Base-DETR.yaml:66-69) -- those tensors step with their own learning rate"""
        if self.m is None:
            self.m, self.v = torch.zeros_like(self.master), torch.zeros_like(self.master)
        self.step += 1
        segs, at = [], 0
        if lr_scale:
            for k in self.spec:
                f = next((v for s_, v in lr_scale.items() if s_ in k), None)
                if f is not None:
                    a, b = self.ranges([k])[0]
                    if a > at:
                        segs.append((at, a, 1.0))
                    segs.append((a, b, f))
                    at = b
        segs.append((at, self.n, 1.0))
        for a, b, f in segs:
            if b > a:
                V.adamw_step(self.master[a:b], self.grad[a:b], self.m[a:b], self.v[a:b], None, lr=lr * f, betas=betas, eps=eps, weight_decay=weight_decay,
                             step=self.step, grad_scale=grad_scale)


def _linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, relu: bool = False, res: torch.Tensor = None) -> torch.Tensor:
    """x [T, Cin] fp32, w [Cout, Cin] -> [T, Cout] (+ res, ReLU in the epilogue)"""
    T, Cin = x.shape
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


class _Grads:
    """gradient slots of the tape's intermediate tensors, keyed by tensor identity; `add` accumulates"""
    def __init__(self):
        self.d = {}

    def add(self, t: torch.Tensor, g: torch.Tensor):
        k = id(t)
        g = g.reshape(t.shape) if g.shape != t.shape else g
        e = self.d.get(k)
        if e is None:
            self.d[k] = [t, g, False]              # not ours yet: the same tensor may sit in another slot (a + b hands g to both)
        elif e[2]:
            ops.add_f32(e[1], g, e[1])
        else:
            e[1], e[2] = ops.add_f32(e[1].contiguous(), g.contiguous(), torch.empty_like(e[1], memory_format=torch.contiguous_format)), True

    def pop(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        e = self.d.pop(id(t), None)
        return None if e is None else e[1]


class DeformableTransformer:
    def __init__(self, params, *, d_model=256, num_levels=4, enc_layers=6, dec_layers=6, n_heads=8, enc_points=4, dec_points=4, device="cuda",
                 trainable: bool = True, dropout: float = 0.0, seed: int = 0):
        """params: a FlatParams, or a state dict in the authors' names (then a FlatParams is built from its shapes)"""
        if not torch.cuda.is_available():
            raise RuntimeError("the Deformable-DETR path runs on the MI355X HIP library only; there is no CPU fallback")
        self.d, self.L, self.ne, self.nd, self.M, self.pe, self.pd = d_model, num_levels, enc_layers, dec_layers, n_heads, enc_points, dec_points
        self.dev = torch.device(device)
        if not isinstance(params, FlatParams):
            sd = params
            chans = tuple(sd[f"input_proj.{l}.0.weight"].shape[1] for l in range(num_levels) if sd[f"input_proj.{l}.0.weight"].shape[-1] == 1)
            spec = param_spec(d_model=d_model, num_levels=num_levels, enc_layers=enc_layers, dec_layers=dec_layers, n_heads=n_heads, enc_points=enc_points,
                              dec_points=dec_points, ffn=sd["transformer.encoder.layers.0.linear1.weight"].shape[0], num_queries=sd["query_embed.weight"].shape[0],
                              num_classes=sd["class_embed.weight"].shape[0], backbone_channels=chans)
            params = FlatParams(spec, device, trainable)
            params.load(sd)
        self.P = params
        self._tables = {}
        self.tape = None
        self.dropout, self._seed0, self._fwd_count, self._drop_on = float(dropout), int(seed), 0, False
        self.drop_sites = {}                   # site name -> (seed, shape) of the last recorded forward (tests rebuild the masks from it)

    # ------------------------------------------------------------------------------------------------ tape plumbing
    def _rec(self, fn):
        if self.tape is not None:
            self.tape.append(fn)

    def _lin(self, x: torch.Tensor, wname: str, *, relu=False, res=None, w=None, b=None, gw=None, gb=None, rows=None) -> torch.Tensor:
        """linear map by parameter name (or explicit views w / b with their gradient views gw / gb); records its backward"""
        P = self.P
        if w is None:
            w, b = P.p(wname + ".weight"), P.p(wname + ".bias")
            if self.tape is not None:
                gw, gb = P.g(wname + ".weight"), P.g(wname + ".bias")
        if rows is not None:                                               # a row slice of a packed projection (q | k | v)
            w, b = w[rows[0]:rows[1]], b[rows[0]:rows[1]]
            if gw is not None:
                gw, gb = gw[rows[0]:rows[1]], gb[rows[0]:rows[1]]
        Cout, Cin = w.shape
        if Cout % 4:                                                       # (the kernels write 4 channels at a time: the 2-output reference-point layer)
            y = (x @ w.t() + b)                                            # parameter-only arithmetic on [Nq, 2]: torch
            if res is not None:
                y = y + res
            def bwd(G, x=x, y=y, w=w, gw=gw, gb=gb):
                g = G.pop(y)
                if g is None:
                    return
                gw += g.t() @ x
                gb += g.sum(0)
                G.add(x, g @ w)
            self._rec(bwd)
            return y
        y = _linear(x, w, b, relu=relu, res=res)

        def bwd(G, x=x, y=y, w=w, gw=gw, gb=gb, relu=relu, res=res):
            g = G.pop(y)
            if g is None:
                return
            if relu:
                g = ops.add_f32(None, g, torch.empty_like(g), relu_src=y)
            T = x.shape[0]
            ops.conv_wgrad(x.view(T, 1, 1, Cin), g.view(T, 1, 1, Cout), gw, KH=1, KW=1, db=gb)
            wt = ops.dgrad_weights(w.view(Cout, 1, 1, Cin), None, torch.float32)
            G.add(x, ops.conv2d(g.view(T, 1, 1, Cout), wt).view(T, Cin))
            if res is not None:
                G.add(res, g)
        self._rec(bwd)
        return y

    def _ln(self, name: str, x: torch.Tensor) -> torch.Tensor:
        P = self.P
        y, mean, rstd = V.layernorm_forward(x, P.p(name + ".weight"), P.p(name + ".bias"), eps=1e-5)

        def bwd(G, x=x, y=y, mean=mean, rstd=rstd):
            g = G.pop(y)
            if g is not None:
                G.add(x, V.layernorm_backward(g.contiguous(), x, P.p(name + ".weight"), mean, rstd, P.g(name + ".weight"), P.g(name + ".bias")))
        self._rec(bwd)
        return y

    def _site_seed(self, name: str, shape) -> int:
        sd = ((self._seed0 * 1000003 + self._fwd_count) * 4099 + len(self.drop_sites) + 1) & ((1 << 63) - 1)
        self.drop_sites[name] = (sd, tuple(shape))
        return sd

    def _dropout(self, name: str, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """res + dropout(x) (res nullable) at site `name`"""
        p, sd = self.dropout, self._site_seed(name, x.shape)
        out = torch.empty_like(x)
        L.call("aldi_dropout_add", _p(x), _p(res), _p(out), x.numel(), p, sd, stream_ptr())

        def bwd(G, x=x, res=res, out=out):
            g = G.pop(out)
            if g is None:
                return
            gx = torch.empty_like(x)
            L.call("aldi_dropout_add", _p(g.contiguous()), None, _p(gx), x.numel(), p, sd, stream_ptr())
            G.add(x, gx)
            if res is not None:
                G.add(res, g)
        self._rec(bwd)
        return out

    def _lin_res(self, x: torch.Tensor, wname: str, res: torch.Tensor, site: str) -> torch.Tensor:
        """res + dropout(linear(x)): the residual add stays in the linear map's epilogue when dropout is off"""
        if not self._drop_on:
            return self._lin(x, wname, res=res)
        return self._dropout(site, self._lin(x, wname), res)

    def _add(self, a: torch.Tensor, b: torch.Tensor, b_const: bool = False) -> torch.Tensor:
        y = ops.add_f32(a, b, torch.empty_like(a))

        def bwd(G, a=a, b=b, y=y):
            g = G.pop(y)
            if g is not None:
                G.add(a, g)
                if not b_const:
                    G.add(b, g)
        self._rec(bwd)
        return y

    # ------------------------------------------------------------------------------------------------ input-independent tables
    def tables(self, image_mask: torch.Tensor, shapes: Sequence[Tuple[int, int]]):
        """per (padding mask, level shapes): level masks, position embeddings, valid ratios, the encoder's reference points"""
        key = (tuple(shapes), tuple(image_mask.shape), hash(image_mask.cpu().numpy().tobytes()))
        t = self._tables.get(key)
        if t is not None:
            return t
        m = image_mask.cpu()
        ms = [torch.nn.functional.interpolate(m[None].float(), size=s).to(torch.bool)[0] for s in shapes]
        pos = torch.cat([sine_position_embedding(mm, self.d).flatten(1, 2) for mm in ms], 1)                 # (B, S, d) without the level embedding
        lvl_of = torch.cat([torch.full((h * w,), l, dtype=torch.long) for l, (h, w) in enumerate(shapes)])
        keep = torch.cat([(~mm).flatten(1) for mm in ms], 1).to(torch.uint8)
        vr = torch.stack([torch.stack([(~mm[:, 0, :]).sum(1).float() / mm.shape[2], (~mm[:, :, 0]).sum(1).float() / mm.shape[1]], -1) for mm in ms], 1)   # (B, L, 2) = (w, h)
        ref = []
        for l, (H, W_) in enumerate(shapes):
            ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
            ref.append(torch.stack((rx.reshape(-1)[None] / (vr[:, None, l, 0] * W_), ry.reshape(-1)[None] / (vr[:, None, l, 1] * H)), -1))
        ref_enc = (torch.cat(ref, 1)[:, :, None] * vr[:, None]).contiguous()                       # (B, S, L, 2)
        sh = torch.tensor([[h, w] for h, w in shapes], dtype=torch.int32)
        ls = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1]), dtype=torch.int32)
        t = dict(pos=pos.to(self.dev).contiguous(), lvl_of=lvl_of.to(self.dev), keep=keep.to(self.dev).contiguous(), ref_enc=ref_enc.to(self.dev), vr=vr.to(self.dev),
                 shapes=sh.to(self.dev), lstart=ls.to(self.dev), S=int(pos.shape[1]), shapes_host=np.ascontiguousarray(sh.numpy().astype(np.int32)))
        if len(self._tables) >= 16:                 # (batches of varying sizes / paddings: keep the most recent geometries only)
            self._tables.pop(next(iter(self._tables)))
        self._tables[key] = t
        return t

    # ------------------------------------------------------------------------------------------------ pieces
    def _group_norm(self, x: torch.Tensor, name: str) -> torch.Tensor:
        P = self.P
        N, H, W_, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((N, 32), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(max(int(L.lib.aldi_group_norm_workspace(N, H * W_, 32)), 4), dtype=torch.uint8, device=x.device)
        L.call("aldi_group_norm_forward", _p(x), _p(P.p(name + ".weight")), _p(P.p(name + ".bias")), _p(y), _p(mean), _p(rstd), _p(ws), N, H * W_, C, 32, 1e-5, stream_ptr())

        def bwd(G, x=x, y=y, mean=mean, rstd=rstd):
            g = G.pop(y)
            if g is None:
                return
            dx = torch.empty_like(x)
            ws2 = torch.empty(int(L.lib.aldi_group_norm_backward_workspace(N, H * W_, C, 32)), dtype=torch.uint8, device=x.device)
            L.call("aldi_group_norm_backward", _p(g.contiguous()), _p(x), _p(P.p(name + ".weight")), _p(mean), _p(rstd), _p(dx), _p(P.g(name + ".weight")),
                   _p(P.g(name + ".bias")), _p(ws2), N, H * W_, C, 32, stream_ptr())
            G.add(x, dx)
        self._rec(bwd)
        return y

    def _input_conv(self, x: torch.Tensor, name: str, stride2: bool) -> torch.Tensor:
        """the input projection's convolution: 1x1, or (extra levels) 3x3 / stride 2 / pad 1 = the stride-1 conv sampled at the even
        positions (the maps are tiny there; keeps every launch on the stride-1 kernels, forward and backward)"""
        P = self.P
        w, b = P.p(name + ".weight"), P.p(name + ".bias")
        k = w.shape[1]
        full = ops.conv2d(x, w, shift=b, pad=k // 2)
        y = ops.subsample2(full) if stride2 else full

        def bwd(G, x=x, y=y, full=full):
            g = G.pop(y)
            if g is None:
                return
            if stride2:
                gf = torch.zeros_like(full)
                ops.subsample2_bwd(g.contiguous(), gf)
                g = gf
            ops.conv_wgrad(x, g, P.g(name + ".weight"), KH=k, KW=k, stride=1, pad=k // 2, db=P.g(name + ".bias"))
            if x.requires_grad_flag:
                G.add(x, ops.conv2d(g, ops.dgrad_weights(w, None, torch.float32), pad=k // 2))
        self._rec(bwd)
        return y

    def _msda_workspace(self, t: dict, B: int, S: int, M: int, Lv: int, points: int) -> torch.Tensor:
        """the sample lists of the encoder's gathered value gradient (reused by every layer and every geometry)"""
        need = int(L.lib.aldi_ms_deform_attn_backward_self_workspace(t["shapes_host"].ctypes.data, B, S, M, Lv, points))
        ws = getattr(self, "_msda_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._msda_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)       # ONE buffer, grown to the largest geometry seen
        return ws

    def _deform_attn(self, pre: str, query: torch.Tensor, ref: torch.Tensor, value_in: torch.Tensor, res: torch.Tensor, t: dict, B: int, points: int,
                     ref_grad: Optional[torch.Tensor] = None):
        """query [B*Q, d] (already + position), ref [B*Q, L, 2], value_in [B*S, d]; -> output_proj(attention) + res.
        ref_grad: a [B*Q, L, 2] tensor whose gradient slot receives the reference points' gradient (the decoder's are learnt)"""
        P, M, Lv, d = self.P, self.M, self.L, self.d
        T = query.shape[0]
        n = M * Lv * points
        so, aw_ = pre + ".sampling_offsets", pre + ".attention_weights"
        wcat = P.span(P.master, so + ".weight", aw_ + ".weight", (3 * n, d))
        bcat = P.span(P.master, so + ".bias", aw_ + ".bias", (3 * n,))
        gw = P.span(P.grad, so + ".weight", aw_ + ".weight", (3 * n, d)) if self.tape is not None else None
        gb = P.span(P.grad, so + ".bias", aw_ + ".bias", (3 * n,)) if self.tape is not None else None
        raw = self._lin(query, None, w=wcat, b=bcat, gw=gw, gb=gb)
        value = self._lin(value_in, pre + ".value_proj")
        L.call("aldi_mask_rows", _p(value), _p(t["keep"]), value.shape[0], d, stream_ptr())
        loc = torch.empty((T, M, Lv, points, 2), dtype=torch.float32, device=self.dev)
        aw = torch.empty((T, M, Lv, points), dtype=torch.float32, device=self.dev)
        L.call("aldi_msda_prepare", _p(raw), _p(ref), _p(t["shapes"]), _p(loc), _p(aw), T, M, Lv, points, stream_ptr())
        out = torch.empty((T, d), dtype=torch.float32, device=self.dev)
        S = t["S"]
        L.call("aldi_ms_deform_attn_forward", _p(value), _p(t["shapes"]), _p(t["lstart"]), _p(loc), _p(aw), _p(out), B, S, M, d // M, T // B, Lv, points, stream_ptr())

        def bwd(G, raw=raw, value=value, loc=loc, aw=aw, out=out):
            g = G.pop(out)
            if g is None:
                return
            gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(aw)
            if pre.endswith(".self_attn") and T == B * S:    # the encoder's self attention: the queries are the pyramid's positions
                ws = self._msda_workspace(t, B, S, M, Lv, points)
                L.call("aldi_ms_deform_attn_backward_self", _p(value), _p(t["shapes"]), _p(t["lstart"]), t["shapes_host"].ctypes.data, _p(loc), _p(aw),
                       _p(g.contiguous()), _p(gv), _p(gl), _p(ga), _p(ws), ws.numel(), B, S, M, d // M, Lv, points, stream_ptr())
            else:
                L.call("aldi_ms_deform_attn_backward", _p(value), _p(t["shapes"]), _p(t["lstart"]), _p(loc), _p(aw), _p(g.contiguous()), _p(gv), _p(gl), _p(ga),
                       B, S, M, d // M, T // B, Lv, points, stream_ptr())
            L.call("aldi_mask_rows", _p(gv), _p(t["keep"]), gv.shape[0], d, stream_ptr())
            g_raw = torch.empty_like(raw)
            g_ref = torch.empty((T, Lv, 2), dtype=torch.float32, device=self.dev) if ref_grad is not None else None
            L.call("aldi_msda_prepare_backward", _p(gl), _p(ga), _p(aw), _p(t["shapes"]), _p(g_raw), _p(g_ref), T, M, Lv, points, stream_ptr())
            G.add(raw, g_raw)
            G.add(value, gv)
            if ref_grad is not None:
                G.add(ref_grad, g_ref)
        self._rec(bwd)
        return self._lin_res(out, pre + ".output_proj", res, pre.rsplit(".", 1)[0] + ".dropout1")

    def _self_attn(self, pre: str, tgt: torch.Tensor, qpos: torch.Tensor, B: int, Nq: int) -> torch.Tensor:
        P, d, M = self.P, self.d, self.M
        dh = d // M
        w, b = P.p(pre + ".in_proj_weight"), P.p(pre + ".in_proj_bias")
        gw = P.g(pre + ".in_proj_weight") if self.tape is not None else None
        gb = P.g(pre + ".in_proj_bias") if self.tape is not None else None
        x_qk = self._add(tgt, qpos)
        qk = self._lin(x_qk, None, w=w, b=b, gw=gw, gb=gb, rows=(0, 2 * d))
        v = self._lin(tgt, None, w=w, b=b, gw=gw, gb=gb, rows=(2 * d, 3 * d))
        att = torch.empty((B * Nq, d), dtype=torch.float32, device=self.dev)
        lse = torch.empty((B, M, Nq), dtype=torch.float32, device=self.dev)
        scale = float(dh) ** -0.5
        dp = self.dropout if self._drop_on else 0.0
        sd = self._site_seed(pre + ".attn", (B, M, Nq, Nq)) if dp > 0 else 0
        L.call("aldi_mha_small_forward", _p(qk), qk.data_ptr() + 4 * d, _p(v), _p(att), _p(lse), B, Nq, M, dh, 2 * d, 2 * d, d, scale, dp, sd, stream_ptr())

        def bwd(G, qk=qk, v=v, att=att, lse=lse):
            g = G.pop(att)
            if g is None:
                return
            dqk, dv = torch.empty_like(qk), torch.empty_like(v)
            delta = torch.empty_like(lse)
            L.call("aldi_mha_small_backward", _p(qk), qk.data_ptr() + 4 * d, _p(v), _p(att), _p(g.contiguous()), _p(lse), _p(dqk), dqk.data_ptr() + 4 * d, _p(dv),
                   _p(delta), B, Nq, M, dh, 2 * d, 2 * d, d, 2 * d, 2 * d, d, scale, dp, sd, stream_ptr())
            G.add(qk, dqk)
            G.add(v, dv)
        self._rec(bwd)
        return self._lin_res(att, pre + ".out_proj", tgt, pre.rsplit(".", 1)[0] + ".dropout2")

    # ------------------------------------------------------------------------------------------------ forward / backward
    def forward(self, feats: List[torch.Tensor], image_mask: torch.Tensor, record: bool = False, feats_need_grad: bool = False):
        """feats: the backbone's maps, NHWC fp32 on the device (C3, C4, C5); image_mask (B, H, W) bool, True = padding.
        -> (logits [dec_layers, B, Nq, K], boxes [dec_layers, B, Nq, 4] as (cx, cy, w, h) in [0, 1]).  record: keep the tape for `backward`."""
        P, d = self.P, self.d
        self.tape = [] if record else None
        self._drop_on = bool(record) and self.dropout > 0.0          # training forwards only (the teacher's inference is in eval mode)
        self.drop_sites = {}
        self._fwd_count += 1
        B = feats[0].shape[0]
        for f in feats:
            f.requires_grad_flag = feats_need_grad                          # (plain attribute: does the backbone want d(loss)/d(feature map))
        srcs = []
        for l in range(self.L):
            src = feats[l] if l < len(feats) else (feats[-1] if l == len(feats) else srcs[-1])
            if not hasattr(src, "requires_grad_flag"):
                src.requires_grad_flag = True
            x = self._input_conv(src, f"input_proj.{l}.0", stride2=l >= len(feats))
            srcs.append(self._group_norm(x, f"input_proj.{l}.1"))
        shapes = [(s.shape[1], s.shape[2]) for s in srcs]
        t = self.tables(image_mask, shapes)
        S = t["S"]
        x = torch.cat([s.view(B, -1, d) for s in srcs], 1).view(B * S, d).contiguous()
        if record:
            def bwd_cat(G, x=x, srcs=srcs):
                g = G.pop(x)
                if g is None:
                    return
                g = g.view(B, S, d)
                o = 0
                for s_ in srcs:
                    n = s_.shape[1] * s_.shape[2]
                    G.add(s_, g[:, o:o + n].reshape(s_.shape).contiguous())
                    o += n
            self._rec(bwd_cat)
        # position embedding + the level embedding of every token's level (a parameter: its gradient is the column sum per level)
        lvl = P.p("transformer.level_embed")
        pos = (t["pos"] + lvl[t["lvl_of"]][None]).view(B * S, d).contiguous()
        if record:
            def bwd_pos(G, pos=pos):
                g = G.pop(pos)
                if g is not None:
                    g = g.view(B, S, d)
                    o = 0
                    for l, (h_, w_) in enumerate(shapes):          # per level: the sum over its tokens (contiguous ranges)
                        P.g("transformer.level_embed")[l] += g[:, o:o + h_ * w_].sum((0, 1))
                        o += h_ * w_
            self._rec(bwd_pos)
        for i in range(self.ne):
            pre = f"transformer.encoder.layers.{i}"
            q = self._add(x, pos)
            x = self._ln(pre + ".norm1", self._deform_attn(pre + ".self_attn", q, t["ref_enc"].view(B * S, self.L, 2), x, x, t, B, self.pe))
            h = self._lin(x, pre + ".linear1", relu=True)
            if self._drop_on:
                h = self._dropout(pre + ".dropout2", h)
            x = self._ln(pre + ".norm2", self._lin_res(h, pre + ".linear2", x, pre + ".dropout3"))
        memory = x
        qe = P.p("query_embed.weight")
        Nq = qe.shape[0]
        query_pos, tgt0 = qe[:, :d].contiguous(), qe[:, d:].contiguous()
        if record:
            def bwd_qe(G, query_pos=query_pos, tgt0=tgt0):
                gq, gt = G.pop(query_pos), G.pop(tgt0)
                ge = P.g("query_embed.weight")
                if gq is not None:
                    ge[:, :d] += gq
                if gt is not None:
                    ge[:, d:] += gt
            self._rec(bwd_qe)
        # reference points of the queries: parameters only ([Nq, 2]: torch arithmetic), learnt through the sampling locations and the box head
        ref_lin = self._lin(query_pos, "transformer.reference_points")
        reference = torch.sigmoid(ref_lin)
        ref_dec = (reference[None, :, None] * t["vr"][:, None]).reshape(B * Nq, self.L, 2).contiguous()
        if record:
            def bwd_ref(G, ref_lin=ref_lin, reference=reference, ref_dec=ref_dec):
                g = G.pop(reference)
                gd = G.pop(ref_dec)
                if gd is not None:
                    gd = (gd.view(B, Nq, self.L, 2) * t["vr"][:, None]).sum((0, 2))
                    g = gd if g is None else g + gd
                if g is not None:
                    G.add(ref_lin, g * reference * (1 - reference))
            self._rec(bwd_ref)
        qpos = query_pos[None].expand(B, -1, -1).reshape(B * Nq, d).contiguous()
        tgt = tgt0[None].expand(B, -1, -1).reshape(B * Nq, d).contiguous()
        if record:
            def bwd_expand(G, qpos=qpos, tgt=tgt):
                for big, small in ((qpos, query_pos), (tgt, tgt0)):
                    g = G.pop(big)
                    if g is not None:
                        G.add(small, g.view(B, Nq, d).sum(0))
            self._rec(bwd_expand)
        hs = []
        for i in range(self.nd):
            pre = f"transformer.decoder.layers.{i}"
            tgt = self._ln(pre + ".norm2", self._self_attn(pre + ".self_attn", tgt, qpos, B, Nq))
            q = self._add(tgt, qpos)
            tgt = self._ln(pre + ".norm1", self._deform_attn(pre + ".cross_attn", q, ref_dec, memory, tgt, t, B, self.pd, ref_grad=ref_dec))
            h = self._lin(tgt, pre + ".linear1", relu=True)
            if self._drop_on:
                h = self._dropout(pre + ".dropout3", h)
            tgt = self._ln(pre + ".norm3", self._lin_res(h, pre + ".linear2", tgt, pre + ".dropout4"))
            hs.append(tgt)
        hs_all = torch.cat(hs)                                                     # [dec_layers * B * Nq, d]
        if record:
            def bwd_hs(G, hs_all=hs_all, hs=hs):
                g = G.pop(hs_all)
                if g is not None:
                    for j, h_ in enumerate(hs):
                        G.add(h_, g[j * B * Nq:(j + 1) * B * Nq].contiguous())
            self._rec(bwd_hs)
        logits = self._lin(hs_all, "class_embed")
        h = self._lin(hs_all, "bbox_embed.layers.0", relu=True)
        h = self._lin(h, "bbox_embed.layers.1", relu=True)
        tb = self._lin(h, "bbox_embed.layers.2")
        boxes = torch.empty_like(tb)
        L.call("aldi_detr_box_finish", _p(tb), _p(reference), _p(boxes), tb.shape[0], Nq, stream_ptr())
        if record:
            def bwd_box(G, tb=tb, boxes=boxes, reference=reference):
                g = G.pop(boxes)
                if g is None:
                    return
                g_t = torch.empty_like(tb)
                g_ref = torch.zeros_like(reference)
                L.call("aldi_detr_box_finish_backward", _p(g.contiguous()), _p(boxes), _p(reference), _p(g_t), _p(g_ref), tb.shape[0], Nq, stream_ptr())
                G.add(tb, g_t)
                G.add(reference, g_ref)
            self._rec(bwd_box)
        K = logits.shape[1]
        self._out = (logits, boxes, feats)
        return logits.view(self.nd, B, Nq, K), boxes.view(self.nd, B, Nq, 4)

    def backward(self, g_logits: torch.Tensor, g_boxes: torch.Tensor) -> List[Optional[torch.Tensor]]:
        """walk the tape of the last recorded forward: parameter gradients ACCUMULATE into FlatParams.grad; -> d(loss)/d(feature map) per
        backbone level (None unless the forward was asked for them)"""
        assert self.tape is not None, "forward(record=True) first"
        logits, boxes, feats = self._out
        G = _Grads()
        G.add(logits, g_logits.reshape(logits.shape).contiguous())
        G.add(boxes, g_boxes.reshape(boxes.shape).contiguous())
        for fn in reversed(self.tape):
            fn(G)
        self.tape = None
        return [G.pop(f) for f in feats]
