"""ViTDet plain-ViT trunk on the HIP path (BASELINE cfg 4).

Host-side mirror of what the reference drives through `aldi/backbone.py:21-43` (checkpointed_vit_forward over
detectron2's `ViT`: PatchEmbed -> + get_abs_pos(pos_embed) -> 12 x Block -> NHWC feature map) and of the optimizer
of `aldi/backbone.py:66-84` (AdamW, no decay on norms / pos_embed).  Parameter names are detectron2's
(`backbone.net.blocks.3.attn.qkv.weight` ...), so reference checkpoints / EMA see the same keys.

`checkpoint(blk, x, use_reentrant=False)` (VIT.USE_ACT_CHECKPOINT) is a memory/time trade with identical results; with
288 GB of HBM the activations of all 12 blocks (~0.2 GB per block for two 800x1344 images) simply stay resident.

Everything numeric is a C-ABI call (ops.py / vit_ops.py); this file only sequences launches and owns the buffers.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from . import vit_ops as V


@dataclass
class VitConfig:
    embed: int = 768
    depth: int = 12
    heads: int = 12
    patch: int = 16
    window: int = 14
    global_blocks: Tuple[int, ...] = (2, 5, 8, 11)
    mlp_ratio: int = 4
    pretrain_grid: int = 14          # pos_embed is (1, pretrain_grid^2 + 1, embed): 224/16, with the cls slot
    rel_input: int = 64              # img_size // patch (1024/16): table length of the global blocks = 2*rel_input - 1
    drop_path_rate: float = 0.1
    sfp: bool = False                # also hold SimpleFeaturePyramid parameters (backbone.simfp_*)
    num_classes: int = 0             # > 0: also hold the ViTDet RPN / box head (configs/Base-RCNN-VitDetB.yaml:7-14)
    fpn_channels: int = 256
    box_convs: int = 4
    fc_dim: int = 1024
    pool: int = 7
    num_anchors: int = 3
    ln_eps: float = 1e-6
    prefix: str = "backbone.net."
    pixel_mean: Tuple[float, float, float] = (123.675, 116.28, 103.53)
    pixel_std: Tuple[float, float, float] = (58.395, 57.12, 57.375)

    def spec(self) -> "OrderedDict[str, Tuple[Tuple[int, ...], bool]]":
        """name -> (state_dict shape, weight_decay?) in detectron2 state_dict order.  4-D conv / deconv weights outside the
        patch embedding are held channel-last in the flat buffers (see VitParams.NHWC)."""
        E, P, p = self.embed, self.patch, self.prefix
        s: "OrderedDict[str, Tuple[Tuple[int, ...], bool]]" = OrderedDict()
        s[p + "pos_embed"] = ((1, self.pretrain_grid ** 2 + 1, E), False)
        s[p + "patch_embed.proj.weight"] = ((E, 3, P, P), True)
        s[p + "patch_embed.proj.bias"] = ((E,), True)
        for i in range(self.depth):
            b = f"{p}blocks.{i}."
            S = self.rel_input if i in self.global_blocks else self.window
            s[b + "norm1.weight"] = ((E,), False)
            s[b + "norm1.bias"] = ((E,), False)
            s[b + "attn.rel_pos_h"] = ((2 * S - 1, E // self.heads), True)
            s[b + "attn.rel_pos_w"] = ((2 * S - 1, E // self.heads), True)
            s[b + "attn.qkv.weight"] = ((3 * E, E), True)
            s[b + "attn.qkv.bias"] = ((3 * E,), True)
            s[b + "attn.proj.weight"] = ((E, E), True)
            s[b + "attn.proj.bias"] = ((E,), True)
            s[b + "norm2.weight"] = ((E,), False)
            s[b + "norm2.bias"] = ((E,), False)
            s[b + "mlp.fc1.weight"] = ((self.mlp_ratio * E, E), True)
            s[b + "mlp.fc1.bias"] = ((self.mlp_ratio * E,), True)
            s[b + "mlp.fc2.weight"] = ((E, self.mlp_ratio * E), True)
            s[b + "mlp.fc2.bias"] = ((E,), True)
        if self.sfp:
            s.update(self.sfp_spec())
        if self.num_classes > 0:
            s.update(self.heads_spec())
        return s

    def heads_spec(self):
        """RPN.CONV_DIMS [-1, -1] (two 3x3+ReLU convs, detectron2 StandardRPNHead), ROI_BOX_HEAD NUM_CONV 4 / NORM "LN" / NUM_FC 1
        (FastRCNNConvFCHead), FastRCNNOutputLayers."""
        C, K, A = self.fpn_channels, self.num_classes, self.num_anchors
        s = OrderedDict()
        rp = "proposal_generator.rpn_head."
        for i in range(2):
            s[f"{rp}conv.conv{i}.weight"] = ((C, C, 3, 3), True)
            s[f"{rp}conv.conv{i}.bias"] = ((C,), True)
        s[rp + "objectness_logits.weight"] = ((A, C, 1, 1), True)
        s[rp + "anchor_deltas.weight"] = ((4 * A, C, 1, 1), True)
        s[rp + "objectness_logits.bias"] = ((A,), True)
        s[rp + "anchor_deltas.bias"] = ((4 * A,), True)
        bh = "roi_heads.box_head."
        for i in range(1, self.box_convs + 1):
            s[f"{bh}conv{i}.weight"] = ((C, C, 3, 3), True)
            s[f"{bh}conv{i}.norm.weight"] = ((C,), True)
            s[f"{bh}conv{i}.norm.bias"] = ((C,), True)
        s[bh + "fc1.weight"] = ((self.fc_dim, C * self.pool * self.pool), True)
        s[bh + "fc1.bias"] = ((self.fc_dim,), True)
        bp = "roi_heads.box_predictor."
        s[bp + "cls_score.weight"] = ((K + 1, self.fc_dim), True)
        s[bp + "bbox_pred.weight"] = ((4 * K, self.fc_dim), True)
        s[bp + "cls_score.bias"] = ((K + 1,), True)
        s[bp + "bbox_pred.bias"] = ((4 * K,), True)
        return s

    def packs(self) -> "OrderedDict[str, Tuple[List[str], int]]":
        """engine tensors made of several state_dict entries stored back to back (rows concatenated, zero rows up to a multiple
        of 16): pack name -> (member names, total elements incl. padding)"""
        if self.num_classes <= 0:
            return OrderedDict()
        C, K, A = self.fpn_channels, self.num_classes, self.num_anchors
        rp, bp = "proposal_generator.rpn_head.", "roi_heads.box_predictor."
        r1, r2 = (5 * A + 15) // 16 * 16, (5 * K + 1 + 15) // 16 * 16
        return OrderedDict([
            ("rpn_head_out.weight", ([rp + "objectness_logits.weight", rp + "anchor_deltas.weight"], r1 * C)),
            ("rpn_head_out.bias", ([rp + "objectness_logits.bias", rp + "anchor_deltas.bias"], r1)),
            ("box_pred.weight", ([bp + "cls_score.weight", bp + "bbox_pred.weight"], r2 * self.fc_dim)),
            ("box_pred.bias", ([bp + "cls_score.bias", bp + "bbox_pred.bias"], r2)),
        ])

    def sfp_spec(self):
        """detectron2 SimpleFeaturePyramid(scale_factors=(4, 2, 1, 0.5), out_channels=256, norm="LN") module names.  Its norms are
        detectron2's channel-first `LayerNorm` class, which is NOT in get_default_optimizer_params' norm_module_types, so they
        take the regular weight decay (unlike the nn.LayerNorm of the blocks)."""
        E, C, q = self.embed, self.fpn_channels, "backbone."
        s = OrderedDict()

        def conv_ln(name, cin, k):
            s[name + ".weight"] = ((C, cin, k, k), True)
            s[name + ".norm.weight"] = ((C,), True)
            s[name + ".norm.bias"] = ((C,), True)
        s[q + "simfp_2.0.weight"] = ((E, E // 2, 2, 2), True)
        s[q + "simfp_2.0.bias"] = ((E // 2,), True)
        s[q + "simfp_2.1.weight"] = ((E // 2,), True)
        s[q + "simfp_2.1.bias"] = ((E // 2,), True)
        s[q + "simfp_2.3.weight"] = ((E // 2, E // 4, 2, 2), True)
        s[q + "simfp_2.3.bias"] = ((E // 4,), True)
        conv_ln(q + "simfp_2.4", E // 4, 1)
        conv_ln(q + "simfp_2.5", C, 3)
        s[q + "simfp_3.0.weight"] = ((E, E // 2, 2, 2), True)
        s[q + "simfp_3.0.bias"] = ((E // 2,), True)
        conv_ln(q + "simfp_3.1", E // 2, 1)
        conv_ln(q + "simfp_3.2", C, 3)
        conv_ln(q + "simfp_4.0", E, 1)
        conv_ln(q + "simfp_4.1", C, 3)
        conv_ln(q + "simfp_5.1", E, 1)
        conv_ln(q + "simfp_5.2", C, 3)
        return s


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class VitParams:
    """Flat fp32 master [decayed | not decayed], bf16 compute copy, fp32 gradient and AdamW moments."""

    def __init__(self, cfg: VitConfig, device, trainable: bool = True):
        self.cfg, self.device, self.trainable = cfg, device, trainable
        self.spec = cfg.spec()
        self.off: Dict[str, int] = {}
        self.pack_off: Dict[str, Tuple[int, int]] = {}
        packs = cfg.packs()
        first = {members[0]: pk for pk, (members, _) in packs.items()}
        packed = {m for members, _ in packs.values() for m in members}
        off = 0
        for decay in (True, False):
            for name, (shape, d) in self.spec.items():
                if d != decay:
                    continue
                if name in first:                              # a pack: members back to back, padding after the last
                    members, total = packs[first[name]]
                    self.pack_off[first[name]] = (off, total)
                    o = off
                    for mname in members:
                        self.off[mname] = o
                        o += torch.Size(self.spec[mname][0]).numel()
                    off += _pad64(total)
                    continue
                if name in packed:
                    continue
                self.off[name] = off
                n = 1
                for v in shape:
                    n *= v
                off += _pad64(n)
            if decay:
                self.n_decay = off
        self.n = off
        self.n_train = off
        self.master = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.compute = torch.zeros(self.n, dtype=torch.bfloat16, device=device)
        self._grad = self._m = self._v = None
        self.step_count = 0
        self._wt_plan: Optional[ops.DgradWeightsPlan] = None
        self._wt_names: List[str] = []

    def _numel(self, name):
        n = 1
        for v in self.spec[name][0]:
            n *= v
        return n

    def full(self, name: str) -> str:
        return name if name.startswith(("backbone.", "proposal_generator.", "roi_heads.")) else self.cfg.prefix + name

    def nhwc(self, name: str) -> bool:
        """conv [Cout,Cin,k,k] / deconv [Cin,Cout,2,2] weights live as [d0][k][k][d1] in the flat buffers (the kernels' layout);
        the patch embedding stays (c, ph, pw): that is the order aldi_patchify emits."""
        return len(self.spec[name][0]) == 4 and "patch_embed" not in name and not self.depthwise(name) and \
            not name.endswith("downsample_layers.0.0.weight")      # ConvNeXt's 4x4/4 stem runs on patch rows, like patch_embed

    def depthwise(self, name: str) -> bool:
        """ConvNeXt depthwise kernels [C,1,7,7] live as [7,7,C] (what aldi_dwconv7 reads)"""
        return name.endswith("dwconv.weight")

    def shape(self, name: str):
        sh = self.spec[name][0]
        if self.depthwise(name):
            return (sh[2], sh[3], sh[0])
        return (sh[0], sh[2], sh[3], sh[1]) if self.nhwc(name) else sh

    def _view(self, buf, name, shape=None):
        o = self.off[name]
        return buf[o:o + self._numel(name)].view(shape or self.shape(name))

    def m(self, name, shape=None):            # fp32 master
        return self._view(self.master, self.full(name), shape)

    def w(self, name, shape=None):            # bf16 compute copy
        return self._view(self.compute, self.full(name), shape)

    @property
    def grad(self):
        if self._grad is None:
            self._grad = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        return self._grad

    def g(self, name, shape=None):
        return self._view(self.grad, self.full(name), shape)

    def pack(self, buf, name, shape):         # a packed engine tensor (see VitConfig.packs)
        o, n = self.pack_off[name]
        return buf[o:o + n].view(shape)

    def ranges(self, names) -> List[Tuple[int, int]]:
        """flat-buffer element ranges of the given state_dict names (full names), for the overlapped gradient exchange.  Extents
        include the 64-element layout padding (never written) so that neighbouring tensors merge into one contiguous range"""
        packed = {m: pk for pk, (members, _) in self.cfg.packs().items() for m in members}
        out = []
        for n in names:
            if n in packed:
                o, tot = self.pack_off[packed[n]]
                r = (o, o + _pad64(tot))
            else:
                r = (self.off[n], self.off[n] + _pad64(self._numel(n)))
            if r not in out:
                out.append(r)
        return out

    def state_dict_keys(self) -> List[str]:
        return list(self.spec.keys())

    @property
    def layout(self):                         # the trainer / EMA / checkpointer address a model's parameters through `.layout`
        return self

    def lin_w(self, name):                    # Linear weight [out, in] as the 1x1 conv weight [out, 1, 1, in]
        o, i = self.spec[self.full(name)][0]
        return self.w(name, (o, 1, 1, i))

    def wt(self, name):
        """data-gradient (transposed) weights of a Linear, re-derived in ONE launch per refresh()"""
        if name not in self._wt_names:
            self._wt_names.append(name)
            self._wt_plan = None
        if self._wt_plan is None:
            ent = []
            for nme in self._wt_names:
                sh = self.shape(self.full(nme))
                ent.append((self.m(nme, sh if len(sh) == 4 else (sh[0], 1, 1, sh[1])), None))
            self._wt_plan = ops.DgradWeightsPlan(ent, torch.bfloat16)
            self._wt_plan.run()
        return self._wt_plan.out[self._wt_names.index(name)]

    # ---- state ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self.master.copy_(self.flatten(sd).to(self.device))
        self.refresh()

    def flatten(self, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        """detectron2-shaped state_dict -> the flat fp32 layout (CPU tensor)"""
        missing = [k for k in self.spec if k not in sd]
        if missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        flat = torch.zeros(self.n, dtype=torch.float32)
        for name, (shape, _) in self.spec.items():
            t = sd[name].detach().to(torch.float32).cpu()
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(shape)}")
            if self.nhwc(name):
                t = t.permute(0, 2, 3, 1).contiguous()
            elif self.depthwise(name):
                t = t[:, 0].permute(1, 2, 0).contiguous()
            elif name.endswith("box_head.fc1.weight"):     # detectron2 flattens (C, 7, 7); the ROIAlign output here is (7, 7, C)
                P_ = self.cfg.pool
                t = t.view(shape[0], -1, P_, P_).permute(0, 2, 3, 1).contiguous()
            flat[self.off[name]:self.off[name] + t.numel()] = t.reshape(-1)
        return flat

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return self.state_dict_like(self.master)

    def state_dict_like(self, buf: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """any flat buffer of this layout (master, grad, moments) as detectron2-shaped tensors"""
        flat = buf.detach().cpu()
        out = OrderedDict()
        for name in self.spec:
            t = flat[self.off[name]:self.off[name] + self._numel(name)].view(self.shape(name))
            if name.endswith("box_head.fc1.weight"):
                P_ = self.cfg.pool
                t = t.view(t.shape[0], P_, P_, -1).permute(0, 3, 1, 2).reshape(t.shape[0], -1)
            if self.depthwise(name):
                t = t.permute(2, 0, 1).unsqueeze(1)
            out[name] = (t.permute(0, 3, 1, 2) if self.nhwc(name) else t).contiguous().clone()
        return out

    def init_random(self, seed: int = 0, std: float = 0.02):
        g = torch.Generator().manual_seed(seed)
        sd = OrderedDict()
        custom = getattr(self.cfg, "init_value", None)
        for name, (shape, _) in self.spec.items():
            v = custom(name, shape) if custom is not None else None
            if v is not None:
                sd[name] = v
            elif name.endswith(("norm1.weight", "norm2.weight", "norm.weight", "simfp_2.1.weight")):
                sd[name] = torch.ones(shape)
            elif name.endswith(".bias"):
                sd[name] = torch.zeros(shape)
            else:
                sd[name] = torch.randn(shape, generator=g) * std
        self.load_state_dict(sd)

    def refresh(self):
        ops.cast_from_f32(self.master, torch.bfloat16, out=self.compute)
        if self._wt_plan is not None:
            self._wt_plan.run()

    def zero_grad(self):
        self.grad.zero_()
        self._gscale = 1.0

    def scale_grad(self, f: float):
        """deferred scalar on the gradient (folded into the optimizer kernel), e.g. 1/world after all-reduce(SUM)"""
        self._gscale = getattr(self, "_gscale", 1.0) * f

    def lr_factor(self, name: str, lr_decay_rate: float, num_layers: int) -> float:
        """detectron2 `get_vit_lr_decay_rate` (ViTDet layer-wise lr decay) as the reference enables it for
        build_vitdet_b_backbone (aldi/backbone.py:73-79, aldi/trainer.py:204): blocks.i -> rate^(num_layers - i),
        pos_embed / patch_embed -> rate^(num_layers + 1), everything outside the ViT (pyramid, heads) -> 1."""
        layer_id = num_layers + 1
        if name.startswith("backbone"):
            if ".pos_embed" in name or ".patch_embed" in name:
                layer_id = 0
            elif ".blocks." in name and ".residual." not in name:
                layer_id = int(name[name.find(".blocks."):].split(".")[2]) + 1
        return lr_decay_rate ** (num_layers + 1 - layer_id)

    def lr_groups(self, lr_decay_rate: Optional[float], num_layers: int) -> List[Tuple[int, int, bool, float]]:
        """contiguous (lo, hi, decayed?, lr factor) pieces of the flat layout, adjacent tensors with equal settings merged"""
        key = (lr_decay_rate, num_layers)
        cache = self.__dict__.setdefault("_lr_groups", {})
        if key not in cache:
            nd = self.n_decay
            starts = sorted((self.off[name], 1.0 if lr_decay_rate is None else self.lr_factor(name, lr_decay_rate, num_layers)) for name in self.spec)
            pieces = []
            for i, (lo, f) in enumerate(starts):                  # a tensor owns everything up to the next tensor (layout padding included)
                hi = starts[i + 1][0] if i + 1 < len(starts) else self.n
                lo = 0 if i == 0 else lo
                for a, b in ((lo, min(hi, nd)), (max(lo, nd), hi)):   # never straddle the decayed / non-decayed boundary
                    if b > a:
                        if pieces and pieces[-1][1] == a and pieces[-1][2] == (a < nd) and pieces[-1][3] == f:
                            pieces[-1][1] = b
                        else:
                            pieces.append([a, b, a < nd, f])
            cache[key] = [tuple(p) for p in pieces]
        return cache[key]

    def adamw_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.1, grad_scale: float = 1.0,
                   lr_decay_rate: Optional[float] = None, num_layers: int = 12):
        """torch.optim.AdamW over [decayed | weight_decay = 0 (norms, pos_embed)] -- detectron2 get_default_optimizer_params with
        weight_decay_norm = 0 and the pos_embed override of aldi/backbone.py:80; `lr_decay_rate` turns on the layer-wise lr
        decay (one launch per run of equal lr factor: 28 for ViT-B instead of 2)."""
        if self._m is None:
            self._m = torch.zeros(self.n, dtype=torch.float32, device=self.device)
            self._v = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.step_count += 1
        for lo, hi, dec, f in self.lr_groups(lr_decay_rate, num_layers):
            if hi > lo:
                V.adamw_step(self.master[lo:hi], self.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], self.compute[lo:hi], lr=lr * f, betas=betas,
                             eps=eps, weight_decay=weight_decay if dec else 0.0, step=self.step_count,
                             grad_scale=grad_scale * getattr(self, "_gscale", 1.0))
        if self._wt_plan is not None:
            self._wt_plan.run()

    def ema_from(self, student: "VitParams", alpha: float, copy_only: bool = False):
        ops.ema_update(self.master, student.master, None, self.n, alpha, copy_only, torch.float32)
        self.refresh()


def window_maps(N: int, gh: int, gw: int, ws: int, device):
    """detectron2 window_partition as a row gather: (win_map [nW*ws*ws] -> token or -1, inv_map [N*gh*gw] -> window row, nW)"""
    ph, pw = (ws - gh % ws) % ws, (ws - gw % ws) % ws
    Hp, Wp = gh + ph, gw + pw
    idx = torch.full((N, Hp, Wp), -1, dtype=torch.int32)
    idx[:, :gh, :gw] = torch.arange(N * gh * gw, dtype=torch.int32).view(N, gh, gw)
    win = idx.view(N, Hp // ws, ws, Wp // ws, ws).permute(0, 1, 3, 2, 4).reshape(-1).contiguous()
    inv = torch.empty(N * gh * gw, dtype=torch.int32)
    valid = win >= 0
    inv[win[valid].long()] = torch.nonzero(valid).flatten().to(torch.int32)
    return win.to(device), inv.to(device), N * (Hp // ws) * (Wp // ws)


class Ctx(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class ViT:
    """forward(img_u8, sizes) -> tokens [N*gh*gw, embed] (== the NHWC feature map [N, gh, gw, embed]); backward(ctx, g)."""

    def __init__(self, params: VitParams):
        self.p = params
        self.cfg = params.cfg
        self.device = params.device
        self._geom: Dict[Tuple[int, int, int], dict] = {}
        self.grad_ready = None           # callback(ranges): data-parallel overlapped exchange (reduce.BucketedReducer.ready)

    MAX_GEOMETRIES = 4            # multi-scale training sees many (N, gh, gw): each holds ~0.1 GB of attention operands per block

    def geometry(self, N: int, gh: int, gw: int) -> dict:
        key = (N, gh, gw)
        if key in self._geom:
            self._geom[key] = self._geom.pop(key)                     # most recently used last
            return self._geom[key]
        c = self.cfg
        win, inv, nW = window_maps(N, gh, gw, c.window, self.device)
        self._geom[key] = dict(win=win, inv=inv, nW=nW, att={},
                               att_w=V.Attention(nW, c.window, c.window, c.heads, self.device),
                               att_g=V.Attention(N, gh, gw, c.heads, self.device))
        while len(self._geom) > self.MAX_GEOMETRIES:                  # least recently used geometry (and its workspaces) goes
            self._geom.pop(next(iter(self._geom)))
        return self._geom[key]

    def _attention(self, geo: dict, i: int, N: int, gh: int, gw: int, save: bool) -> V.Attention:
        """inference shares one workspace per geometry; a training pass keeps one per block so that backward finds the
        operands (Q', K', their transposes) still in place instead of rebuilding them (~0.1 GB per block, HBM is 288 GB)"""
        glob = i in self.cfg.global_blocks
        if not save:
            return geo["att_g"] if glob else geo["att_w"]
        if i not in geo["att"]:
            c = self.cfg
            geo["att"][i] = V.Attention(N, gh, gw, c.heads, self.device) if glob else V.Attention(geo["nW"], c.window, c.window, c.heads, self.device)
        return geo["att"][i]

    def _linear(self, x, name, res=None):
        y = ops.conv2d(x.view(x.shape[0], 1, 1, x.shape[1]), self.p.lin_w(name + ".weight"), shift=self.p.m(name + ".bias"),
                       res=None if res is None else res.view(res.shape[0], 1, 1, res.shape[1]), res_mode=0 if res is None else 1)
        return y.view(x.shape[0], -1)

    def _linear_bwd(self, x, g, name, need_dx=True):
        """accumulates d(weight), d(bias); returns dx"""
        T = x.shape[0]
        x4, g4 = x.view(T, 1, 1, -1), g.view(T, 1, 1, -1)
        ops.conv_wgrad(x4, g4, self.p.g(name + ".weight"), KH=1, KW=1)
        ops.bias_grad(g, self.p.g(name + ".bias"))
        if not need_dx:
            return None
        return ops.conv2d(g4, self.p.wt(name + ".weight")).view(T, -1)

    def _rel_tables(self, i: int, gh: int, gw: int):
        c = self.cfg
        rh, rw = self.p.m(f"blocks.{i}.attn.rel_pos_h"), self.p.m(f"blocks.{i}.attn.rel_pos_w")
        if i in c.global_blocks:
            th = rh if rh.shape[0] == 2 * gh - 1 else V.linear_resize(rh, 2 * gh - 1)
            tw = rw if rw.shape[0] == 2 * gw - 1 else V.linear_resize(rw, 2 * gw - 1)
            return th, tw
        return rh, rw

    def drop_path_scales(self, N: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
        """per (block, branch, sample) multipliers of stochastic depth (0 or 1/keep), rates linspace(0, drop_path_rate, depth).
        The reference draws them with torch.rand on the GPU inside each block; here they come from the host generator
        (same distribution, different stream -- documented in DESIGN.md)."""
        c = self.cfg
        if c.drop_path_rate <= 0:
            return None
        rates = torch.linspace(0, c.drop_path_rate, c.depth)
        keep = (1.0 - rates).view(-1, 1, 1)
        u = torch.rand((c.depth, 2, N), generator=generator)
        return (torch.floor(keep + u) / keep).to(torch.float32)

    # ------------------------------------------------------------------------------------------------- forward
    def forward(self, img_u8: torch.Tensor, sizes: Sequence[Sequence[int]], save: bool = True,
                drop_scales: Optional[torch.Tensor] = None, hw_dev: Optional[torch.Tensor] = None) -> Ctx:
        c, p = self.cfg, self.p
        N, _, Hs, Ws = img_u8.shape
        gh, gw = Hs // c.patch, Ws // c.patch
        T, E = N * gh * gw, c.embed
        geo = self.geometry(N, gh, gw)
        # (the fused step hands over its persistent [N][2] image-size buffer: a per-call pinned upload cannot be recorded into a graph)
        hw = hw_dev.view(-1) if hw_dev is not None and hw_dev.numel() == 2 * len(sizes) else \
            ops.upload_packed([torch.tensor([[int(h), int(w)] for h, w in sizes], dtype=torch.int32).flatten()], self.device)[0]
        ctx = Ctx(N=N, gh=gh, gw=gw, blocks=[], save=save)
        patches = V.patchify(img_u8, hw, c.patch, c.pixel_mean, c.pixel_std, torch.bfloat16)
        tok = ops.conv2d(patches.view(T, 1, 1, -1), p.w("patch_embed.proj.weight", (E, 1, 1, 3 * c.patch * c.patch)),
                         shift=p.m("patch_embed.proj.bias")).view(T, E)
        G = c.pretrain_grid
        pos = V.bicubic_resize(p.m("pos_embed")[0, 1:].view(G, G, E), gh, gw) if (G, G) != (gh, gw) else p.m("pos_embed")[0, 1:].view(gh, gw, E)
        x = V.add_pos(tok, pos.contiguous(), N)
        if save:
            ctx.patches = patches
        ds = drop_scales.to(self.device) if drop_scales is not None else None
        for i in range(c.depth):
            glob = i in c.global_blocks
            b = f"blocks.{i}."
            att = self._attention(geo, i, N, gh, gw, save)
            y1, mean1, rstd1 = V.layernorm_forward(x, p.m(b + "norm1.weight"), p.m(b + "norm1.bias"), eps=c.ln_eps,
                                                   row_map=None if glob else geo["win"])
            qkv = self._linear(y1, b + "attn.qkv")
            th, tw = self._rel_tables(i, gh, gw)
            O, lse = att.forward(qkv, th, tw)
            att_ver = att.version
            s1 = ds[i, 0] if ds is not None else None
            if glob and s1 is None:
                x1 = self._linear(O, b + "attn.proj", res=x)
            else:
                proj = self._linear(O, b + "attn.proj")
                x1 = V.rows_add(x, proj, rows=T, row_map=None if glob else geo["inv"], scale=s1, rows_per_sample=gh * gw)
            y2, mean2, rstd2 = V.layernorm_forward(x1, p.m(b + "norm2.weight"), p.m(b + "norm2.bias"), eps=c.ln_eps)
            h1 = self._linear(y2, b + "mlp.fc1")
            a1 = V.gelu(h1)
            s2 = ds[i, 1] if ds is not None else None
            if s2 is None:
                x2 = self._linear(a1, b + "mlp.fc2", res=x1)
            else:
                x2 = V.rows_add(x1, self._linear(a1, b + "mlp.fc2"), rows=T, scale=s2, rows_per_sample=gh * gw)
            if save:
                ctx.blocks.append(Ctx(x=x, y1=y1, mean1=mean1, rstd1=rstd1, qkv=qkv, O=O, lse=lse, x1=x1, y2=y2, mean2=mean2, rstd2=rstd2,
                                      h1=h1, a1=a1, s1=s1, s2=s2, att_ver=att_ver))
            x = x2
        ctx.out = x
        return ctx

    # ------------------------------------------------------------------------------------------------- backward
    def backward(self, ctx: Ctx, g: torch.Tensor) -> None:
        """g = d loss / d tokens [N*gh*gw, embed] (bf16); parameter gradients accumulate into params.grad."""
        c, p = self.cfg, self.p
        N, gh, gw = ctx.N, ctx.gh, ctx.gw
        T, E = N * gh * gw, c.embed
        geo = self.geometry(N, gh, gw)
        for i in reversed(range(c.depth)):
            glob = i in c.global_blocks
            b = f"blocks.{i}."
            s = ctx.blocks[i]
            att = self._attention(geo, i, N, gh, gw, True)
            # ---- MLP branch
            df2 = g if s.s2 is None else V.rows_add(None, g, rows=T, scale=s.s2, rows_per_sample=gh * gw)
            da1 = self._linear_bwd(s.a1, df2, b + "mlp.fc2")
            dh1 = V.gelu_backward(s.h1, da1)
            dy2 = self._linear_bwd(s.y2, dh1, b + "mlp.fc1")
            dx1 = V.layernorm_backward(dy2, s.x1, p.m(b + "norm2.weight"), s.mean2, s.rstd2, p.g(b + "norm2.weight"), p.g(b + "norm2.bias"),
                                       res=g)
            # ---- attention branch
            if glob and s.s1 is None:
                dproj = dx1
            else:
                rows = T if glob else geo["win"].numel()
                dproj = V.rows_add(None, dx1, rows=rows, row_map=None if glob else geo["win"], scale=None, rows_per_sample=1)
                if s.s1 is not None:
                    # the per-sample scale follows the SOURCE token's image: scale in token order first
                    dproj = V.rows_add(None, V.rows_add(None, dx1, rows=T, scale=s.s1, rows_per_sample=gh * gw), rows=rows,
                                       row_map=None if glob else geo["win"])
            dO = self._linear_bwd(s.O, dproj, b + "attn.proj")
            th, tw = self._rel_tables(i, gh, gw)
            rh_m, rw_m = p.m(b + "attn.rel_pos_h"), p.m(b + "attn.rel_pos_w")
            resized_h, resized_w = th.data_ptr() != rh_m.data_ptr(), tw.data_ptr() != rw_m.data_ptr()
            dth = torch.zeros_like(th) if resized_h else p.g(b + "attn.rel_pos_h")
            dtw = torch.zeros_like(tw) if resized_w else p.g(b + "attn.rel_pos_w")
            dqkv = att.backward(s.qkv, th, tw, s.O, s.lse, dO, dth, dtw, prepared=(att.version == s.att_ver))
            if resized_h:
                V.linear_resize_backward(dth, p.g(b + "attn.rel_pos_h"))
            if resized_w:
                V.linear_resize_backward(dtw, p.g(b + "attn.rel_pos_w"))
            dy1 = self._linear_bwd(s.y1, dqkv, b + "attn.qkv")
            g = V.layernorm_backward(dy1, s.x, p.m(b + "norm1.weight"), s.mean1, s.rstd1, p.g(b + "norm1.weight"), p.g(b + "norm1.bias"),
                                     row_map=None if glob else geo["win"], res=dx1)
            if self.grad_ready is not None:                     # this block's gradients are final: the exchange may start on them
                self.grad_ready(p.ranges([n for n in p.spec if n.startswith(f"{c.prefix}blocks.{i}.")]))
        # ---- embeddings
        G = c.pretrain_grid
        dpos = V.sum_batch(g, N).view(gh, gw, E)
        gpos = p.g("pos_embed")[0, 1:].view(G, G, E)
        if (G, G) != (gh, gw):
            V.bicubic_resize_backward(dpos, gpos)
        else:                                                          # same grid as pre-training: plain accumulation (a kernel, not torch arithmetic)
            V.rows_add(gpos.reshape(-1, E), dpos.reshape(-1, E), rows=gh * gw, out=gpos.reshape(-1, E))
        self._linear_bwd_patch(ctx.patches, g)
        if self.grad_ready is not None:
            self.grad_ready(p.ranges([c.prefix + "pos_embed", c.prefix + "patch_embed.proj.weight", c.prefix + "patch_embed.proj.bias"]))

    def _linear_bwd_patch(self, patches, g):
        T = patches.shape[0]
        ops.conv_wgrad(patches.view(T, 1, 1, -1), g.view(T, 1, 1, -1), self.p.g("patch_embed.proj.weight"), KH=1, KW=1)
        ops.bias_grad(g, self.p.g("patch_embed.proj.bias"))


class SimpleFeaturePyramid:
    """detectron2 SimpleFeaturePyramid(scale_factors=(4, 2, 1, 0.5), norm="LN", top_block=LastLevelMaxPool) on the ViT map
    [N, gh, gw, E] -> p2..p6 (NHWC, `fpn_channels` wide).  ConvTranspose2d(k=2, s=2) runs as four 1x1 igemm launches that scatter
    their tap to (2y+dy, 2x+dx); its data gradient is the adjoint 2x2/2 convolution, its weight gradient that convolution's."""

    def __init__(self, params: VitParams):
        assert params.cfg.sfp
        self.p = params
        self.cfg = params.cfg
        self._fw: Dict[str, Tuple[int, torch.Tensor]] = {}

    def _deconv_fw(self, name: str) -> torch.Tensor:
        """[Cin,2,2,Cout] (the adjoint conv's layout, which is what the optimizer owns) -> per-tap [2,2,Cout,Cin]; pure data
        movement, refreshed when the optimizer has stepped"""
        ver, t = self._fw.get(name, (-1, None))
        if ver != self.p.step_count or t is None:
            t = self.p.w(name + ".weight").permute(1, 2, 3, 0).contiguous()
            self._fw[name] = (self.p.step_count, t)
        return t

    def _deconv(self, x, name):
        N, H, W, Cin = x.shape
        fw = self._deconv_fw(name)
        Cout = fw.shape[2]
        out = torch.empty((N, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
        flat = out.view(-1)
        for dy in (0, 1):
            for dx in (0, 1):
                ops.conv2d(x, fw[dy, dx].view(Cout, 1, 1, Cin), shift=self.p.m(name + ".bias"), out=flat[(dy * 2 * W + dx) * Cout:],
                           out_scale=2, out_hw=(2 * H, 2 * W))
        return out

    def _deconv_bwd(self, x, g, name):
        p = self.p
        ops.conv_wgrad(g, x, p.g(name + ".weight"), KH=2, KW=2, stride=2, pad=0)
        ops.bias_grad(g.view(-1, g.shape[-1]), p.g(name + ".bias"))
        return ops.conv2d(g, p.w(name + ".weight"), stride=2, pad=0)

    def _conv_ln(self, x, name, k, save):
        p = self.p
        y = ops.conv2d(x, p.w(name + ".weight"), pad=k // 2)
        C = y.shape[-1]
        yn, mean, rstd = V.layernorm_forward(y.view(-1, C), p.m(name + ".norm.weight"), p.m(name + ".norm.bias"), eps=self.cfg.ln_eps)
        if save is not None:
            save.append((name, k, x, y, mean, rstd))
        return yn.view(y.shape)

    def _conv_ln_bwd(self, rec, g):
        name, k, x, y, mean, rstd = rec
        p = self.p
        C = y.shape[-1]
        dy = V.layernorm_backward(g.reshape(-1, C), y.view(-1, C), p.m(name + ".norm.weight"), mean, rstd, p.g(name + ".norm.weight"),
                                  p.g(name + ".norm.bias")).view(y.shape)
        ops.conv_wgrad(x, dy, p.g(name + ".weight"), KH=k, KW=k, stride=1, pad=k // 2)
        return ops.conv2d(dy, p.wt(name + ".weight"), pad=k // 2)

    def forward(self, x: torch.Tensor, save: bool = True) -> Ctx:
        """x [N, gh, gw, E] bf16"""
        p, q = self.p, "backbone."
        c = Ctx(x=x, recs={}, save=save)
        rec = lambda lvl: c.recs.setdefault(lvl, []) if save else None
        # stride 4
        d1 = self._deconv(x, q + "simfp_2.0")
        C1 = d1.shape[-1]
        n1, mean1, rstd1 = V.layernorm_forward(d1.view(-1, C1), p.m(q + "simfp_2.1.weight"), p.m(q + "simfp_2.1.bias"), eps=self.cfg.ln_eps)
        a1 = V.gelu(n1).view(d1.shape)
        d2 = self._deconv(a1, q + "simfp_2.3")
        p2 = self._conv_ln(self._conv_ln(d2, q + "simfp_2.4", 1, rec(2)), q + "simfp_2.5", 3, rec(2))
        # stride 8
        e1 = self._deconv(x, q + "simfp_3.0")
        p3 = self._conv_ln(self._conv_ln(e1, q + "simfp_3.1", 1, rec(3)), q + "simfp_3.2", 3, rec(3))
        # stride 16
        p4 = self._conv_ln(self._conv_ln(x, q + "simfp_4.0", 1, rec(4)), q + "simfp_4.1", 3, rec(4))
        # stride 32 (+ LastLevelMaxPool: p6 = p5[:, ::2, ::2])
        mp, mp_idx = V.maxpool2(x)
        p5 = self._conv_ln(self._conv_ln(mp, q + "simfp_5.1", 1, rec(5)), q + "simfp_5.2", 3, rec(5))
        p6 = ops.subsample2(p5)
        c.P = [p2, p3, p4, p5, p6]
        if save:
            c.update(d1=d1, n1=n1, mean1=mean1, rstd1=rstd1, a1=a1, d2=d2, e1=e1, mp_idx=mp_idx)
        return c

    def backward(self, c: Ctx, gP: Sequence[torch.Tensor]) -> torch.Tensor:
        """gP = gradients wrt p2..p5 (p6's already folded into p5's), bf16 NHWC -> gradient wrt x"""
        p, q = self.p, "backbone."
        x = c.x
        N, gh, gw, E = x.shape
        # stride 32
        g = self._conv_ln_bwd(c.recs[5][0], self._conv_ln_bwd(c.recs[5][1], gP[3]))
        dx = V.maxpool2_backward(g, c.mp_idx, gh, gw)
        # stride 16
        g = self._conv_ln_bwd(c.recs[4][0], self._conv_ln_bwd(c.recs[4][1], gP[2]))
        dx = V.rows_add(dx.view(-1, E), g.view(-1, E), rows=N * gh * gw)
        # stride 8
        g = self._conv_ln_bwd(c.recs[3][0], self._conv_ln_bwd(c.recs[3][1], gP[1]))
        dx = V.rows_add(dx, self._deconv_bwd(x, g, q + "simfp_3.0").view(-1, E), rows=N * gh * gw)
        # stride 4
        g = self._conv_ln_bwd(c.recs[2][0], self._conv_ln_bwd(c.recs[2][1], gP[0]))
        da1 = self._deconv_bwd(c.a1, g, q + "simfp_2.3")
        C1 = da1.shape[-1]
        dn1 = V.gelu_backward(c.n1, da1.view(-1, C1))
        dd1 = V.layernorm_backward(dn1, c.d1.view(-1, C1), p.m(q + "simfp_2.1.weight"), c.mean1, c.rstd1, p.g(q + "simfp_2.1.weight"),
                                   p.g(q + "simfp_2.1.bias")).view(c.d1.shape)
        dx = V.rows_add(dx, self._deconv_bwd(x, dd1, q + "simfp_2.0").view(-1, E), rows=N * gh * gw)
        return dx
