"""ConvNeXt trunk on the HIP path (reference aldi/backbone.py:189-352; configs/Base-RCNN-ConvNeXt-FPN.yaml: ConvNeXt-L, depths
[3, 3, 27, 3], dims [192, 384, 768, 1536], drop path 0.2, layer scale 1e-6, outputs of all four stages).

Host-side mirror of the reference's `ConvNeXt.forward_features`: stem (4x4/4 conv + channel LayerNorm), three LayerNorm + 2x2/2
conv downsamplers, `ConvNextBlock`s (depthwise 7x7 -> LayerNorm -> Linear 4x -> GELU -> Linear -> layer scale -> stochastic
depth -> residual) and one output LayerNorm per stage.  Parameter names are the reference module's state_dict keys under
`backbone.bottom_up.`; the flat container is `vit.VitParams` (depthwise kernels stored [7,7,C], 2x2 conv kernels channel-last).
Everything numeric is a C-ABI call.  The FPN / heads / trainer wiring of the ConvNeXt-FPN detector is not built yet: this is the
trunk, pinned against the reference's own class (golden g10).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import ops
from . import vit_ops as V
from .vit import Ctx, VitParams


@dataclass
class ConvNeXtConfig:
    depths: Tuple[int, ...] = (3, 3, 27, 3)
    dims: Tuple[int, ...] = (192, 384, 768, 1536)
    drop_path_rate: float = 0.2
    layer_scale_init_value: float = 1e-6
    ln_eps: float = 1e-6
    prefix: str = "backbone.bottom_up."
    pool: int = 7
    num_classes: int = 0             # > 0: also hold FPN + the standard Faster R-CNN heads (build_convnext_fpn_backbone detector)
    fpn_channels: int = 256
    fc_dim: int = 1024
    num_anchors: int = 3
    anchor_sizes: Tuple[int, ...] = (64, 128, 256, 512, 1024)      # configs/Base-RCNN-ConvNeXt-FPN.yaml:18
    pixel_mean: Tuple[float, float, float] = (103.530, 116.280, 123.675)      # configs/Base-RCNN-ConvNeXt-FPN.yaml keeps detectron2's defaults
    pixel_std: Tuple[float, float, float] = (1.0, 1.0, 1.0)

    def init_value(self, name, shape):
        """the reference's initial values where they are not a random matrix (aldi/backbone.py:207,287-290): LayerNorm weights 1,
        layer scale gamma = LAYER_SCALE_INIT_VALUE (biases 0 and trunc_normal(0.02) matrices are VitParams.init_random's defaults)"""
        if name.startswith(self.prefix) and name.endswith("gamma"):
            return torch.full(shape, float(self.layer_scale_init_value))
        if name.startswith(self.prefix) and len(shape) == 1 and name.endswith(".weight"):
            return torch.ones(shape)
        return None

    def packs(self):
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

    def detector_spec(self):
        """detectron2 FPN (fuse sum, no norm, LastLevelMaxPool) on the four stage outputs + StandardRPNHead + FastRCNNConvFCHead(2 FC)
        + FastRCNNOutputLayers, as `build_convnext_fpn_backbone` (aldi/backbone.py:373-392) and Base-RCNN-ConvNeXt-FPN.yaml assemble"""
        C, K, A, d = self.fpn_channels, self.num_classes, self.num_anchors, self.dims
        s = OrderedDict()
        for lvl in (2, 3, 4, 5):
            s[f"backbone.fpn_lateral{lvl}.weight"] = ((C, d[lvl - 2], 1, 1), True)
            s[f"backbone.fpn_lateral{lvl}.bias"] = ((C,), True)
            s[f"backbone.fpn_output{lvl}.weight"] = ((C, C, 3, 3), True)
            s[f"backbone.fpn_output{lvl}.bias"] = ((C,), True)
        rp = "proposal_generator.rpn_head."
        s[rp + "conv.weight"] = ((C, C, 3, 3), True)
        s[rp + "conv.bias"] = ((C,), True)
        s[rp + "objectness_logits.weight"] = ((A, C, 1, 1), True)
        s[rp + "anchor_deltas.weight"] = ((4 * A, C, 1, 1), True)
        s[rp + "objectness_logits.bias"] = ((A,), True)
        s[rp + "anchor_deltas.bias"] = ((4 * A,), True)
        bh = "roi_heads.box_head."
        s[bh + "fc1.weight"] = ((self.fc_dim, C * self.pool * self.pool), True)
        s[bh + "fc1.bias"] = ((self.fc_dim,), True)
        s[bh + "fc2.weight"] = ((self.fc_dim, self.fc_dim), True)
        s[bh + "fc2.bias"] = ((self.fc_dim,), True)
        bp = "roi_heads.box_predictor."
        s[bp + "cls_score.weight"] = ((K + 1, self.fc_dim), True)
        s[bp + "bbox_pred.weight"] = ((4 * K, self.fc_dim), True)
        s[bp + "cls_score.bias"] = ((K + 1,), True)
        s[bp + "bbox_pred.bias"] = ((4 * K,), True)
        return s

    def spec(self):
        """reference module state_dict order (aldi/backbone.py:239-285); weight decay on everything (the reference's AdamW setup
        for this trunk is not reproduced yet)"""
        p, d = self.prefix, self.dims
        s = OrderedDict()
        s[p + "downsample_layers.0.0.weight"] = ((d[0], 3, 4, 4), True)
        s[p + "downsample_layers.0.0.bias"] = ((d[0],), True)
        s[p + "downsample_layers.0.1.weight"] = ((d[0],), True)
        s[p + "downsample_layers.0.1.bias"] = ((d[0],), True)
        for i in range(1, 4):
            s[f"{p}downsample_layers.{i}.0.weight"] = ((d[i - 1],), True)
            s[f"{p}downsample_layers.{i}.0.bias"] = ((d[i - 1],), True)
            s[f"{p}downsample_layers.{i}.1.weight"] = ((d[i], d[i - 1], 2, 2), True)
            s[f"{p}downsample_layers.{i}.1.bias"] = ((d[i],), True)
        for i in range(4):
            for j in range(self.depths[i]):
                b = f"{p}stages.{i}.{j}."
                s[b + "gamma"] = ((d[i],), True)
                s[b + "dwconv.weight"] = ((d[i], 1, 7, 7), True)
                s[b + "dwconv.bias"] = ((d[i],), True)
                s[b + "norm.weight"] = ((d[i],), True)
                s[b + "norm.bias"] = ((d[i],), True)
                s[b + "pwconv1.weight"] = ((4 * d[i], d[i]), True)
                s[b + "pwconv1.bias"] = ((4 * d[i],), True)
                s[b + "pwconv2.weight"] = ((d[i], 4 * d[i]), True)
                s[b + "pwconv2.bias"] = ((d[i],), True)
        for i in range(4):
            s[f"{p}norm{i}.weight"] = ((d[i],), True)
            s[f"{p}norm{i}.bias"] = ((d[i],), True)
        if self.num_classes > 0:
            s.update(self.detector_spec())
        return s


class ConvNeXt:
    """forward(img_u8, sizes) -> Ctx with .outs = [stage 0..3 feature maps, NHWC, after norm{i}]; backward(ctx, grads)."""

    def __init__(self, params: VitParams):
        self.p, self.cfg, self.device = params, params.cfg, params.device

    def drop_path_scales(self, N: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
        """[sum(depths), N] multipliers (0 or 1/keep), rates linspace(0, drop_path_rate, sum(depths)) as in aldi/backbone.py:270;
        host-drawn (see ViT.drop_path_scales)"""
        c = self.cfg
        if c.drop_path_rate <= 0:
            return None
        rates = torch.linspace(0, c.drop_path_rate, sum(c.depths))
        keep = (1.0 - rates).view(-1, 1)
        return (torch.floor(keep + torch.rand((sum(c.depths), N), generator=generator)) / keep).to(torch.float32)

    def _ln(self, x, name, save=None):
        p = self.p
        C = x.shape[-1]
        y, mean, rstd = V.layernorm_forward(x.reshape(-1, C), p.m(name + ".weight"), p.m(name + ".bias"), eps=self.cfg.ln_eps)
        if save is not None:
            save[name] = (x, mean, rstd)
        return y.view(x.shape)

    def _ln_bwd(self, rec, name, g, res=None):
        x, mean, rstd = rec[name]
        p = self.p
        C = x.shape[-1]
        return V.layernorm_backward(g.reshape(-1, C), x.reshape(-1, C), p.m(name + ".weight"), mean, rstd, p.g(name + ".weight"), p.g(name + ".bias"),
                                    res=None if res is None else res.reshape(-1, C)).view(x.shape)

    def _linear(self, x2d, name):
        return ops.conv2d(x2d.view(x2d.shape[0], 1, 1, -1), self.p.lin_w(name + ".weight"), shift=self.p.m(name + ".bias")).view(x2d.shape[0], -1)

    def _linear_bwd(self, x2d, g2d, name, need_dx=True):
        T = x2d.shape[0]
        ops.conv_wgrad(x2d.view(T, 1, 1, -1), g2d.view(T, 1, 1, -1), self.p.g(name + ".weight"), KH=1, KW=1)
        ops.bias_grad(g2d, self.p.g(name + ".bias"))
        return ops.conv2d(g2d.view(T, 1, 1, -1), self.p.wt(name + ".weight")).view(T, -1) if need_dx else None

    def forward(self, img_u8: torch.Tensor, sizes: Sequence[Sequence[int]], save: bool = True, drop_scales: Optional[torch.Tensor] = None,
                hw_dev: Optional[torch.Tensor] = None) -> Ctx:
        c, p = self.cfg, self.p
        N, _, Hs, Ws = img_u8.shape
        # (the fused step hands over its persistent [N][2] image-size buffer: a per-call pinned upload cannot be recorded into a graph)
        hw = hw_dev.view(-1) if hw_dev is not None and hw_dev.numel() == 2 * len(sizes) else \
            ops.upload_packed([torch.tensor([[int(h), int(w)] for h, w in sizes], dtype=torch.int32).flatten()], self.device)[0]
        ctx = Ctx(N=N, save=save, rec={}, blocks=[])
        rec = ctx.rec if save else None
        pre = "downsample_layers."
        patches = V.patchify(img_u8, hw, 4, c.pixel_mean, c.pixel_std, torch.bfloat16)             # [N*H/4*W/4, 48] in (c, ph, pw) order
        H, W = Hs // 4, Ws // 4
        x = ops.conv2d(patches.view(-1, 1, 1, 48), p.w(pre + "0.0.weight", (c.dims[0], 1, 1, 48)), shift=p.m(pre + "0.0.bias")).view(N, H, W, c.dims[0])
        x = self._ln(x, pre + "0.1", rec)
        if save:
            ctx.patches = patches
        ds = drop_scales.to(self.device) if drop_scales is not None else None
        outs, k = [], 0
        for i in range(4):
            if i > 0:
                xn = self._ln(x, f"{pre}{i}.0", rec)
                x = ops.conv2d(xn, p.w(f"{pre}{i}.1.weight"), stride=2, pad=0, shift=p.m(f"{pre}{i}.1.bias"))
                if save:
                    ctx.rec[f"ds{i}"] = xn
                H, W = H // 2, W // 2
            C = c.dims[i]
            for j in range(c.depths[i]):
                b = f"stages.{i}.{j}."
                d = V.dwconv7(x, p.w(b + "dwconv.weight"), p.m(b + "dwconv.bias"))
                dn = self._ln(d, b + "norm", rec)
                h1 = self._linear(dn.view(-1, C), b + "pwconv1")
                a1 = V.gelu(h1)
                y = self._linear(a1, b + "pwconv2")
                s_ = ds[k] if ds is not None else None
                xo = V.scale_add(x.reshape(-1, C), y, p.m(b + "gamma"), s_, H * W).view(x.shape)
                if save:
                    ctx.blocks.append(Ctx(x=x, dn=dn, h1=h1, a1=a1, y=y, s=s_, name=b, hw=(H, W)))
                x = xo
                k += 1
            outs.append(self._ln(x, f"norm{i}", rec))
        ctx.outs = outs
        return ctx

    def backward(self, ctx: Ctx, grads: Sequence[Optional[torch.Tensor]]) -> None:
        """grads[i] = d loss / d outs[i] (bf16 NHWC, or None); parameter gradients accumulate into params.grad"""
        c, p = self.cfg, self.p
        N = ctx.N
        pre = "downsample_layers."
        g = None                                                        # gradient w.r.t. the stage's (pre-norm) output
        bi = len(ctx.blocks) - 1
        for i in (3, 2, 1, 0):
            C = c.dims[i]
            if grads[i] is not None:
                g = self._ln_bwd(ctx.rec, f"norm{i}", grads[i], res=g)
            assert g is not None, "no gradient reaches stage %d" % i
            for j in reversed(range(c.depths[i])):
                blk = ctx.blocks[bi]
                bi -= 1
                b = blk.name
                H, W = blk.hw
                g2 = g.reshape(-1, C)
                dy = V.scale_add_backward(g2, blk.y, p.m(b + "gamma"), blk.s, p.g(b + "gamma"), H * W)
                da1 = self._linear_bwd(blk.a1, dy, b + "pwconv2")
                dh1 = V.gelu_backward(blk.h1, da1)
                ddn = self._linear_bwd(blk.dn.view(-1, C), dh1, b + "pwconv1")
                dd = self._ln_bwd(ctx.rec, b + "norm", ddn.view(blk.x.shape))
                V.dwconv7_wgrad(blk.x, dd, p.g(b + "dwconv.weight"))
                ops.bias_grad(dd.view(-1, C), p.g(b + "dwconv.bias"))
                g = V.rows_add(g2, V.dwconv7(dd, p.w(b + "dwconv.weight"), None, flip=True).view(-1, C), rows=g2.shape[0]).view(blk.x.shape)
            if i > 0:
                xn = ctx.rec[f"ds{i}"]
                ops.conv_wgrad(xn, g, p.g(f"{pre}{i}.1.weight"), KH=2, KW=2, stride=2, pad=0)
                ops.bias_grad(g.view(-1, C), p.g(f"{pre}{i}.1.bias"))
                Hin, Win = xn.shape[1], xn.shape[2]
                gx = torch.empty_like(xn)
                wt = p.wt(f"{pre}{i}.1.weight")                      # [Cin, 2, 2, Cout], taps rotated for the data gradient
                for dy_ in (0, 1):                                      # stride = kernel: every input pixel belongs to exactly one tap
                    for dx_ in (0, 1):
                        ops.conv2d(g, wt[:, 1 - dy_, 1 - dx_].contiguous().view(wt.shape[0], 1, 1, wt.shape[3]),
                                   out=gx.view(-1)[(dy_ * Win + dx_) * xn.shape[3]:], out_scale=2, out_hw=(Hin, Win))
                g = self._ln_bwd(ctx.rec, f"{pre}{i}.0", gx)
        # stem
        g = self._ln_bwd(ctx.rec, pre + "0.1", g)
        T = ctx.patches.shape[0]
        ops.conv_wgrad(ctx.patches.view(T, 1, 1, 48), g.reshape(T, 1, 1, -1), p.g(pre + "0.0.weight"), KH=1, KW=1)
        ops.bias_grad(g.reshape(T, -1), p.g(pre + "0.0.bias"))


def _engine_base():
    from .vitdet import FlatParamRCNN
    return FlatParamRCNN


class ConvNeXtRCNN(_engine_base()):
    """ConvNeXt-FPN Faster R-CNN on the HIP engine (reference aldi/backbone.py:373-392 + configs/Base-RCNN-ConvNeXt-FPN.yaml): only the
    trunk, the FPN and the heads' parameter plumbing are specific; anchors (sizes 64..1024), proposals, matching, sampling, ROIAlign,
    losses, distillation and the fused student pass are engine.RCNN's."""

    def __init__(self, params: VitParams, num_classes: int, seed: int = 0):
        self._init_flat(params, num_classes, seed)
        self.net = ConvNeXt(params)
        self.anchor_sizes = tuple(params.cfg.anchor_sizes)

    graph_safe = True            # (fused_step: the launches of a pass are the same every step)

    def refresh_drop_scales(self, N: int):
        from .vitdet import _refresh
        _refresh(self, self.net.drop_path_scales(N, self.drop_gen))

    # ------------------------------------------------------------------ forward
    def trunk(self, st_u8: torch.Tensor, sizes, save: bool) -> Ctx:
        N = st_u8.shape[0]
        from .vitdet import _staged_drop_scales
        ds = _staged_drop_scales(self, None, N) if save else None
        cn = self.net.forward(st_u8, sizes, save=save, drop_scales=ds, hw_dev=self.__dict__.get("_hw_dev", {}).get(st_u8.data_ptr()))
        cs = cn.outs
        prev, P = {}, {}
        prev[5] = ops.conv2d(cs[3], self._w("backbone.fpn_lateral5"), shift=self._b("backbone.fpn_lateral5"))
        P[5] = ops.conv2d(prev[5], self._w("backbone.fpn_output5"), pad=1, shift=self._b("backbone.fpn_output5"))
        for lvl in (4, 3, 2):
            prev[lvl] = ops.conv2d(cs[lvl - 2], self._w(f"backbone.fpn_lateral{lvl}"), shift=self._b(f"backbone.fpn_lateral{lvl}"),
                                   res=prev[lvl + 1], res_mode=2)
            P[lvl] = ops.conv2d(prev[lvl], self._w(f"backbone.fpn_output{lvl}"), pad=1, shift=self._b(f"backbone.fpn_output{lvl}"))
        c = Ctx()
        c.P = [P[2], P[3], P[4], P[5], ops.subsample2(P[5])]
        if save:
            c.cn_ctx, c.cs, c.prev = cn, cs, prev
        return c

    def rpn_head(self, c: Ctx, save: bool):
        rp = "proposal_generator.rpn_head."
        w_out, b_out = self._pack_w("rpn_head_out", self.vp.cfg.fpn_channels)
        heads, ts = [], []
        for f in c.P:
            t = ops.conv2d(f, self._w(rp + "conv"), pad=1, shift=self._b(rp + "conv"), relu=True)
            heads.append(ops.conv2d(t, w_out, shift=b_out, want_f32=True))
            if save:
                ts.append(t)
        c.head = heads
        if save:
            c.rpn_t = ts

    def box_head(self, pooled: torch.Tensor, c: Optional[Ctx] = None):
        cfg = self.vp.cfg
        R = pooled.shape[0]
        bh = "roi_heads.box_head."
        x = pooled.view(R, 1, 1, -1)
        fc1 = ops.conv2d(x, self.vp.w(bh + "fc1.weight", (cfg.fc_dim, 1, 1, x.shape[-1])), shift=self._b(bh + "fc1"), relu=True)
        fc2 = ops.conv2d(fc1, self.vp.lin_w(bh + "fc2.weight"), shift=self._b(bh + "fc2"), relu=True)
        w_out, b_out = self._pack_w("box_pred", cfg.fc_dim)
        pred = ops.conv2d(fc2, w_out, shift=b_out, want_f32=True).view(R, self.Cp)
        if c is not None:
            c.pooled, c.fc1, c.fc2 = pooled, fc1, fc2
        return pred, fc2

    # ------------------------------------------------------------------ backward
    def _backward_trunk(self, c: Ctx, align_list):
        assert not align_list, "adversarial alignment is not wired for the ConvNeXt trunk"
        cfg, vp, T, dev = self.vp.cfg, self.vp, torch.bfloat16, self.device
        C, bh, rp = cfg.fpn_channels, "roi_heads.box_head.", "proposal_generator.rpn_head."
        gP_roi = [(torch.empty if c.R > 0 else torch.zeros)(f.shape, dtype=torch.float32, device=dev) for f in c.P[:4]]
        if c.R > 0:
            gpred = ops.cast_from_f32(c.gpred[:c.R], T).view(c.R, 1, 1, self.Cp)
            self._wg_pack("box_pred", c.fc2, gpred, cfg.fc_dim)
            g_fc2 = ops.conv2d(gpred, self._pack_wt("box_pred", cfg.fc_dim), mask=c.fc2)
            ops.conv_wgrad(c.fc1, g_fc2, vp.g(bh + "fc2.weight"), KH=1, KW=1)
            ops.bias_grad(g_fc2.view(c.R, -1), vp.g(bh + "fc2.bias"))
            g_fc1 = ops.conv2d(g_fc2, vp.wt(bh + "fc2.weight"), mask=c.fc1)
            x = c.pooled.view(c.R, 1, 1, -1)
            ops.conv_wgrad(x, g_fc1, vp.g(bh + "fc1.weight"), KH=1, KW=1)
            ops.bias_grad(g_fc1.view(c.R, -1), vp.g(bh + "fc1.bias"))
            g_pooled = ops.conv2d(g_fc1, vp.wt(bh + "fc1.weight")).view(c.R, cfg.pool, cfg.pool, C)
            ops.roialign_backward(self.roi_feats(c, gP_roi), c.rois, c.R, cfg.pool, g_pooled, c.N, rois_sorted=True)
        gP = []
        wt_out = self._pack_wt("rpn_head_out", C)
        for l in range(5):
            gh = ops.cast_from_f32(c.ghead[l], T)
            self._wg_pack("rpn_head_out", c.rpn_t[l], gh, C)
            g_t = ops.conv2d(gh, wt_out, mask=c.rpn_t[l])
            self._wg(rp + "conv", c.P[l], g_t, 3)
            ops.bias_grad(g_t.view(-1, C), vp.g(rp + "conv.bias"))
            gP.append(ops.conv2d(g_t, vp.wt(rp + "conv.weight"), pad=1))
        ops.subsample2_bwd(gP[4], gP[3])
        for l in range(4):
            ops.add_f32(gP[l], gP_roi[l], gP[l])
        cb = getattr(self, "grad_ready", None)
        if cb is not None:
            cb(vp.ranges([n for n in vp.spec if n.startswith(("proposal_generator.", "roi_heads."))]))
        # FPN
        gprev = {}
        for i, lvl in enumerate((2, 3, 4, 5)):
            self._wg(f"backbone.fpn_output{lvl}", c.prev[lvl], gP[i], 3)
            ops.bias_grad(gP[i].view(-1, C), vp.g(f"backbone.fpn_output{lvl}.bias"))
            gprev[lvl] = ops.conv2d(gP[i], vp.wt(f"backbone.fpn_output{lvl}.weight"), pad=1)
        for lvl in (3, 4, 5):
            ops.upsample2_bwd(gprev[lvl - 1], gprev[lvl], accumulate=True)
        gc = []
        for lvl in (2, 3, 4, 5):
            self._wg(f"backbone.fpn_lateral{lvl}", c.cs[lvl - 2], gprev[lvl], 1)
            ops.bias_grad(gprev[lvl].view(-1, C), vp.g(f"backbone.fpn_lateral{lvl}.bias"))
            gc.append(ops.conv2d(gprev[lvl], vp.wt(f"backbone.fpn_lateral{lvl}.weight")))
        if cb is not None:
            cb(vp.ranges([n for n in vp.spec if n.startswith("backbone.fpn_")]))
        self.net.backward(c.cn_ctx, gc)
        if cb is not None:
            cb(vp.ranges([n for n in vp.spec if n.startswith(cfg.prefix)]))
