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
    pixel_mean: Tuple[float, float, float] = (103.530, 116.280, 123.675)      # configs/Base-RCNN-ConvNeXt-FPN.yaml keeps detectron2's defaults
    pixel_std: Tuple[float, float, float] = (1.0, 1.0, 1.0)

    def packs(self):
        return OrderedDict()

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

    def forward(self, img_u8: torch.Tensor, sizes: Sequence[Sequence[int]], save: bool = True, drop_scales: Optional[torch.Tensor] = None) -> Ctx:
        c, p = self.cfg, self.p
        N, _, Hs, Ws = img_u8.shape
        hw = ops.upload_packed([torch.tensor([[int(h), int(w)] for h, w in sizes], dtype=torch.int32).flatten()], self.device)[0]
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
