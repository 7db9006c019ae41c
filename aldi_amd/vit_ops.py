"""Torch-tensor wrappers over the ViTDet part of the C ABI (include/aldi_hip.h, "ViTDet trunk" section).

As in ops.py nothing is computed here: each function marshals pointers and sizes into one C-ABI call on torch's current
HIP stream.  Reference modules replaced: detectron2 modeling/backbone/vit.py + utils.py, driven by aldi/backbone.py:21-43.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L
from .ops import _p, dtype_code, stream_ptr


def layernorm_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, eps: float = 1e-6,
                      row_map: Optional[torch.Tensor] = None, relu: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x [rows_src, C] -> (y [rows, C], mean [rows], rstd [rows]); rows = len(row_map) if given (gather, -1 = zero row)."""
    Cc = x.shape[-1]
    rows = row_map.numel() if row_map is not None else x.numel() // Cc
    y = torch.empty((rows, Cc), dtype=x.dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.call("aldi_layernorm_forward", _p(x), _p(row_map), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, Cc, float(eps),
           int(relu), dtype_code(x.dtype), stream_ptr())
    return y, mean, rstd


def layernorm_backward(g: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                       dgamma: torch.Tensor, dbeta: torch.Tensor, *, row_map: Optional[torch.Tensor] = None,
                       res: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dx shaped like x (+ res); dgamma / dbeta (fp32) accumulate."""
    Cc = x.shape[-1]
    rows = g.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    L.call("aldi_layernorm_backward", _p(g), _p(x), _p(row_map), _p(gamma), _p(mean), _p(rstd), _p(res), _p(mask), _p(out), _p(dgamma), _p(dbeta),
           rows, Cc, dtype_code(x.dtype), stream_ptr())
    return out


def gelu(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    L.call("aldi_gelu", _p(x), None, _p(out), x.numel(), dtype_code(x.dtype), stream_ptr())
    return out


def gelu_backward(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    L.call("aldi_gelu", _p(x), _p(g), _p(out), x.numel(), dtype_code(x.dtype), stream_ptr())
    return out


def rows_add(a: Optional[torch.Tensor], b: torch.Tensor, *, rows: int, row_map: Optional[torch.Tensor] = None,
             scale: Optional[torch.Tensor] = None, rows_per_sample: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    Cc = b.shape[-1]
    if out is None:
        out = torch.empty((rows, Cc), dtype=b.dtype, device=b.device)
    L.call("aldi_rows_add", _p(a), _p(b), _p(row_map), _p(scale), _p(out), rows, Cc, rows_per_sample, dtype_code(b.dtype), stream_ptr())
    return out


def patchify(img_u8: torch.Tensor, hw: torch.Tensor, P: int, mean, std, dtype: torch.dtype) -> torch.Tensor:
    N, _, Hs, Ws = img_u8.shape
    out = torch.empty((N * (Hs // P) * (Ws // P), 3 * P * P), dtype=dtype, device=img_u8.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    L.call("aldi_patchify", _p(img_u8), _p(out), N, Hs, Ws, P, _p(hw), m, s, dtype_code(dtype), stream_ptr())
    return out


def linear_resize(table: torch.Tensor, L1: int) -> torch.Tensor:
    L0, Cc = table.shape
    out = torch.empty((L1, Cc), dtype=torch.float32, device=table.device)
    L.call("aldi_linear_resize", _p(table), _p(out), L0, L1, Cc, 0, stream_ptr())
    return out


def linear_resize_backward(g: torch.Tensor, dtable: torch.Tensor) -> None:
    L1, Cc = g.shape
    L.call("aldi_linear_resize", _p(g), _p(dtable), dtable.shape[0], L1, Cc, 1, stream_ptr())


def bicubic_resize(grid: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    S0h, S0w, Cc = grid.shape
    out = torch.empty((gh, gw, Cc), dtype=torch.float32, device=grid.device)
    L.call("aldi_bicubic_resize", _p(grid), _p(out), S0h, S0w, gh, gw, Cc, 0, stream_ptr())
    return out


def bicubic_resize_backward(g: torch.Tensor, dgrid: torch.Tensor) -> None:
    gh, gw, Cc = g.shape
    L.call("aldi_bicubic_resize", _p(g), _p(dgrid), dgrid.shape[0], dgrid.shape[1], gh, gw, Cc, 1, stream_ptr())


def add_pos(x: torch.Tensor, pos: torch.Tensor, N: int) -> torch.Tensor:
    y = torch.empty_like(x)
    L.call("aldi_add_pos", _p(x), _p(pos), _p(y), N, pos.numel(), dtype_code(x.dtype), stream_ptr())
    return y


def sum_batch(g: torch.Tensor, N: int) -> torch.Tensor:
    TC = g.numel() // N
    out = torch.empty(TC, dtype=torch.float32, device=g.device)
    L.call("aldi_sum_batch", _p(g), _p(out), N, TC, dtype_code(g.dtype), stream_ptr())
    return out


def maxpool2(x: torch.Tensor):
    N, H, W_, Cc = x.shape
    y = torch.empty((N, H // 2, W_ // 2, Cc), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, H // 2, W_ // 2, Cc), dtype=torch.uint8, device=x.device)
    L.call("aldi_maxpool2", _p(x), _p(y), _p(idx), N, H, W_, Cc, 0, dtype_code(x.dtype), stream_ptr())
    return y, idx


def maxpool2_backward(g: torch.Tensor, idx: torch.Tensor, H: int, W_: int) -> torch.Tensor:
    N, _, _, Cc = g.shape
    dx = torch.empty((N, H, W_, Cc), dtype=g.dtype, device=g.device)
    L.call("aldi_maxpool2", _p(g), _p(dx), _p(idx), N, H, W_, Cc, 1, dtype_code(g.dtype), stream_ptr())
    return dx


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, p_compute: Optional[torch.Tensor], *, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.1, step: int = 1, grad_scale: float = 1.0) -> None:
    dt = dtype_code(p_compute.dtype) if p_compute is not None else L.F32
    L.call("aldi_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(p_compute), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
           float(weight_decay), int(step), float(grad_scale), dt, stream_ptr())


class Attention:
    """Workspaces + argument block of one attention geometry (nB windows/images of gh x gw tokens, `heads` heads of 64)."""

    def __init__(self, nB: int, gh: int, gw: int, heads: int, device, rel: bool = True):
        self.nB, self.gh, self.gw, self.heads, self.rel = nB, gh, gw, heads, rel
        self.L = gh * gw
        self.Lp = (self.L + 63) // 64 * 64
        dq, tiled, vtc = C.c_int(), C.c_int(), C.c_long()
        L.call("aldi_attn_layout", gh, gw, int(rel), C.byref(dq), C.byref(tiled), C.byref(vtc))
        self.Dq, self.tiled = dq.value, bool(tiled.value)
        BH = nB * heads
        bf = dict(dtype=torch.bfloat16, device=device)
        self.Qp = torch.empty((BH, self.L, self.Dq), **bf)
        self.Kp = torch.empty((BH, self.L, self.Dq), **bf)
        self.KpT = torch.empty((BH, self.Dq, self.Lp), **bf)
        self.VT = torch.empty((BH, 64, vtc.value), **bf)
        self.QsT = torch.empty((BH, 64, self.Lp), **bf)
        self.dOT = torch.empty((BH, 64, self.Lp), **bf)
        self.dQp = torch.empty((BH, self.L, self.Dq), **bf)
        self.delta = torch.empty((BH, self.L), dtype=torch.float32, device=device)
        self.version = 0          # bumped by every prepare: tells a later backward whether the operands are still its own

    def _args(self, qkv, rel_h, rel_w, O, lse, dO=None, dqkv=None, drel_h=None, drel_w=None):
        a = L.AttnArgs()
        a.qkv, a.rel_h, a.rel_w = _p(qkv), _p(rel_h), _p(rel_w)
        a.Qp, a.Kp, a.KpT, a.VT, a.QsT = _p(self.Qp), _p(self.Kp), _p(self.KpT), _p(self.VT), _p(self.QsT)
        a.O, a.lse, a.dO, a.dOT, a.dQp, a.delta = _p(O), _p(lse), _p(dO), _p(self.dOT), _p(self.dQp), _p(self.delta)
        a.dqkv, a.drel_h, a.drel_w = _p(dqkv), _p(drel_h), _p(drel_w)
        a.nB, a.gh, a.gw, a.heads, a.Dq, a.scale = self.nB, self.gh, self.gw, self.heads, self.Dq, 64 ** -0.5
        return a

    def prepare(self, qkv, rel_h, rel_w):
        a = self._args(qkv, rel_h, rel_w, None, None)
        L.call("aldi_attn_prepare", C.byref(a), stream_ptr())

    def forward(self, qkv: torch.Tensor, rel_h: Optional[torch.Tensor], rel_w: Optional[torch.Tensor]):
        """qkv [nB*L, 3*heads*64] bf16 -> (O [nB*L, heads*64] bf16, lse [nB*heads, L] fp32)"""
        assert qkv.dtype == torch.bfloat16 and qkv.shape == (self.nB * self.L, 3 * self.heads * 64) and qkv.is_contiguous()
        assert (rel_h is not None) == self.rel
        if self.rel:
            assert rel_h.shape == (2 * self.gh - 1, 64) and rel_w.shape == (2 * self.gw - 1, 64) and rel_h.dtype == torch.float32
        O = torch.empty((self.nB * self.L, self.heads * 64), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((self.nB * self.heads, self.L), dtype=torch.float32, device=qkv.device)
        a = self._args(qkv, rel_h, rel_w, O, lse)
        L.call("aldi_attn_prepare", C.byref(a), stream_ptr())
        self.version += 1
        L.call("aldi_attn_forward", C.byref(a), stream_ptr())
        return O, lse

    def backward(self, qkv, rel_h, rel_w, O, lse, dO, drel_h=None, drel_w=None, prepared: bool = False) -> torch.Tensor:
        """-> dqkv like qkv; drel_h / drel_w (fp32) accumulate.  Re-runs prepare unless the workspaces still hold this qkv."""
        dqkv = torch.empty_like(qkv)
        a = self._args(qkv, rel_h, rel_w, O, lse, dO, dqkv, drel_h, drel_w)
        if not prepared:
            L.call("aldi_attn_prepare", C.byref(a), stream_ptr())
            self.version += 1
        L.call("aldi_attn_backward", C.byref(a), stream_ptr())
        return dqkv


# ------------------------------------------------------------------------------------------------- ConvNeXt
def dwconv7(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], flip: bool = False) -> torch.Tensor:
    """x [N,H,W,C], wt [7,7,C] (same dtype), bias fp32 -> depthwise 7x7 / pad 3; flip = data gradient (mirrored taps, no bias)"""
    N, H, W_, Cc = x.shape
    y = torch.empty_like(x)
    L.call("aldi_dwconv7", _p(x), _p(wt), _p(bias), _p(y), N, H, W_, Cc, int(flip), dtype_code(x.dtype), stream_ptr())
    return y


def dwconv7_wgrad(x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor) -> None:
    N, H, W_, Cc = x.shape
    L.call("aldi_dwconv7_wgrad", _p(x), _p(g), _p(dw), N, H, W_, Cc, dtype_code(x.dtype), stream_ptr())


def scale_add(x: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, scale: Optional[torch.Tensor], rows_per_sample: int) -> torch.Tensor:
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    L.call("aldi_scale_add", _p(x), _p(y), _p(gamma), _p(scale), _p(out), x.numel() // Cc, Cc, rows_per_sample, dtype_code(x.dtype), stream_ptr())
    return out


def scale_add_backward(g: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, scale: Optional[torch.Tensor], dgamma: torch.Tensor,
                       rows_per_sample: int) -> torch.Tensor:
    Cc = g.shape[-1]
    dy = torch.empty_like(g)
    L.call("aldi_scale_add_backward", _p(g), _p(y), _p(gamma), _p(scale), _p(dy), _p(dgamma), g.numel() // Cc, Cc, rows_per_sample,
           dtype_code(g.dtype), stream_ptr())
    return dy
