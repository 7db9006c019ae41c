"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return L.F32
    if t == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def conv2d(x: torch.Tensor, w: torch.Tensor, *, stride: int = 1, pad: int = 0,
           scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None, res_mode: int = 0, relu: bool = False,
           mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           out_f32: Optional[torch.Tensor] = None, want_f32: bool = False,
           out_scale: int = 1, out_hw=None) -> torch.Tensor:
    """x [N,H,W,Cin] (NHWC), w [Cout,KH,KW,Cin] -> y [N,Ho,Wo,Cout] (or the scattered
    [N,OH,OW,Cout] tensor when out_scale > 1, which must be pre-zeroed by the caller)."""
    N, H, W_, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    if out_scale > 1:
        OH, OW = out_hw
        shape = (N, OH, OW, Cout)
    else:
        OH = OW = 0
        shape = (N, Ho, Wo, Cout)
    if out is None and not want_f32:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty(shape, dtype=torch.float32, device=x.device)
    a = L.ConvArgs(_p(x), _p(w), _p(out), _p(out_f32), _p(scale), _p(shift), _p(res), _p(mask),
                   N, H, W_, Cin, Cout, KH, KW, stride, pad, Ho, Wo,
                   int(relu), res_mode, out_scale, OH, OW, dtype_code(x.dtype))
    L.check(L.conv_igemm(C.byref(a), stream_ptr()), "aldi_conv_igemm")
    return out_f32 if want_f32 and out is None else out


def conv_wgrad(x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, *, KH: int, KW: int, stride: int = 1, pad: int = 0,
               scale: Optional[torch.Tensor] = None) -> None:
    """dw [Cout,KH,KW,Cin] fp32 += scale * (g^T . im2col(x)); x [N,H,W,Cin], g [N,Ho,Wo,Cout]."""
    N, H, W_, Cin = x.shape
    _, Ho, Wo, Cout = g.shape
    assert dw.dtype == torch.float32 and dw.numel() == Cout * KH * KW * Cin and x.dtype == g.dtype
    a = L.WgradArgs(_p(x), _p(g), _p(dw), _p(scale), N, H, W_, Cin, Cout, KH, KW, stride, pad, Ho, Wo, dtype_code(x.dtype))
    L.check(L.conv_wgrad(C.byref(a), stream_ptr()), "aldi_conv_wgrad")


def bias_grad(g: torch.Tensor, db: torch.Tensor) -> None:
    Cc = g.shape[-1]
    L.check(L.bias_grad(_p(g), _p(db), g.numel() // Cc, Cc, dtype_code(g.dtype), stream_ptr()), "aldi_bias_grad")


def dgrad_weights(w_master: torch.Tensor, scale: Optional[torch.Tensor], dtype: torch.dtype,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w_master fp32 [Cout,KH,KW,Cin] -> [Cin,KH,KW,Cout] rotated/transposed (x scale[co]) in `dtype`."""
    Cout, KH, KW, Cin = w_master.shape
    if out is None:
        out = torch.empty((Cin, KH, KW, Cout), dtype=dtype, device=w_master.device)
    L.check(L.dgrad_weights(_p(w_master), _p(scale), _p(out), Cout, KH, KW, Cin, dtype_code(dtype), stream_ptr()), "aldi_dgrad_weights")
    return out
