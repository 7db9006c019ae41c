"""`MSDeformAttnFunction` with the call signature of Deformable-DETR's CUDA extension (what the reference's absent
`aldi/detr/libs` submodule provides: .gitmodules:4-6, configs/Base-DETR.yaml:1-81), on the HIP library.

`MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)` is a
torch.autograd.Function, so a Deformable-DETR encoder / decoder written against the extension runs unchanged; both directions
are one C-ABI call each (aldi_ms_deform_attn_forward / _backward).  fp32 (the reference runs DETR with AMP off)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..ops import _p, stream_ptr


def _check(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    if not value.is_cuda:
        raise RuntimeError("ms_deform_attn runs on the MI355X HIP path only; there is no CPU fallback")
    N, S, M, D = value.shape
    _, Lq, M2, Lv, P, two = sampling_locations.shape
    assert M2 == M and two == 2 and attention_weights.shape == (N, Lq, M, Lv, P) and spatial_shapes.shape == (Lv, 2)
    assert value.dtype == torch.float32 and sampling_locations.dtype == torch.float32 and attention_weights.dtype == torch.float32
    return N, S, M, D, Lq, Lv, P


class MSDeformAttnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step=64):
        N, S, M, D, Lq, Lv, P = _check(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        value, loc, w = value.contiguous(), sampling_locations.contiguous(), attention_weights.contiguous()
        shapes = value_spatial_shapes.to(torch.int32).contiguous()
        lstart = value_level_start_index.to(torch.int32).contiguous()
        out = torch.empty((N, Lq, M * D), dtype=torch.float32, device=value.device)
        L.call("aldi_ms_deform_attn_forward", _p(value), _p(shapes), _p(lstart), _p(loc), _p(w), _p(out), N, S, M, D, Lq, Lv, P, stream_ptr())
        ctx.save_for_backward(value, shapes, lstart, loc, w)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        value, shapes, lstart, loc, w = ctx.saved_tensors
        N, S, M, D = value.shape
        _, Lq, _, Lv, P, _ = loc.shape
        g = grad_output.contiguous().to(torch.float32)
        gv, gl, gw = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(w)
        L.call("aldi_ms_deform_attn_backward", _p(value), _p(shapes), _p(lstart), _p(loc), _p(w), _p(g), _p(gv), _p(gl), _p(gw),
               N, S, M, D, Lq, Lv, P, stream_ptr())
        return gv, None, None, gl, gw, None


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    return MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, 64)
