"""CPU restatement (plain torch, any float dtype) of multi-scale deformable attention sampling.

TEST INFRASTRUCTURE ONLY.  The reference reaches this op through the absent submodule `aldi/detr/libs`
(.gitmodules:4-6; Deformable-DETR `models/ops`), so the algorithm is restated from its published description: for every
(image, query, head), a weighted sum over levels and points of bilinear samples of that head's value map, with pixel
coordinate x = loc_x * W - 0.5 (grid_sample align_corners=False) and zeros outside the map.  Written as explicit corner
gathers (not grid_sample) so that it is an independent statement; pinned in tests/test_msda_oracle_cpu.py against
transformers' `MultiScaleDeformableAttention` (pure-PyTorch grid_sample form, present in the image)."""
from __future__ import annotations

import torch


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N, S, M, D); spatial_shapes [(H, W)] * L; sampling_locations (N, Lq, M, L, P, 2); attention_weights (N, Lq, M, L, P)
    -> (N, Lq, M * D)"""
    N, S, M, D = value.shape
    _, Lq, _, Lv, P, _ = sampling_locations.shape
    out = value.new_zeros(N, Lq, M, D)
    start = 0
    n_idx = torch.arange(N).view(N, 1, 1, 1)
    m_idx = torch.arange(M).view(1, 1, M, 1)
    for l, (H, W) in enumerate(spatial_shapes):
        v = value[:, start:start + H * W].view(N, H, W, M, D)
        start += H * W
        x = sampling_locations[:, :, :, l, :, 0] * W - 0.5          # (N, Lq, M, P)
        y = sampling_locations[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        acc = value.new_zeros(N, Lq, M, P, D)
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                g = v[n_idx, yi.clamp(0, H - 1), xi.clamp(0, W - 1), m_idx]          # (N, Lq, M, P, D)
                acc = acc + g * (wy * wx * ok.to(value.dtype)).unsqueeze(-1)
        out = out + (acc * attention_weights[:, :, :, l].unsqueeze(-1)).sum(3)
    return out.reshape(N, Lq, M * D)
