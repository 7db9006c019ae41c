"""Pins oracle/ms_deform_attn.py against transformers' pure-PyTorch MultiScaleDeformableAttention (CPU)."""
import torch


def _inputs(seed, N=2, M=4, D=8, Lq=11, shapes=((6, 9), (3, 5), (2, 2)), P=3, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D, generator=g, dtype=dtype)
    loc = torch.rand(N, Lq, M, len(shapes), P, 2, generator=g, dtype=dtype) * 1.3 - 0.15      # some samples fall outside the maps
    w = torch.softmax(torch.randn(N, Lq, M, len(shapes) * P, generator=g, dtype=dtype), -1).view(N, Lq, M, len(shapes), P)
    return value, list(shapes), loc, w


def test_oracle_matches_transformers_msda():
    from transformers.models.deformable_detr.modeling_deformable_detr import MultiScaleDeformableAttention
    from oracle.ms_deform_attn import ms_deform_attn
    for seed in range(3):
        value, shapes, loc, w = _inputs(seed)
        v1, l1, w1 = (t.clone().requires_grad_(True) for t in (value, loc, w))
        ref = MultiScaleDeformableAttention()(v1, torch.tensor(shapes), shapes, None, l1, w1, 64)
        v2, l2, w2 = (t.clone().requires_grad_(True) for t in (value, loc, w))
        out = ms_deform_attn(v2, shapes, l2, w2)
        assert (out - ref).abs().max().item() < 1e-10
        g = torch.randn_like(ref)
        ref.backward(g)
        out.backward(g)
        assert (v1.grad - v2.grad).abs().max().item() < 1e-10
        assert (l1.grad - l2.grad).abs().max().item() < 1e-9
        assert (w1.grad - w2.grad).abs().max().item() < 1e-10
