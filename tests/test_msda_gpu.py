"""Multi-scale deformable attention sampling on the HIP library vs the CPU oracle (fp32 kernel: 1e-5 forward, 1e-4 on
gradients relative to their range -- the value gradient is an fp32 atomic scatter)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("N,M,D,Lq,shapes,P", [
    (2, 8, 32, 300, ((25, 42), (13, 21), (7, 11), (4, 6)), 4),        # Deformable-DETR decoder: 300 queries, 4 levels, 8 heads x 4 points
    (1, 8, 32, 1511, ((25, 42), (13, 21), (7, 11), (4, 6)), 4),       # encoder: every pixel of every level is a query
    (3, 2, 64, 37, ((5, 7), (3, 3)), 2),                               # head dim 64, odd sizes
])
def test_msda_fwd_bwd_vs_oracle(N, M, D, Lq, shapes, P):
    from aldi_amd.detr import MSDeformAttnFunction
    from oracle.ms_deform_attn import ms_deform_attn as oracle
    g = torch.Generator().manual_seed(N * 100 + Lq)
    S = sum(h * w for h, w in shapes)
    Lv = len(shapes)
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, Lv, P, 2, generator=g) * 1.2 - 0.1                # a band outside [0, 1]: zero padding and its gradients
    w = torch.softmax(torch.randn(N, Lq, M, Lv * P, generator=g), -1).view(N, Lq, M, Lv, P)
    gout = torch.randn(N, Lq, M * D, generator=g)
    v1, l1, w1 = (t.double().requires_grad_(True) for t in (value, loc, w))
    ref = oracle(v1, list(shapes), l1, w1)
    ref.backward(gout.double())
    v2, l2, w2 = (t.to(DEV).requires_grad_(True) for t in (value, loc, w))
    shp = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    lstart = torch.cat([shp.new_zeros(1), (shp[:, 0] * shp[:, 1]).cumsum(0)[:-1]])
    out = MSDeformAttnFunction.apply(v2, shp, lstart, l2, w2, 64)
    assert relerr(out.cpu().double(), ref) < 1e-5
    out.backward(gout.to(DEV))
    assert relerr(v2.grad.cpu().double(), v1.grad) < 1e-4
    assert relerr(w2.grad.cpu().double(), w1.grad) < 1e-4
    # d/d(loc) is discontinuous where a sample sits on a pixel boundary (the bilinear cell changes): a sample within fp32
    # rounding of an integer coordinate may legitimately fall into the neighbouring cell, so those are left out; elsewhere
    # the corner DIFFERENCES times W (resp. H) cost a few 1e-4 of the largest entry in fp32
    wh = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64).view(1, 1, 1, Lv, 1, 2)
    px = loc.double() * wh - 0.5
    smooth = ((px - px.round()).abs() > 1e-3).all(-1, keepdim=True).expand_as(px)
    diff = ((l2.grad.cpu().double() - l1.grad) * smooth).abs().max() / l1.grad.abs().max()
    assert diff.item() < 5e-4, diff.item()
    assert smooth.double().mean().item() > 0.99


def test_msda_rejects_cpu_tensors():
    from aldi_amd.detr import MSDeformAttnFunction
    v = torch.zeros(1, 4, 1, 32)
    with pytest.raises(RuntimeError):
        MSDeformAttnFunction.apply(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 64)
