"""ConvNeXt trunk (HIP) vs the REFERENCE's own ConvNeXt class (aldi/backbone.py:189-352): golden g10 was produced by importing that
class in the build container (tests/golden/make_golden.py g10) -- state_dict, image, the four normalised stage outputs and every
parameter gradient.  bf16 activations: 4e-2 of range on the maps, 6e-2 (L2 3e-2) on gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def l2err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv7_fwd_bwd_vs_torch(dtype):
    from aldi_amd import vit_ops as V
    torch.manual_seed(0)
    N, H, W, C = 2, 13, 17, 64
    x = torch.randn(N, H, W, C, device=DEV).to(dtype)
    w = (torch.randn(C, 1, 7, 7, device=DEV) * 0.2).to(dtype)
    b = torch.randn(C, device=DEV) * 0.1
    g = torch.randn(N, H, W, C, device=DEV).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=3, groups=C)
    yr.backward(g.float().permute(0, 3, 1, 2))
    wt = w[:, 0].permute(1, 2, 0).contiguous()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(V.dwconv7(x, wt, b), yr.permute(0, 2, 3, 1)) < tol
    assert relerr(V.dwconv7(g, wt, None, flip=True), xr.grad.permute(0, 2, 3, 1)) < tol
    dw = torch.zeros(7, 7, C, device=DEV)
    V.dwconv7_wgrad(x, g, dw)
    assert relerr(dw, wr.grad[:, 0].permute(1, 2, 0)) < max(tol, 1e-4)


def test_scale_add_fwd_bwd_vs_torch():
    from aldi_amd import vit_ops as V
    torch.manual_seed(1)
    rows, C, rps = 60, 32, 30
    x, y, g = (torch.randn(rows, C, device=DEV) for _ in range(3))
    gamma = torch.randn(C, device=DEV)
    s = torch.tensor([0.0, 1.25], device=DEV)
    yr, gr = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    out_r = x + s.repeat_interleave(rps)[:, None] * gr * yr
    out_r.backward(g)
    assert torch.allclose(V.scale_add(x, y, gamma, s, rps), out_r, atol=1e-6)
    dg = torch.zeros(C, device=DEV)
    dy = V.scale_add_backward(g, y, gamma, s, dg, rps)
    assert torch.allclose(dy, yr.grad, atol=1e-6) and torch.allclose(dg, gr.grad, atol=1e-4)


def test_convnext_trunk_vs_reference_class_golden(golden_dir):
    from aldi_amd.convnext import ConvNeXt, ConvNeXtConfig
    from aldi_amd.vit import VitParams
    G = np.load(os.path.join(golden_dir, "g10_convnext.npz"))
    cfg = ConvNeXtConfig(depths=(1, 1, 2, 1), dims=(32, 64, 96, 128), drop_path_rate=0.0)
    params = VitParams(cfg, DEV)
    keys = [str(k) for k in G["keys"]]
    assert [cfg.prefix + k for k in keys] == list(params.spec.keys())          # same state_dict keys, same order as the reference module
    sd = {cfg.prefix + k: torch.from_numpy(G["sd." + k]) for k in keys}
    params.load_state_dict(sd)
    rt = params.state_dict()
    assert all(torch.equal(rt[k], sd[k]) for k in sd)
    net = ConvNeXt(params)
    img = torch.from_numpy(G["img"]).to(DEV)
    params.zero_grad()
    ctx = net.forward(img, [(64, 96), (64, 96)], save=True)
    for i in range(4):
        ref = torch.from_numpy(G[f"out{i}"]).to(DEV).permute(0, 2, 3, 1)
        assert relerr(ctx.outs[i], ref) < 4e-2, i
    net.backward(ctx, [torch.from_numpy(G[f"gout{i}"]).to(DEV).permute(0, 2, 3, 1).contiguous().bfloat16() for i in range(4)])
    grads = params.state_dict_like(params.grad)
    bad = {}
    for k in keys:
        ref = torch.from_numpy(G["grad." + k])
        mine = grads[cfg.prefix + k]
        e_inf, e_2 = relerr(mine, ref), l2err(mine, ref)
        if e_inf > 6e-2 or e_2 > 3e-2:
            bad[k] = (round(e_inf, 4), round(e_2, 4))
    assert not bad, bad
