"""HIP dgrad (igemm with rotated weights) and wgrad vs torch-CPU autograd of the oracle's conv."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 25, 42, 64, 64, 3, 1, 1),
    (2, 24, 40, 256, 64, 1, 1, 0),
    (2, 25, 41, 256, 512, 1, 2, 0),
    (2, 10, 12, 64, 64, 3, 1, 0),
    (1, 17, 19, 128, 16, 1, 1, 0),
    (5, 1, 1, 1024, 48, 1, 1, 0),
    (1, 50, 84, 256, 256, 3, 1, 1),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", CASES)
def test_dgrad_wgrad_vs_autograd(case, dtype, tol):
    from aldi_amd import ops
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    gy = torch.randn(N, Cout, Ho, Wo, generator=g)
    xin = torch.relu(torch.randn(N, Cin, H, W, generator=g))       # "saved activation" for the ReLU-backward mask
    if dtype == torch.bfloat16:
        x, gy = x.to(dtype).float(), gy.to(dtype).float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1)
    (y * gy).sum().backward()
    ref_dx = x.grad * (xin > 0)
    ref_dw = w.grad
    dev = "cuda"
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    xd, gd, md = nhwc(x).to(dev, dtype), nhwc(gy).to(dev, dtype), nhwc(xin).to(dev, dtype)
    wm = nhwc(w).to(dev)                                   # fp32 master, [Cout,KH,KW,Cin]
    sd = scale.to(dev)
    # wgrad
    dw = torch.zeros_like(wm)
    ops.conv_wgrad(xd, gd, dw, KH=k, KW=k, stride=stride, pad=pad, scale=sd)
    # dgrad = conv of g with rotated/transposed weights
    wt = ops.dgrad_weights(wm, sd, dtype)
    if stride == 1:
        dx = ops.conv2d(gd, wt, stride=1, pad=k - 1 - pad, mask=md)
    else:
        dx = torch.zeros(N, H, W, Cin, device=dev, dtype=dtype)
        ops.conv2d(gd, wt, stride=1, pad=0, mask=md, out=dx, out_scale=stride, out_hw=(H, W))
    torch.cuda.synchronize()
    got_dw = dw.cpu().permute(0, 3, 1, 2)
    e = (got_dw - ref_dw).abs().max().item()
    wtol = tol * (1.0 if dtype == torch.float32 else 0.25)     # wgrad accumulates in fp32 from exact bf16 products
    assert e <= wtol * max(1.0, ref_dw.abs().max().item()), ("dw", e, ref_dw.abs().max().item())
    got_dx = dx.float().cpu().permute(0, 3, 1, 2)
    if dtype == torch.bfloat16:      # the HIP dgrad sees bf16-rounded (scale*w)
        pass
    e = (got_dx - ref_dx).abs().max().item()
    assert e <= tol * max(1.0, ref_dx.abs().max().item()), ("dx", e, ref_dx.abs().max().item())


def test_bias_grad():
    from aldi_amd import ops
    g = torch.randn(3000, 48)
    for dt in (torch.float32, torch.bfloat16):
        gd = g.to("cuda", dt)
        db = torch.zeros(48, device="cuda")
        ops.bias_grad(gd, db)
        ref = gd.float().cpu().sum(0)
        assert (db.cpu() - ref).abs().max() < 1e-2 if dt == torch.bfloat16 else 1e-3


@pytest.mark.parametrize("rows,C", [(1, 8), (17, 2048), (4201, 1536), (16800, 3072), (70001, 192), (333, 6144)])
def test_bias_grad_shapes(rows, C):
    """ragged row counts, rows shorter / longer than one workgroup pass, column slices (C > 2048); accumulates into db"""
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(rows + C)
    g = torch.randn(rows, C, generator=gen).bfloat16()
    db0 = torch.randn(C, generator=gen)
    db = db0.cuda()
    ops.bias_grad(g.cuda(), db)
    ref = db0.double() + g.double().sum(0)
    assert (db.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, rows ** 0.5) * 4
