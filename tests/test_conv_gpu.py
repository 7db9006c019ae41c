"""HIP implicit-GEMM conv vs the oracle's conv (torch CPU fp32 F.conv2d)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nchw, w_oihw, stride, pad, scale, shift, res, relu):
    y = F.conv2d(x_nchw, w_oihw, None, stride, pad)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return y


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 25, 42, 64, 64, 3, 1, 1),
    (2, 24, 40, 256, 64, 1, 1, 0),
    (2, 25, 41, 256, 512, 1, 2, 0),
    (1, 13, 21, 256, 256, 3, 1, 1),
    (2, 10, 12, 64, 64, 3, 1, 0),
    (1, 17, 19, 128, 16, 1, 1, 0),
    (3, 1, 1, 12544, 1024, 1, 1, 0),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("case", CASES)
def test_conv_igemm_vs_oracle(case, dtype, tol):
    from aldi_amd import ops
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = torch.randn(Cout, generator=g) * 0.1
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g)
    if dtype == torch.bfloat16:       # the oracle sees the same bf16-rounded operands
        x, w, res = (t.to(dtype).float() for t in (x, w, res))
    ref = _ref(x, w, stride, pad, scale, shift, res, True)
    dev = "cuda"
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    y = ops.conv2d(xd, wd, stride=stride, pad=pad, scale=scale.to(dev), shift=shift.to(dev), res=rd, res_mode=1, relu=True)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err
    # fp32 side output and no-epilogue path
    y32 = ops.conv2d(xd, wd, stride=stride, pad=pad, want_f32=True)
    torch.cuda.synchronize()
    ref2 = F.conv2d(x, w, None, stride, pad)
    err2 = (y32.cpu().permute(0, 3, 1, 2) - ref2).abs().max().item()
    assert err2 <= (2e-5 if dtype == torch.float32 else 2e-3) * max(1.0, ref2.abs().max().item()), err2
