"""The fused res2 bottleneck kernel (csrc/bneck.hip) against a plain PyTorch fp32 reference of the same three convolutions, and the
trunk that uses it against the layer-by-layer trunk."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _reference(x, res, w1, w2, w3, s1, s2, s3, b1, b2, b3):
    """fp32 math on the operands the kernel sees: bf16 inputs, bf16 scale-folded weights, bf16-rounded intermediate maps"""
    xc = x.float().permute(0, 3, 1, 2)
    f1, f2, f3 = _bf(w1 * s1.view(-1, 1, 1, 1)), _bf(w2 * s2.view(-1, 1, 1, 1)), _bf(w3 * s3.view(-1, 1, 1, 1))
    a1 = _bf(F.relu(F.conv2d(xc, f1.permute(0, 3, 1, 2)) + b1.view(1, -1, 1, 1)))
    a2 = _bf(F.relu(F.conv2d(a1, f2.permute(0, 3, 1, 2), padding=1) + b2.view(1, -1, 1, 1)))
    y = _bf(F.conv2d(a2, f3.permute(0, 3, 1, 2)) + b3.view(1, -1, 1, 1))
    return F.relu(y + res.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,W,Cin", [(2, 37, 50, 256), (1, 8, 16, 64), (3, 21, 19, 64), (1, 64, 96, 256)])
def test_bottleneck_fused_vs_torch(N, H, W, Cin):
    from aldi_amd import _lib as L, ops
    g = torch.Generator().manual_seed(H * 100 + W + Cin)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
    res = x if Cin == 256 else torch.randn(N, H, W, 256, generator=g).to(torch.bfloat16)
    w1 = torch.randn(64, 1, 1, Cin, generator=g) / Cin ** 0.5
    w2 = torch.randn(64, 3, 3, 64, generator=g) / 24.0
    w3 = torch.randn(256, 1, 1, 64, generator=g) / 8.0
    s = [torch.rand(c, generator=g) + 0.5 for c in (64, 64, 256)]
    b = [torch.randn(c, generator=g) * 0.3 for c in (64, 64, 256)]
    want = _reference(x, res, w1, w2, w3, *s, *b)
    ws = [w.to(DEV).contiguous() for w in (w1, w2, w3)]
    plan = ops.FoldWeightsPlan([(w, sc.to(DEV)) for w, sc in zip(ws, s)])
    plan.run()
    for w, sc, o in zip((w1, w2, w3), s, plan.out):
        assert torch.equal(o.cpu().float(), _bf(w * sc.view(-1, 1, 1, 1))), "folded weights: bf16(w * scale), round to nearest even"
    xd = x.to(DEV)
    got = ops.bottleneck_fused(xd, xd if Cin == 256 else res.to(DEV), *plan.out, *[t.to(DEV) for t in b])
    torch.cuda.synchronize()
    assert "bottleneck_fused" in L.last_dispatch()
    got = got.cpu().float()
    err = (got - want).abs()
    # identical operands; what differs is the summation order inside a dot product (and thereby, rarely, one bf16 rounding of an
    # intermediate map, which moves an output by up to ~1 % of the map's scale)
    assert float(err.max()) <= 0.05 * float(want.abs().max()), float(err.max())
    assert float(err.mean()) <= 2e-3 * float(want.abs().mean() + 1e-6), float(err.mean())
    assert float(((err > 0.02 * want.abs().max()).float().mean())) < 1e-3


def test_trunk_with_fused_res2_matches_layerwise():
    """the R50 trunk with res2 as three fused kernels == the same trunk launched layer by layer (FPN outputs, bf16)"""
    from aldi_amd import synthetic as syn
    from aldi_amd.arch import ParamLayout
    from aldi_amd.engine import RCNN, Weights
    K, H, W = 8, 192, 256
    sd = syn.init_state_dict(K, seed=1)
    lay = ParamLayout(K)
    w = Weights(lay, torch.device(DEV), torch.bfloat16, trainable=True)
    w.load_state_dict(sd)
    m = RCNN(w, K)
    _, data, _, _ = syn.make_batch(2, 0, H, W, K, seed=0)
    st, sizes, _ = m.stage_images([d["image"] for d in data])
    outs = {}
    for fused in (True, False):
        m.fused_res2 = fused
        c = m.trunk(st, sizes, save=False)
        torch.cuda.synchronize()
        outs[fused] = [p.float() for p in c.P]
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).abs().max()) <= 0.03 * float(b.abs().max()), (float((a - b).abs().max()), float(b.abs().max()))
        assert float((a - b).abs().mean()) <= 2e-2 * float(b.abs().mean())      # (scale folded into bf16 weights vs applied in fp32: bf16-level)
