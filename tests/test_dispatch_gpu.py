"""Every dispatch arm of the dense kernels, compared numerically at the size that SELECTS it
(VERDICT r01 weak #1): the benchmark-scale tile variants (256x128 halo igemm, long-K 256x128,
256x256 wgrad, ...) run at BASELINE.json's shapes through the DEFAULT dispatch, and every template is
additionally forced onto small ragged shapes through aldi_set_tuning.  aldi_last_dispatch() names the
kernel that ran, so a test cannot silently exercise a different arm.

References (what the reference reaches through detectron2: aldi/trainer.py:87 forward, :79 backward):
  * an exact fp64 CPU evaluation of the convolution on a SAMPLE of output pixels that covers the four
    image borders, the image seams of the batch, the first / last rows of every tile size and the last
    partial tile (operands rounded to bf16 first, so the only difference left is accumulation order and
    the final rounding);
  * the whole tensor against fp32 library GEMMs on the GPU (torch.matmul -> hipBLASLt, one per tap on shifted
    views: an independent implementation) -- catches a wrong border mask anywhere in the tensor.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_tuning():
    from aldi_amd import _lib as L
    L.reset_tuning()
    yield
    L.reset_tuning()


def _sample_pixels(N, Ho, Wo, extra=()):
    """flat output-pixel indices: corners, borders, batch seams, tile edges (16/64/128/256 multiples), tail, random"""
    M = N * Ho * Wo
    idx = set()
    for n in (0, N - 1):
        for h in (0, 1, Ho // 2, Ho - 2, Ho - 1):
            for w in (0, 1, Wo // 2, Wo - 2, Wo - 1):
                if 0 <= h < Ho and 0 <= w < Wo:
                    idx.add((n * Ho + h) * Wo + w)
    for t in (16, 64, 128, 256):
        for k in (1, 2, 3, M // t // 2, M // t - 1, M // t):
            for d in (-1, 0, 1):
                q = k * t + d
                if 0 <= q < M:
                    idx.add(q)
    for q in range(max(0, M - 40), M):
        idx.add(q)
    g = torch.Generator().manual_seed(M)
    idx.update(torch.randint(0, M, (64,), generator=g).tolist())
    idx.update(extra)
    return torch.tensor(sorted(idx), dtype=torch.long)


def _conv_ref_at(x, w, pix, stride, pad, Ho, Wo):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] (CPU float) -> fp64 conv output rows at flat output pixels `pix`"""
    N, H, W_, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    xd, wd = x.double(), w.double()
    n = pix // (Ho * Wo)
    r = pix % (Ho * Wo)
    ho, wo = r // Wo, r % Wo
    out = torch.zeros(len(pix), Cout, dtype=torch.float64)
    for kh in range(KH):
        for kw in range(KW):
            hi, wi = ho * stride - pad + kh, wo * stride - pad + kw
            ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W_)
            rows = xd[n[ok], hi[ok], wi[ok]]                      # [m, Cin]
            out[ok] += rows @ wd[:, kh, kw, :].t()
    return out


def _conv_full_gpu(xd, wd, stride, pad, Ho, Wo):
    """whole-tensor second opinion on the GPU: one fp32 library GEMM (torch.matmul -> hipBLASLt) per tap on shifted views"""
    N, H, W_, Cin = xd.shape
    Cout, KH, KW, _ = wd.shape
    xp = F.pad(xd.float(), (0, 0, pad, pad, pad, pad))
    out = torch.zeros(N * Ho * Wo, Cout, dtype=torch.float32, device=xd.device)
    for kh in range(KH):
        for kw in range(KW):
            xs = xp[:, kh: kh + (Ho - 1) * stride + 1: stride, kw: kw + (Wo - 1) * stride + 1: stride].reshape(-1, Cin)
            out += xs @ wd[:, kh, kw, :].float().t()
    return out.view(N, Ho, Wo, Cout)


def _mk(N, H, W_, Cin, Cout, k, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W_, Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    return x, w, g


def _check_forward(case, dtype, expect, *, force=None, full=True):
    from aldi_amd import _lib as L
    from aldi_amd import ops
    N, H, W_, Cin, Cout, k, stride, pad = case
    x, w, g = _mk(N, H, W_, Cin, Cout, k, sum(case), dtype)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Ho, Wo, Cout, generator=g)
    if dtype == torch.bfloat16:
        res = res.bfloat16().float()
    dev = "cuda"
    xd, wd, rd = x.to(dev, dtype), w.to(dev, dtype), res.to(dev, dtype)
    L.set_tuning("igemm_direct", 0)          # these cases are about the tile templates with the STAGED epilogue (the direct one: _check_direct below)
    if force is not None:
        L.set_tuning("igemm_force", force)
    y = ops.conv2d(xd, wd, stride=stride, pad=pad, scale=scale.to(dev), shift=shift.to(dev), res=rd, res_mode=1, relu=True)
    name = L.last_dispatch()
    y32 = ops.conv2d(xd, wd, stride=stride, pad=pad, want_f32=True)          # plain epilogue, fp32 side output
    torch.cuda.synchronize()
    assert name == expect, (name, expect)
    pix = _sample_pixels(N, Ho, Wo)
    ref = _conv_ref_at(x, w, pix, stride, pad, Ho, Wo)
    got32 = y32.view(-1, Cout)[pix.to(dev)].double().cpu()
    tol32 = 3e-5                                                              # bf16 products are exact in fp32; only the sum order differs
    e = (got32 - ref).abs().max().item()
    assert e <= tol32 * max(1.0, ref.abs().max().item()) * max(1.0, (k * k * Cin / 256) ** 0.5), ("raw", e)
    full_ref = ref * scale.double() + shift.double()
    if dtype == torch.bfloat16:
        full_ref = full_ref.float().bfloat16().double()                       # the bf16 epilogue rounds conv*scale+shift before the residual add
    full_ref = torch.relu(full_ref + res.view(-1, Cout)[pix].double())
    got = y.view(-1, Cout)[pix.to(dev)].double().cpu()
    tol = 5e-5 if dtype == torch.float32 else 8e-3                            # one bf16 ulp of the result (2^-8 relative) + a flipped pre-rounding
    e = ((got - full_ref).abs() / full_ref.abs().clamp(min=1.0)).max().item()
    assert e <= tol, ("epilogue", e)
    if full:
        r2 = _conv_full_gpu(xd, wd, stride, pad, Ho, Wo)
        e2 = (y32 - r2).abs().max().item()
        assert e2 <= 2e-3 * max(1.0, r2.abs().max().item()), ("full tensor", e2)
        assert torch.isfinite(y.float()).all()
    return name


# ---------------------------------------------------------------------------------- default dispatch at benchmark scale
FULL_FWD = [
    # (N, H, W, Cin, Cout, k, stride, pad), kernel the DEFAULT dispatch must pick
    ((4, 200, 336, 256, 256, 3, 1, 1), "igemm<bf16,256,256,4,2,halo64>"),         # FPN output p2 / RPN conv p2 (student N=4): 128-byte K slabs
    ((2, 200, 336, 256, 256, 3, 1, 1), "igemm<bf16,256,256,4,2,halo64>"),         # same, teacher N=2
    ((4, 100, 168, 256, 256, 3, 1, 1), "igemm<bf16,256,256,4,2,halo64>"),         # p3 3x3, student
    ((2, 100, 168, 256, 256, 3, 1, 1), "igemm<bf16,128,64,4,1,flat,halo>"),       # p3 3x3, teacher
    ((4, 50, 84, 256, 256, 3, 1, 1), "igemm<bf16,128,64,4,1,flat,halo>"),         # res4 conv2
    ((4, 200, 336, 64, 64, 3, 1, 1), "igemm<bf16,128,64,4,1,flat,halo>"),         # res2 conv2
    ((4, 200, 336, 64, 256, 1, 1, 0), "igemm<bf16,128,64,4,1,pipe,tap>"),         # res2 conv3 (short K: half-width tiles)
    ((4, 200, 336, 256, 64, 1, 1, 0), "igemm<bf16,128,64,4,1,pipe,tap>"),         # res2 conv1
    ((4, 200, 336, 256, 512, 1, 2, 0), "igemm<bf16,128,64,4,1,pipe,tap>"),        # res3 shortcut (stride 2, K = 256)
    ((4, 50, 84, 768, 1024, 1, 2, 0), "igemm<bf16,128,128,2,2,pipe,tap>"),        # a strided 1x1 with K > 512: full tiles
    ((4, 200, 336, 256, 16, 1, 1, 0), "igemm<bf16,128,16,4,1,pipe,tap>"),         # RPN objectness + deltas
    ((1, 120, 140, 3072, 768, 1, 1, 0), "igemm<bf16,256,128,4,2,flat,tap,k64>"),  # 16800 x 3072 -> 768 (ViT MLP fc2): 128-byte K slabs
    ((2048, 1, 1, 12544, 1024, 1, 1, 0), "igemm<bf16,64,64,2,2,flat,tap,k64>"),   # box head FC1 (long K: 128-byte slabs)
    ((4, 25, 42, 512, 2048, 1, 1, 0), "igemm<bf16,128,64,4,1,pipe,tap>"),         # res5 conv3
    ((2, 25, 42, 512, 512, 3, 1, 1), "igemm<bf16,128,64,4,1,flat,halo>"),         # res5 conv2, teacher
]


@pytest.mark.parametrize("case,expect", FULL_FWD)
def test_forward_default_dispatch_fullsize(case, expect):
    _check_forward(case, torch.bfloat16, expect)


def test_forward_default_dispatch_fullsize_fp32():
    """parity mode at benchmark scale (the fp32 arms: direct epilogue, 16x16x4 MFMA)"""
    # fp32 runs on 64 x 64 tiles by default (igemm_f32_tile64_max: the f32-input MFMA is slow enough that workgroups pay, not bytes per flop)
    for case in ((2, 200, 336, 256, 256, 3, 1, 1), (2, 200, 336, 64, 256, 1, 1, 0), (2, 200, 336, 64, 64, 3, 1, 1), (2, 25, 42, 512, 512, 3, 1, 1)):
        _check_forward(case, torch.float32, "igemm<f32,64,64,2,2,pipe,tap>")
    from aldi_amd import _lib as L
    L.set_tuning("igemm_f32_tile64_max", 0)                # ... the bf16 tile rules
    try:
        _check_forward((2, 200, 336, 256, 256, 3, 1, 1), torch.float32, "igemm<f32,256,128,4,2,flat,tap>")
        _check_forward((2, 200, 336, 64, 256, 1, 1, 0), torch.float32, "igemm<f32,128,128,2,2,pipe,tap>")
        _check_forward((2, 200, 336, 64, 64, 3, 1, 1), torch.float32, "igemm<f32,128,64,4,1,pipe,tap>")
    finally:
        L.reset_tuning()
    # the fp32 halo arms (igemm_halo_f32 = least number of half-width tiles; off by default: another summation order than the tap form)
    L.set_tuning("igemm_halo_f32", 400)
    try:
        _check_forward((2, 200, 336, 256, 256, 3, 1, 1), torch.float32, "igemm<f32,256,128,4,2,flat,halo>")
        _check_forward((2, 200, 336, 64, 64, 3, 1, 1), torch.float32, "igemm<f32,128,64,4,1,flat,halo>")
        _check_forward((2, 100, 168, 128, 128, 3, 1, 1), torch.float32, "igemm<f32,128,64,4,1,flat,halo>")      # 526 tiles
        _check_forward((2, 25, 42, 512, 512, 3, 1, 1), torch.float32, "igemm<f32,64,64,2,2,pipe,tap>")          # 136 tiles
    finally:
        L.reset_tuning()


# ---------------------------------------------------------------------------------- every template forced onto small ragged shapes
SMALL = [
    (2, 25, 42, 64, 96, 3, 1, 1),      # halo-eligible, ragged M (2100), Cout not a tile multiple
    (1, 19, 23, 128, 256, 3, 1, 1),    # halo-eligible, tiny image: every tile crosses image rows
    (3, 9, 130, 32, 64, 3, 1, 1),      # halo, wide rows (W > tile rows apart), Cin = one chunk
    (2, 24, 40, 256, 72, 1, 1, 0),     # 1x1
    (2, 25, 41, 256, 136, 1, 2, 0),    # strided 1x1
    (2, 10, 12, 64, 68, 3, 1, 0),      # 3x3 without padding (ConvDiscriminator, aldi/align.py:110); Cout % 8 != 0: direct epilogue
    (1, 17, 19, 64, 16, 3, 2, 1),      # strided 3x3
    (5, 1, 1, 1032, 48, 1, 1, 0),      # linear with a ragged K (1032 = 32*32 + 8)
]
NAMES = {1: "128,128,2,2", 2: "128,64,4,1", 3: "64,64,2,2", 4: "256,128,4,2", 5: "128,16,4,1"}


@pytest.mark.parametrize("force", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", SMALL)
def test_forward_forced_templates_bf16(case, force):
    N, H, W_, Cin, Cout, k, stride, pad = case
    halo = k == 3 and stride == 1 and pad == 1 and Cin % 32 == 0
    if halo and force in (1, 2, 4):
        expect = "igemm<bf16,%s,flat,halo>" % NAMES[force]
    elif halo and force == 3:
        expect = "igemm<bf16,128,64,4,1,flat,halo>"          # no 64x64 halo form: heuristics (small => 128x64)
    elif halo:
        expect = "igemm<bf16,128,16,4,1,pipe,tap>"
    else:
        expect = "igemm<bf16,%s,%s,tap>" % (NAMES[force], "flat" if force == 4 else "pipe")
    _check_forward(case, torch.bfloat16, expect, force=force, full=False)


K64 = [(2, 24, 40, 256, 72, 1, 1, 0), (5, 1, 1, 1032, 48, 1, 1, 0), (3, 7, 9, 64, 200, 1, 1, 0), (130, 1, 1, 2048, 136, 1, 1, 0)]


@pytest.mark.parametrize("force", [6, 7, 8])
@pytest.mark.parametrize("case", K64)
def test_forward_forced_k64_templates_bf16(case, force):
    """the 128-byte K-slab forms (64 bf16 channels per LDS row, two MFMA k-steps per slab) of the plain 1x1 / linear layers:
    ragged M and Cout, a ragged K tail (1032 = 16 * 64 + 8), a single slab (K = 64)"""
    expect = "igemm<bf16,%s,flat,tap,k64>" % NAMES[force - 5]
    _check_forward(case, torch.bfloat16, expect, force=force, full=False)


def test_long_k_linear_takes_k64_by_default():
    from aldi_amd import _lib as L
    L.reset_tuning()
    assert _check_forward((130, 1, 1, 2048, 136, 1, 1, 0), torch.bfloat16, "igemm<bf16,64,64,2,2,flat,tap,k64>", full=False)
    assert _check_forward((130, 1, 1, 512, 136, 1, 1, 0), torch.bfloat16, "igemm<bf16,64,64,2,2,pipe,tap>", full=False)


@pytest.mark.parametrize("force", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", SMALL[:2] + SMALL[3:7])
def test_forward_forced_templates_fp32(case, force):
    expect = "igemm<f32,%s,%s,tap>" % (NAMES[force], "flat" if force == 4 else "pipe")
    _check_forward(case, torch.float32, expect, force=force, full=False)


def test_halo_off_equals_halo_on():
    """the tap-by-tap form of the same 3x3 conv is bit-identical per tile family or within fp32 sum-order noise"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    x, w, _ = _mk(2, 37, 53, 64, 128, 3, 11, torch.bfloat16)
    xd, wd = x.cuda().bfloat16(), w.cuda().bfloat16()
    a = ops.conv2d(xd, wd, pad=1, want_f32=True)
    assert "halo" in L.last_dispatch()
    L.set_tuning("igemm_halo", 0)
    b = ops.conv2d(xd, wd, pad=1, want_f32=True)
    assert "tap" in L.last_dispatch()
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


# ---------------------------------------------------------------------------------- data gradient (mask epilogue, scatter output)
@pytest.mark.parametrize("case,expect", [
    ((4, 200, 336, 256, 256, 3, 1, 1), "igemm<bf16,256,256,4,2,halo64>"),         # a p2-size 3x3 dgrad with mask + residual: the staged epilogue
    ((4, 50, 84, 256, 256, 3, 1, 1), "igemm<bf16,128,64,4,1,flat,halo>"),         # res4 conv2 dgrad with ReLU mask
    ((4, 50, 84, 1024, 256, 1, 1, 0), "igemm<bf16,64,64,2,2,flat,tap,k64>"),      # res4 conv3 dgrad (g: 1024 -> 256)
])
def test_dgrad_fullsize_default_dispatch(case, expect):
    """dgrad = conv of g with the rotated / transposed weights + ReLU-backward mask (+ residual), at benchmark scale"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    N, H, W_, Cg, Cx, k, stride, pad = case          # g has Cg channels, the input gradient Cx
    gen = torch.Generator().manual_seed(sum(case) + 1)
    g = torch.randn(N, H, W_, Cg, generator=gen).bfloat16().float()
    wm = torch.randn(Cg, k, k, Cx, generator=gen) / (Cg * k * k) ** 0.5      # forward weight [Cout=Cg][k][k][Cin=Cx]
    sc = 0.5 + torch.rand(Cg, generator=gen)
    act = torch.relu(torch.randn(N, H, W_, Cx, generator=gen)).bfloat16().float()
    resid = torch.randn(N, H, W_, Cx, generator=gen).bfloat16().float()
    dev = "cuda"
    wt = ops.dgrad_weights(wm.to(dev), sc.to(dev), torch.bfloat16)           # [Cx][k][k][Cg], scale folded, rounded to bf16
    gd, md, rd = g.to(dev).bfloat16(), act.to(dev).bfloat16(), resid.to(dev).bfloat16()
    dx = ops.conv2d(gd, wt, pad=k - 1 - pad, mask=md, res=rd, res_mode=1)
    name = L.last_dispatch()
    torch.cuda.synchronize()
    assert name == expect, name
    pix = _sample_pixels(N, H, W_)
    ref = _conv_ref_at(g, wt.float().cpu(), pix, 1, k - 1 - pad, H, W_)
    ref = ref.float().bfloat16().double() + resid.view(-1, Cx)[pix].double()
    ref = ref * (act.view(-1, Cx)[pix] > 0)
    got = dx.view(-1, Cx)[pix.to(dev)].double().cpu()
    e = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    assert e <= 8e-3, e
    # independent of the sample: masked-out positions are exactly zero everywhere, the rest finite
    assert bool((dx[md == 0] == 0).all()) and torch.isfinite(dx.float()).all()


def test_dgrad_strided_scatter_fullsize():
    """dgrad of a stride-2 1x1 conv (res3.0 conv1 / shortcut): scattered output + accumulate through res_mode=1"""
    from aldi_amd import ops
    N, H, W_, Cg, Cx = 4, 100, 168, 512, 256
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(N, H, W_, Cg, generator=gen).bfloat16().float()
    wm = torch.randn(Cg, 1, 1, Cx, generator=gen) / Cg ** 0.5
    dev = "cuda"
    wt = ops.dgrad_weights(wm.to(dev), None, torch.bfloat16)
    gx = torch.zeros(N, 2 * H, 2 * W_, Cx, device=dev, dtype=torch.bfloat16)
    ops.conv2d(g.to(dev).bfloat16(), wt, out=gx, out_scale=2, out_hw=(2 * H, 2 * W_))
    ops.conv2d(g.to(dev).bfloat16(), wt, out=gx, out_scale=2, out_hw=(2 * H, 2 * W_), res=gx, res_mode=1)   # accumulate a second time
    torch.cuda.synchronize()
    pix = _sample_pixels(N, H, W_)
    ref = _conv_ref_at(g, wt.float().cpu(), pix, 1, 0, H, W_).float().bfloat16().double()
    ref = (ref + ref).float().bfloat16().double()
    n, r = pix // (H * W_), pix % (H * W_)
    got = gx[n.to(dev), (r // W_ * 2).to(dev), (r % W_ * 2).to(dev)].double().cpu()
    e = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    assert e <= 8e-3, e
    odd = gx[:, 1::2].abs().max().item() + gx[:, :, 1::2].abs().max().item()
    assert odd == 0.0                                                           # untouched positions stay zero


# ---------------------------------------------------------------------------------- weight gradient: the four kernels
def _wgrad_ref(x, g, k, stride, pad):
    """fp64 dW [Cout,k,k,Cin] on the GPU (exact products of bf16-rounded operands, fp64 accumulation)"""
    N, H, W_, Cin = x.shape
    _, Ho, Wo, Cout = g.shape
    xd = x.double()
    gd = g.double().reshape(-1, Cout)
    out = torch.zeros(Cout, k, k, Cin, dtype=torch.float64, device=x.device)
    xp = F.pad(xd, (0, 0, pad, pad, pad, pad))
    for kh in range(k):
        for kw in range(k):
            xs = xp[:, kh: kh + (Ho - 1) * stride + 1: stride, kw: kw + (Wo - 1) * stride + 1: stride].reshape(-1, Cin)
            out[:, kh, kw] = gd.t() @ xs
    return out


def _check_wgrad(case, dtype, expect, knobs=()):
    from aldi_amd import _lib as L
    from aldi_amd import ops
    N, H, W_, Cin, Cout, k, stride, pad = case
    gen = torch.Generator().manual_seed(sum(case) + 7)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    dev = "cuda"
    x = torch.randn(N, H, W_, Cin, generator=gen).to(dev, dtype)
    g = (torch.randn(N, Ho, Wo, Cout, generator=gen) * 0.1).to(dev, dtype)
    sc = (0.5 + torch.rand(Cout, generator=gen)).to(dev)
    dw0 = torch.randn(Cout, k, k, Cin, generator=gen).to(dev)                  # accumulates INTO an existing gradient
    dw = dw0.clone()
    for name, v in knobs:
        L.set_tuning(name, v)
    ops.conv_wgrad(x, g, dw, KH=k, KW=k, stride=stride, pad=pad, scale=sc)
    which = L.last_dispatch()
    torch.cuda.synchronize()
    assert which.split(" ")[0] == expect, (which, expect)
    ref = dw0.double() + _wgrad_ref(x, g, k, stride, pad) * sc.double().view(-1, 1, 1, 1)
    M = N * Ho * Wo
    e = (dw.double() - ref).abs().max().item()
    assert e <= 3e-6 * max(1.0, ref.abs().max().item()) * max(1.0, (M / 1024) ** 0.5), (which, e, ref.abs().max().item())
    return which


@pytest.mark.parametrize("case,expect", [
    ((4, 200, 336, 256, 256, 3, 1, 1), "wgrad_bf16_big64"),         # FPN output / RPN conv on p2: the 256x256 tile
    ((4, 100, 168, 256, 256, 3, 1, 1), "wgrad_bf16_big64"),         # p3 3x3
    ((4, 200, 336, 256, 256, 1, 1, 0), "wgrad_bf16_lean"),        # FPN lateral p2
    ((4, 50, 84, 256, 256, 3, 1, 1), "wgrad_bf16_lean"),          # res4 conv2
    ((4, 50, 84, 1024, 256, 1, 1, 0), "wgrad_bf16_lean"),         # res4 conv1
    ((4, 100, 168, 128, 512, 1, 1, 0), "wgrad_bf16_lean"),        # res3 conv3
    ((4, 200, 336, 256, 512, 1, 2, 0), "wgrad_bf16_generic"),     # res3.0 shortcut: strided -> gather kernel
    ((2048, 1, 1, 12544, 1024, 1, 1, 0), "wgrad_bf16_big64"),       # box head FC1
    ((4, 200, 336, 256, 16, 1, 1, 0), "wgrad_bf16_lean"),         # RPN heads (Cout 16 < tile)
    ((2, 200, 336, 256, 256, 3, 1, 0), "wgrad_bf16_generic"),     # image-level discriminator conv (no padding), full p2
])
def test_wgrad_fullsize_default_dispatch(case, expect):
    _check_wgrad(case, torch.bfloat16, expect)


@pytest.mark.parametrize("case", [
    (2, 25, 42, 256, 256, 3, 1, 1),        # M = 2100 (ragged last slab), tile-aligned channels
    (1, 19, 23, 256, 512, 1, 1, 0),
    (3, 9, 130, 64, 256, 3, 1, 1),         # K = 576 is not a multiple of 256: falls back to lean even when big is forced
])
def test_wgrad_forced_big_and_generic_small(case):
    N, H, W_, Cin, Cout, k, stride, pad = case
    K = k * k * Cin
    big_ok = Cout % 256 == 0 and K % 256 == 0
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_big64" if big_ok else "wgrad_bf16_lean", knobs=[("wgrad_big_min", 1), ("wgrad_big_slots", 2)])
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_lean", knobs=[("wgrad_big_min", 0)])
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_generic", knobs=[("wgrad_lean", 0)])
    _check_wgrad(case, torch.float32, "wgrad_f32_t128")
    _check_wgrad(case, torch.float32, "wgrad_f32", knobs=[("wgrad_f32_tile128", 0)])


@pytest.mark.parametrize("slots", [1, 64, 1000])
def test_wgrad_split_count_does_not_change_the_result(slots):
    """split-K over pixel ranges: one long split, the default, and many short ones (ragged last split)"""
    case = (2, 50, 84, 256, 256, 3, 1, 1)
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_lean", knobs=[("wgrad_slots", slots), ("wgrad_big_min", 0)])
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_big64", knobs=[("wgrad_big_slots", max(slots, 1)), ("wgrad_big_min", 1)])


@pytest.mark.parametrize("case", [(4, 100, 168, 256, 256, 3, 1, 1), (2, 25, 42, 256, 256, 3, 1, 1), (1, 19, 23, 256, 512, 1, 1, 0), (2048, 1, 1, 1024, 256, 1, 1, 0)])
def test_wgrad_big_tile_register_staged_arm(case):
    """wgrad_dma64 0: the 256x256 tile on the register-staged loop (global -> VGPR -> 8x8 transposes -> LDS) instead of the LDS-DMA + transpose-read one"""
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_big", knobs=[("wgrad_dma64", 0), ("wgrad_big_min", 1), ("wgrad_big_slots", 8)])
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_big64", knobs=[("wgrad_dma64", 1), ("wgrad_big_min", 1), ("wgrad_big_slots", 8)])


def test_wgrad_fp32_fullsize():
    _check_wgrad((2, 100, 168, 256, 256, 3, 1, 1), torch.float32, "wgrad_f32_t128")
    _check_wgrad((2, 100, 168, 256, 256, 3, 1, 1), torch.float32, "wgrad_f32", knobs=[("wgrad_f32_tile128", 0)])


@pytest.mark.parametrize("case,expect", [
    ((44646, 1, 1, 256, 1024, 1, 1, 0), "wgrad_f32_t128"),        # the Deformable-DETR encoder's feed-forward maps at 1333 x 800 (2 images)
    ((44646, 1, 1, 1024, 256, 1, 1, 0), "wgrad_f32_t128"),
    ((600, 1, 1, 256, 92, 1, 1, 0), "wgrad_f32"),                 # class head: Cout < 128
    ((2, 50, 84, 1024, 256, 1, 2, 0), "wgrad_f32_t128"),          # strided shortcut (gather addressing)
    ((2, 23, 37, 132, 200, 3, 1, 1), "wgrad_f32_t128"),           # ragged channels and taps: Cin 132 (K = 1188), Cout 200
    ((2, 200, 336, 64, 64, 3, 1, 1), "wgrad_f32"),
])
def test_wgrad_fp32_dispatch_and_values(case, expect):
    _check_wgrad(case, torch.float32, expect)


# ---------------------------------------------------------------------------------- LDS-DMA + transpose-read weight gradient
DMA_CASES = [
    (2, 25, 42, 256, 256, 3, 1, 1),        # ragged M, border taps, tile-aligned channels
    (1, 19, 23, 256, 512, 1, 1, 0),        # 1x1
    (3, 9, 130, 64, 256, 3, 1, 1),         # Cin = 64: two taps per 128-column tile; rows wider than a slab
    (2, 13, 21, 256, 256, 3, 1, 1),        # p6-sized level: W < 32, a slab spans several image rows
    (4, 1, 1, 1032, 48, 1, 1, 0),          # linear, ragged K and Cout below the tile
    (4, 50, 84, 256, 256, 3, 1, 1),        # res4 conv2 at benchmark scale
    (4, 100, 168, 128, 512, 1, 1, 0),      # res3 conv3
    (4, 200, 336, 256, 256, 3, 1, 1),      # p2 3x3 (the 256x256 tile's shape)
    (2048, 1, 1, 12544, 1024, 1, 1, 0),    # box head FC1
    (4, 200, 336, 256, 16, 1, 1, 0),       # RPN heads
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_wgrad_dma_kernel(case):
    _check_wgrad(case, torch.bfloat16, "wgrad_bf16_dma", knobs=[("wgrad_dma", 2)])


@pytest.mark.parametrize("slots", [1, 1000])
def test_wgrad_dma_split_counts(slots):
    _check_wgrad((2, 50, 84, 256, 256, 3, 1, 1), torch.bfloat16, "wgrad_bf16_dma", knobs=[("wgrad_dma", 2), ("wgrad_slots", slots)])


# ---------------------------------------------------------------------------------- grouped weight gradients (one launch per layer group)
def test_wgrad_group_register_staged_loops():
    """wgrad_dma64 0: both grouped tiles on the register-staged loops of rounds 1-4 (same sums; the default is the LDS-DMA + transpose-read loop)"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(9)
    dev = "cuda"
    cases = [(4, 50, 84, 256, 256, 3, 1, 1), (4, 50, 84, 1024, 256, 1, 1, 0), (2, 100, 168, 128, 128, 3, 1, 1), (2, 100, 168, 512, 128, 1, 1, 0), (1, 37, 41, 72, 136, 1, 1, 0)]
    outs = {}
    for dma in (3, 0):
        L.reset_tuning(); L.set_tuning("wgrad_dma64", dma)
        g2 = torch.Generator().manual_seed(9)
        probs = []
        for (N, H, W_, Cin, Cout, k, stride, pad) in cases:
            x = torch.randn(N, H, W_, Cin, generator=g2).to(dev, torch.bfloat16)
            g = (torch.randn(N, H, W_, Cout, generator=g2) * 0.1).to(dev, torch.bfloat16)
            probs.append((x, g, torch.zeros(Cout, k, k, Cin, device=dev), dict(KH=k, KW=k, stride=stride, pad=pad, db=torch.zeros(Cout, device=dev))))
        ops.conv_wgrad_group(probs)
        name = L.last_dispatch()
        torch.cuda.synchronize()
        assert ("64_group" in name) == bool(dma) and "_group" in name, name
        outs[dma] = [(p[2], p[3]["db"]) for p in probs]
    for (a, ab), (b, bb), case in zip(outs[3], outs[0], cases):
        M = case[0] * case[1] * case[2]
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()) * max(1.0, (M / 1024) ** 0.5), case
        assert (ab - bb).abs().max().item() <= 2e-5 * max(1.0, bb.abs().max().item()) * max(1.0, (M / 1024) ** 0.5), case


@pytest.mark.parametrize("knobs", [(), (("wgrad_big_group", 0),), (("wgrad_ordered", 0),), (("wgrad_big_group", 0), ("wgrad_ordered", 0))],
                         ids=["default", "no-big-group", "atomics", "r02"])
def test_wgrad_group_equals_single_launches(knobs):
    """a res4-like stage (the small GEMMs over the same pixels), a p2 3x3, a strided 1x1 and an fp32-ineligible problem (forwarded to the
    single-problem dispatcher) through aldi_conv_wgrad_group == one aldi_conv_wgrad each; default: the layers whose Cout and K are
    multiples of 256 run as ONE launch of 256x256 tiles, the rest as 128x128 tiles, pixel splits summed in order by a second launch"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(3)
    dev = "cuda"
    cases = [(4, 50, 84, 256, 256, 3, 1, 1), (4, 50, 84, 1024, 256, 1, 1, 0), (4, 50, 84, 256, 1024, 1, 1, 0), (4, 50, 84, 256, 256, 3, 1, 1),
             (4, 25, 42, 512, 512, 3, 1, 1), (2, 200, 336, 256, 256, 3, 1, 1), (4, 100, 168, 256, 512, 1, 2, 0), (4096, 1, 1, 2304, 256, 1, 1, 0),
             (4096, 1, 1, 256, 16, 1, 1, 0)]
    for kn, v in knobs:
        L.set_tuning(kn, v)
    probs, refs = [], []
    for (N, H, W_, Cin, Cout, k, stride, pad) in cases:
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
        x = torch.randn(N, H, W_, Cin, generator=gen).to(dev, torch.bfloat16)
        g = (torch.randn(N, Ho, Wo, Cout, generator=gen) * 0.1).to(dev, torch.bfloat16)
        sc = (0.5 + torch.rand(Cout, generator=gen)).to(dev)
        dw0 = torch.randn(Cout, k, k, Cin, generator=gen).to(dev)
        db0 = torch.randn(Cout, generator=gen).to(dev) if Cout >= 64 and stride == 1 else None
        geo = dict(KH=k, KW=k, stride=stride, pad=pad, scale=sc)
        one, one_b = dw0.clone(), (db0.clone() if db0 is not None else None)
        ops.conv_wgrad(x, g, one, db=one_b, **geo)
        refs.append((one, one_b))
        probs.append((x, g, dw0.clone(), dict(geo, db=db0.clone() if db0 is not None else None)))
    ops.conv_wgrad_group(probs)
    name = L.last_dispatch()
    torch.cuda.synchronize()
    kn = dict(knobs)
    ordered = " ordered" if kn.get("wgrad_ordered", 1) else ""
    if kn.get("wgrad_big_group", 1):     # 9 problems: strided -> generic, Cout 16 -> 128x128 group of one, 7 in the 256x256 group
        assert name.startswith("wgrad_bf16_lean64_group n=1") and "| wgrad_bf16_big64_group n=7" in name and name.endswith(ordered or name[-1]), name
        wgs = int(name.split("wgrad_bf16_big64_group")[1].split("wgs=")[1].split()[0])
        assert 128 <= wgs <= 1024, name         # one to four rounds of one workgroup per CU
    else:                                # r02 form: p2 3x3 -> its own big-tile launch, strided -> generic, 7 grouped
        assert name.startswith("wgrad_bf16_lean64_group n=7") and (ordered in name), name
        wgs = int(name.split("wgs=")[1].split()[0])
        assert 300 <= wgs <= 2400, name          # fewer pixel splits than the single launches
    for (x, g, dw, geo), (ref, ref_b), case in zip(probs, refs, cases):
        e = (dw - ref).abs().max().item()
        assert e <= 2e-5 * max(1.0, ref.abs().max().item()) * max(1.0, (x.shape[0] * x.shape[1] * x.shape[2] / 1024) ** 0.5), (case, e)
        if ref_b is not None:
            eb = (geo["db"] - ref_b).abs().max().item()
            assert eb <= 2e-5 * max(1.0, ref_b.abs().max().item()) * max(1.0, (x.shape[0] * x.shape[1] * x.shape[2] / 1024) ** 0.5), (case, eb)


def test_wgrad_ordered_epilogue_is_reproducible_and_equals_the_fp64_sum():
    """the ordered epilogue (partials to a workspace, splits added in index order) gives the same BITS on every run -- the float-atomic
    one does not promise that -- for a split single launch, an unsplit one, and a group; checked against the fp64 product too"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(11)
    dev = "cuda"
    cases = [(4, 50, 84, 256, 256, 3), (4, 50, 84, 1024, 256, 1), (2, 100, 168, 128, 128, 3), (4, 25, 42, 2048, 512, 1), (300, 1, 1, 1024, 1024, 1)]
    data = []
    for (N, H, W_, Cin, Cout, k) in cases:
        x = torch.randn(N, H, W_, Cin, generator=gen).to(dev, torch.bfloat16)
        g = (torch.randn(N, H, W_, Cout, generator=gen) * 0.1).to(dev, torch.bfloat16)
        data.append((x, g, k, torch.randn(Cout, k, k, Cin, generator=gen).to(dev), torch.randn(Cout, generator=gen).to(dev)))
    def run(group):
        outs = [(dw0.clone(), db0.clone()) for (_, _, _, dw0, db0) in data]
        if group:
            ops.conv_wgrad_group([(x, g, o[0], dict(KH=k, KW=k, stride=1, pad=k // 2, db=o[1])) for (x, g, k, _, _), o in zip(data, outs)])
        else:
            for (x, g, k, _, _), o in zip(data, outs):
                ops.conv_wgrad(x, g, o[0], KH=k, KW=k, stride=1, pad=k // 2, db=o[1])
                assert L.last_dispatch().endswith(" ordered"), L.last_dispatch()
        torch.cuda.synchronize()
        return outs
    for group in (False, True):
        a, b = run(group), run(group)
        for (wa, ba), (wb, bb), (x, g, k, dw0, db0), case in zip(a, b, data, cases):
            assert torch.equal(wa, wb) and torch.equal(ba, bb), (group, case)
            ref = dw0.double() + _wgrad_ref(x, g, k, 1, k // 2)
            M = x.shape[0] * x.shape[1] * x.shape[2]
            assert (wa.double() - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item()) * max(1.0, (M / 1024) ** 0.5), (group, case)
            refb = db0.double() + g.double().reshape(-1, g.shape[-1]).sum(0)
            assert (ba.double() - refb).abs().max().item() <= 3e-6 * max(1.0, refb.abs().max().item()) * max(1.0, (M / 1024) ** 0.5), (group, case)


def test_wgrad_group_layers_sharing_one_gradient_buffer():
    """one conv applied to several pyramid levels (the RPN conv): the problems of the group accumulate into the SAME dw / db -- the
    ordered epilogue's plain read-modify-write would race, those problems keep the atomic one; == the fp64 sum over levels"""
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(21)
    dw = torch.zeros(256, 3, 3, 256, device="cuda")
    db = torch.zeros(256, device="cuda")
    other = torch.zeros(256, 1, 1, 512, device="cuda")
    ref = torch.zeros(256, 3, 3, 256, dtype=torch.float64, device="cuda")
    refb = torch.zeros(256, dtype=torch.float64, device="cuda")
    probs = []
    for (N, H, W_) in [(2, 100, 168), (2, 50, 84), (2, 25, 42), (2, 13, 21)]:
        x = torch.randn(N, H, W_, 256, generator=gen).to("cuda", torch.bfloat16)
        g = (torch.randn(N, H, W_, 256, generator=gen) * 0.1).to("cuda", torch.bfloat16)
        probs.append((x, g, dw, dict(KH=3, KW=3, stride=1, pad=1, db=db)))
        ref += _wgrad_ref(x, g, 3, 1, 1)
        refb += g.double().reshape(-1, 256).sum(0)
    x = torch.randn(2, 50, 84, 512, generator=gen).to("cuda", torch.bfloat16)
    g = (torch.randn(2, 50, 84, 256, generator=gen) * 0.1).to("cuda", torch.bfloat16)
    probs.append((x, g, other, dict(KH=1, KW=1, stride=1, pad=0)))
    ops.conv_wgrad_group(probs)
    torch.cuda.synchronize()
    assert (dw.double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    assert (db.double() - refb).abs().max().item() <= 3e-5 * refb.abs().max().item()
    assert (other.double() - _wgrad_ref(x, g, 1, 1, 0)).abs().max().item() <= 3e-5 * other.abs().max().item()


@pytest.mark.parametrize("slots", [0, 1, 4000])
def test_wgrad_group_split_policy(slots):
    from aldi_amd import _lib as L
    from aldi_amd import ops
    L.set_tuning("wgrad_group_slots", slots)
    gen = torch.Generator().manual_seed(4)
    probs, refs = [], []
    for (N, H, W_, Cin, Cout, k) in [(2, 25, 42, 256, 256, 3), (1, 19, 23, 256, 512, 1), (3, 9, 130, 64, 256, 3)]:
        x = torch.randn(N, H, W_, Cin, generator=gen).to("cuda", torch.bfloat16)
        g = (torch.randn(N, H, W_, Cout, generator=gen) * 0.1).to("cuda", torch.bfloat16)
        dw = torch.zeros(Cout, k, k, Cin, device="cuda")
        refs.append(_wgrad_ref(x, g, k, 1, k // 2))
        probs.append((x, g, dw, dict(KH=k, KW=k, stride=1, pad=k // 2)))
    ops.conv_wgrad_group(probs)
    torch.cuda.synchronize()
    for (x, g, dw, _), ref in zip(probs, refs):
        assert (dw.double() - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item()) * 2


# ---------------------------------------------------------------------------------- grouped convolutions (student + teacher through one layer)
GROUP_CASES = [
    # (H, W, Cin, Cout, k, stride, pad), batch sizes, template of the COMBINED problem
    ((50, 84, 256, 256, 3, 1, 1), (4, 2), "igemm_group2<bf16,128,64,4,1,flat,halo>"),
    ((200, 336, 256, 256, 3, 1, 1), (4, 2), "igemm_group2<bf16,256,256,4,2,halo64>"),
    ((100, 168, 256, 256, 3, 1, 1), (4, 2), "igemm_group2<bf16,256,256,4,2,halo64>"),          # 1575 128x128 tiles together
    ((50, 84, 1024, 256, 1, 1, 0), (4, 2), "igemm_group2<bf16,64,64,2,2,flat,tap,k64>"),
    ((25, 42, 1024, 2048, 1, 2, 0), (4, 2), "igemm_group2<bf16,128,128,2,2,pipe,tap>"),         # strided shortcut
    ((13, 21, 256, 16, 1, 1, 0), (4, 2), "igemm_group2<bf16,128,16,4,1,pipe,tap>"),             # RPN heads on p6: ragged, tiny
    ((19, 23, 64, 96, 3, 1, 1), (3, 1, 2), "igemm_group3<bf16,128,64,4,1,flat,halo>"),
]


@pytest.mark.parametrize("geo,batches,expect", GROUP_CASES)
def test_conv_group_equals_single_launches(geo, batches, expect):
    """n problems of one layer shape (own weights, own batch size, own epilogue tensors) in one launch == n single launches"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    H, W_, Cin, Cout, k, stride, pad = geo
    gen = torch.Generator().manual_seed(sum(geo) + sum(batches))
    dev = "cuda"
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    calls, singles = [], []
    for N in batches:
        x = torch.randn(N, H, W_, Cin, generator=gen).to(dev, torch.bfloat16)
        w = (torch.randn(Cout, k, k, Cin, generator=gen) / (Cin * k * k) ** 0.5).to(dev, torch.bfloat16)
        sc = (0.5 + torch.rand(Cout, generator=gen)).to(dev)
        sh = (torch.randn(Cout, generator=gen) * 0.1).to(dev)
        res = torch.randn(N, Ho, Wo, Cout, generator=gen).to(dev, torch.bfloat16)
        kw = dict(stride=stride, pad=pad, scale=sc, shift=sh, res=res, res_mode=1, relu=True)
        calls.append((x, w, kw))
        singles.append(ops.conv2d(x, w, **kw))
    single_name = L.last_dispatch()
    outs = ops.conv2d_group(calls)
    name = L.last_dispatch()
    torch.cuda.synchronize()
    assert name == expect, (name, single_name)
    for (x, w, kw), a, b in zip(calls, outs, singles):
        assert a.shape == b.shape
        # same products, possibly another tile template (sum order) and therefore another bf16 rounding of a few outputs
        d = (a.float() - b.float()).abs().max().item()
        assert d <= 2e-2 * max(1.0, b.float().abs().max().item()), d
        pix = _sample_pixels(x.shape[0], Ho, Wo)
        ref = _conv_ref_at(x.float().cpu(), w.float().cpu(), pix, stride, pad, Ho, Wo) * kw["scale"].double().cpu() + kw["shift"].double().cpu()
        ref = torch.relu(ref.float().bfloat16().double() + kw["res"].view(-1, Cout)[pix.to(dev)].double().cpu())
        got = a.view(-1, Cout)[pix.to(dev)].double().cpu()
        assert ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item() <= 8e-3
    # fp32 side output variant and the switch-off knob
    calls32 = [(x, w, dict(stride=stride, pad=pad, want_f32=True)) for x, w, _ in calls]
    o32 = ops.conv2d_group(calls32)
    L.set_tuning("igemm_group", 0)
    o32b = ops.conv2d_group(calls32)
    assert not L.last_dispatch().startswith("igemm_group")
    torch.cuda.synchronize()
    for a, b in zip(o32, o32b):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("direct", [0, 7])
@pytest.mark.parametrize("case", [(2, 50, 84, 256, 1024, 1, 0), (1, 19, 23, 64, 72, 3, 1), (2, 25, 42, 512, 2048, 1, 0), (2, 24, 40, 256, 72, 1, 0), (3, 7, 9, 128, 88, 1, 0)])
def test_relu_masks_as_bits(case, direct):
    """aldi_conv_args.bits_out / mask_bits: the forward launch writes (y > 0) of its output as one bit per element beside y, the backward
    launch multiplies by those bits -- the same result, bit for bit, as masking by the bf16 activation itself; ragged M and a Cout that is
    not a tile multiple included"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    L.set_tuning("igemm_direct", direct)          # 0: the staged epilogue everywhere; 7: the direct epilogue where it applies
    N, H, W_, Cin, Cout, k, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W_, Cin, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to("cuda", torch.bfloat16)
    res = torch.randn(N, H, W_, Cout, generator=g).to("cuda", torch.bfloat16)
    sh = torch.randn(Cout, generator=g).to("cuda")
    M = N * H * W_
    bits = torch.full((M * Cout // 8 + 64,), 0xA5, dtype=torch.uint8, device="cuda")          # (guard bytes behind the mask)
    y = ops.conv2d(x, w, pad=pad, shift=sh, res=res, res_mode=1, relu=True, bits_out=bits)
    y_plain = ops.conv2d(x, w, pad=pad, shift=sh, res=res, res_mode=1, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_plain)
    expect = (y.view(M, Cout // 8, 8).float() > 0).to(torch.uint8)
    packed = (expect * (2 ** torch.arange(8, device="cuda", dtype=torch.uint8))).sum(-1).to(torch.uint8).view(-1)
    assert torch.equal(bits[:M * Cout // 8], packed) and bool((bits[M * Cout // 8:] == 0xA5).all())
    assert 0.2 < float(expect.float().mean()) < 0.8
    # backward: a data-gradient launch masked by the bits == masked by the activation
    gy = torch.randn(N, H, W_, Cin, generator=g).to("cuda", torch.bfloat16)
    wt = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to("cuda", torch.bfloat16)
    r2 = torch.randn(N, H, W_, Cout, generator=g).to("cuda", torch.bfloat16)
    a = ops.conv2d(gy, wt, res=r2, res_mode=1, mask=y)
    b = ops.conv2d(gy, wt, res=r2, res_mode=1, mask_bits=bits)
    torch.cuda.synchronize()
    if direct == 0:
        assert torch.equal(a, b)
    else:
        # the direct epilogue adds the residual in fp32 and rounds once; the staged one (which the `mask` tensor form always takes) rounds the
        # product first: the same masked zeros, values within the product's bf16 rounding (an exact cancellation may round to zero in one only)
        assert bool((b[y == 0] == 0).all()) and bool((a[y == 0] == 0).all())
        prod = ops.conv2d(gy, wt, want_f32=True)
        assert bool(((a.float() - b.float()).abs() <= 2.0 ** -7 * prod.abs() + 2.0 ** -7 * b.float().abs() + 1e-30).all())
    with pytest.raises(Exception):
        ops.conv2d(gy, wt, mask=y, mask_bits=bits)


def test_conv_group_of_different_layers_falls_back():
    from aldi_amd import _lib as L
    from aldi_amd import ops
    x1 = torch.randn(2, 20, 24, 64, device="cuda").bfloat16()
    x2 = torch.randn(2, 10, 12, 32, device="cuda").bfloat16()
    w1 = torch.randn(64, 3, 3, 64, device="cuda").bfloat16()
    w2 = torch.randn(64, 3, 3, 32, device="cuda").bfloat16()
    a, b = ops.conv2d_group([(x1, w1, dict(pad=1)), (x2, w2, dict(pad=1))])          # different Cin: not one layer shape
    assert not L.last_dispatch().startswith("igemm_group")
    torch.cuda.synchronize()
    assert torch.equal(a, ops.conv2d(x1, w1, pad=1)) and torch.equal(b, ops.conv2d(x2, w2, pad=1))


@pytest.mark.parametrize("k,pad", [(3, 1), (1, 0)])
def test_conv_group_over_pyramid_levels(k, pad):
    """one layer (shared or per-level weights of the same shape) on maps of DIFFERENT H x W and batch size -- the FPN output convs, the
    RPN conv on p2..p6, student and teacher together -- is one launch (up to 12 problems), bit-identical to the single launches where both
    take tiles of one K order (the 3x3 group is big enough for the 128-byte-slab tile, which sums the 64 channels of a chunk tap by tap: its
    outputs agree with the single launches' 32-channel order to one bf16 rounding)"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(31)
    calls = []
    for (N, H, W_) in [(4, 100, 168), (4, 50, 84), (4, 25, 42), (4, 13, 21), (2, 100, 168), (2, 50, 84), (2, 25, 42), (2, 13, 21), (1, 7, 11)]:
        x = torch.randn(N, H, W_, 256, generator=gen).to("cuda", torch.bfloat16)
        w = (torch.randn(256, k, k, 256, generator=gen) * 0.05).to("cuda", torch.bfloat16)
        sh = torch.randn(256, generator=gen).to("cuda")
        calls.append((x, w, dict(pad=pad, shift=sh, relu=True)))
    outs = ops.conv2d_group(calls)
    name = L.last_dispatch()
    torch.cuda.synchronize()
    assert name.startswith("igemm_group9<"), name
    assert ("halo64" in name) == (k == 3), name
    for (x, w, kw), y in zip(calls, outs):
        ref = ops.conv2d(x, w, **kw)
        if "halo64" in name:
            assert bool(((y.float() - ref.float()).abs() <= 2.0 ** -7 * ref.float().abs() + 2e-3).all()), (tuple(x.shape), name, L.last_dispatch())
        else:
            assert torch.equal(y, ref), (tuple(x.shape), name, L.last_dispatch())
    outs13 = ops.conv2d_group(calls + calls[:4])            # more than 12: single launches
    assert not L.last_dispatch().startswith("igemm_group")
    torch.cuda.synchronize()
    for a, b in zip(outs13, outs + outs[:4]):
        if "halo64" in name:
            assert bool(((a.float() - b.float()).abs() <= 2.0 ** -7 * b.float().abs() + 2e-3).all())
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(2, 50, 84, 256, 512), (1, 25, 41, 128, 64), (3, 13, 21, 64, 256)])
def test_strided_1x1_wgrad_equals_gemm_on_subsampled_input(shape):
    """the engine turns the weight gradient of a stride-2 1x1 conv into a plain GEMM over the subsampled input
    (ops.subsample2) so that it can join a grouped launch: same numbers as the strided gather kernel, odd sizes included"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    L.reset_tuning()
    N, H, W_, Cin, Cout = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W_, Cin, generator=gen).to("cuda", torch.bfloat16)
    Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    g = (torch.randn(N, Ho, Wo, Cout, generator=gen) * 0.1).to("cuda", torch.bfloat16)
    a = torch.zeros(Cout, 1, 1, Cin, device="cuda")
    b = torch.zeros(Cout, 1, 1, Cin, device="cuda")
    ops.conv_wgrad(x, g, a, KH=1, KW=1, stride=2, pad=0)
    xs = ops.subsample2(x)
    assert torch.equal(xs, x[:, ::2, ::2].contiguous())
    ops.conv_wgrad(xs, g, b, KH=1, KW=1, stride=1, pad=0)
    torch.cuda.synchronize()
    ref = _wgrad_ref(x, g, 1, 2, 0)
    scale = max(1.0, ref.abs().max().item()) * max(1.0, (N * Ho * Wo / 1024) ** 0.5)
    assert (a.double().view_as(ref) - ref).abs().max().item() <= 2e-5 * scale
    assert (b.double().view_as(ref) - ref).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("M,Cout,K,ks", [(2048, 1024, 12544, 4), (1000, 1024, 12544, 4), (300, 256, 4096, 2), (2048, 1024, 12544, 7)])
def test_splitk_linear_equals_plain(M, Cout, K, ks):
    """split-K for long-K linear layers (the box head's FC1): K slices on ksplit times the workgroups, fp32 partial tiles summed in
    slice order by a second launch -- against the unsplit kernel and an fp64 evaluation; two runs are bit-identical (no atomics)"""
    from aldi_amd import _lib as L, ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, 1, 1, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 1, 1, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    L.reset_tuning()
    y0 = ops.conv2d(x, w, scale=sc, shift=sh, relu=True, ksplit=0)
    name0 = L.last_dispatch()
    if K % (64 * ks):
        with pytest.raises(L.AldiHipError):
            ops.conv2d(x, w, scale=sc, shift=sh, relu=True, ksplit=ks)
        return
    y1 = ops.conv2d(x, w, scale=sc, shift=sh, relu=True, ksplit=ks)
    name1 = L.last_dispatch()
    y2 = ops.conv2d(x, w, scale=sc, shift=sh, relu=True, ksplit=ks)
    torch.cuda.synchronize()
    assert "splitk" in name1 and "splitk" not in name0, (name0, name1)
    # default tile of the split launch: 256x128 (8 waves) when that still gives about one workgroup per CU, else 128x128 with 128-byte slabs
    wide = ((M + 255) // 256) * ((Cout + 127) // 128) * ks >= 200
    assert name1.startswith("igemm<bf16,256,128,4,2,flat,tap,k64>" if wide else "igemm<bf16,128,128,2,2,flat,tap,k64>"), name1
    assert torch.equal(y1, y2)
    for tile, tname in ((0, "igemm<bf16,128,128,2,2,flat,tap,k64>"), (1, "igemm<bf16,256,128,4,2,flat,tap>"), (3, "igemm<bf16,256,128,4,2,flat,tap,k64>")):     # every tile, forced
        L.set_tuning("igemm_splitk_tile", tile)
        yt = ops.conv2d(x, w, scale=sc, shift=sh, relu=True, ksplit=ks)
        assert L.last_dispatch().startswith(tname), L.last_dispatch()
        assert float((yt.float() - y0.float()).abs().max()) <= 1e-2 * float(y0.float().abs().max())
    L.reset_tuning()
    rows = torch.arange(0, M, max(M // 64, 1), device="cuda")
    ref = torch.relu((x.view(M, K)[rows].double() @ w.view(Cout, K).double().t()) * sc.double() + sh.double())
    for y in (y0, y1):
        assert float((y.view(M, Cout)[rows].double() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert float((y0.float() - y1.float()).abs().max()) <= 1e-2 * float(y0.float().abs().max())
    # the heuristic takes FC1's shape and leaves a short-K layer alone
    ops.conv2d(x, w, relu=True)
    assert ("splitk" in L.last_dispatch()) == (K >= 4096 and K % 256 == 0 and 16 <= ((M + 127) // 128) * ((Cout + 127) // 128) <= 160)


@pytest.mark.parametrize("case", ["lean_1x1", "lean_3x3", "big_3x3", "generic_stride2", "fp32", "group"])
def test_bias_gradient_rides_in_the_wgrad_launch(case):
    """aldi_wgrad_args.db: db[co] += sum over pixels of g[p][co] in the weight-gradient call itself -- as one more MFMA column
    (a constant ones fragment) in the lean / 256x256 bf16 kernels, through the column-sum launch behind the others -- against a
    float64 column sum; the weight gradient itself must not change."""
    from aldi_amd import _lib as L, ops
    torch.manual_seed(3)
    dt = torch.float32 if case == "fp32" else torch.bfloat16
    N, H, W, Cin, Cout, k, s = {"lean_1x1": (2, 50, 84, 256, 512, 1, 1), "lean_3x3": (2, 25, 42, 128, 128, 3, 1), "big_3x3": (2, 200, 336, 256, 256, 3, 1),
                                "generic_stride2": (2, 50, 84, 256, 512, 1, 2), "fp32": (1, 25, 42, 64, 64, 3, 1), "group": (2, 50, 84, 256, 256, 1, 1)}[case]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    g = torch.randn(N, Ho, Wo, Cout, device="cuda").to(dt)
    ref = g.double().sum((0, 1, 2))
    L.reset_tuning()
    dw0 = torch.zeros(Cout, k, k, Cin, device="cuda")
    dw1 = torch.zeros_like(dw0)
    db = torch.full((Cout,), 0.5, device="cuda")                       # (accumulates: starts from a non-zero value)
    geo = dict(KH=k, KW=k, stride=s, pad=k // 2)
    ops.conv_wgrad(x, g, dw0, **geo)
    if case == "group":
        x2 = torch.randn(N, H, W, 128, device="cuda").to(dt)
        dw2, db2 = torch.zeros(Cout, 1, 1, 128, device="cuda"), torch.zeros(Cout, device="cuda")
        ops.conv_wgrad_group([(x, g, dw1, dict(geo, db=db)), (x2, g, dw2, dict(geo, db=db2))])
        assert "group" in L.last_dispatch()
        assert float((db2.double() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-2
    else:
        ops.conv_wgrad(x, g, dw1, db=db, **geo)
        want = {"lean_1x1": "wgrad_bf16_lean", "lean_3x3": "wgrad_bf16_lean", "big_3x3": "wgrad_bf16_big64", "generic_stride2": "wgrad_bf16_generic", "fp32": "wgrad_f32"}[case]
        assert L.last_dispatch().startswith(want), L.last_dispatch()
    torch.cuda.synchronize()
    assert float((db.double() - 0.5 - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-2, float((db.double() - 0.5 - ref).abs().max())
    assert float((dw0 - dw1).abs().max()) <= 1e-4 * float(dw0.abs().max())      # (fp32 atomics: summation order only)


# ---------------------------------------------------------------------------------- the direct epilogue (igemm_epilogue_direct)
def _bits_of(t):
    M = t.numel() // t.shape[-1]
    e = (t.reshape(M, t.shape[-1] // 8, 8).float() > 0).to(torch.uint8)
    return (e * (2 ** torch.arange(8, device=t.device, dtype=torch.uint8))).sum(-1).to(torch.uint8).view(-1)


def _check_direct(case, kind, expect, *, force=None):
    """one launch through the direct epilogue against an fp64 evaluation of  y = mask * relu(conv * scale + shift + residual)  rounded ONCE to
    bf16 (sampled pixels: borders, seams, tile edges, the ragged tail), the mask bits it writes against (y > 0) of the whole tensor, masked
    positions exactly zero.  kinds: f3 forward conv3 (scale, shift, residual, ReLU, bits_out) / f1 forward conv1, conv2 (scale, shift, ReLU,
    bits_out) / sc shortcut (scale, shift) / b bias only / d1 conv1 dgrad (residual + mask bits) / d3 conv3, conv2 dgrad (mask bits) /
    up FPN lateral (bias + the coarser level's map, nearest-upsampled) / f1x scale, shift, ReLU without bits / n no epilogue operand"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    N, H, W_, Cin, Cout, k, stride, pad = case
    x, w, g = _mk(N, H, W_, Cin, Cout, k, sum(case) + len(kind), torch.bfloat16)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    M = N * Ho * Wo
    scale, shift = 0.5 + torch.rand(Cout, generator=g), torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(N, Ho, Wo, Cout, generator=g).bfloat16().float()
    up = torch.randn(N, Ho // 2, Wo // 2, Cout, generator=g).bfloat16().float()
    act = torch.relu(torch.randn(N, Ho, Wo, Cout, generator=g)).bfloat16()
    dev = "cuda"
    xd, wd = x.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16)
    kw = dict(stride=stride, pad=pad)
    bits = torch.full((M * Cout // 8 + 64,), 0xA5, dtype=torch.uint8, device=dev)
    mb = _bits_of(act.to(dev))
    use_scale = kind in ("f3", "f1", "f1x", "sc")
    use_shift = kind in ("f3", "f1", "f1x", "sc", "b", "up")
    if use_scale:
        kw.update(scale=scale.to(dev))
    if use_shift:
        kw.update(shift=shift.to(dev))
    if kind in ("f3", "d1"):
        kw.update(res=res.to(dev, torch.bfloat16), res_mode=1)
    if kind == "up":
        kw.update(res=up.to(dev, torch.bfloat16), res_mode=2)
    if kind in ("f3", "f1"):
        kw.update(relu=True, bits_out=bits)
    if kind == "f1x":
        kw.update(relu=True)
    if kind in ("d1", "d3"):
        kw.update(mask_bits=mb)
    if force is not None:
        L.set_tuning("igemm_force", force)
    y = ops.conv2d(xd, wd, **kw)
    name = L.last_dispatch()
    torch.cuda.synchronize()
    assert name == expect, (name, expect)
    pix = _sample_pixels(N, Ho, Wo)
    ref = _conv_ref_at(x, w, pix, stride, pad, Ho, Wo)
    mag = _conv_ref_at(x.abs(), w.abs(), pix, stride, pad, Ho, Wo)      # sum of |products|: what the fp32 accumulation noise scales with
    if use_scale:
        ref = ref * scale.double()
        mag = mag * scale.double()
    if use_shift:
        ref = ref + shift.double()
        mag = mag + shift.double().abs()
    if kind in ("f3", "d1"):
        r_ = res.view(-1, Cout)[pix].double()
        ref, mag = ref + r_, mag + r_.abs()
    if kind == "up":
        n_, r2 = pix // (Ho * Wo), pix % (Ho * Wo)
        r_ = up[n_, (r2 // Wo) // 2, (r2 % Wo) // 2].double()
        ref, mag = ref + r_, mag + r_.abs()
    if kind in ("f3", "f1", "f1x"):
        ref = torch.relu(ref)
    if kind in ("d1", "d3"):
        ref = ref * (act.view(-1, Cout)[pix] > 0)
    got = y.view(-1, Cout)[pix.to(dev)].double().cpu()
    # ONE rounding of the fp32 sum: half a bf16 ulp of the result + the fp32 accumulation noise of the terms
    err = (got - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -21 * mag + 1e-30
    assert bool((err <= bound).all()), (float((err / bound).max()), float(err.max()))
    assert torch.isfinite(y.float()).all()
    if "bits_out" in kw:
        assert torch.equal(bits[:M * Cout // 8], _bits_of(y)) and bool((bits[M * Cout // 8:] == 0xA5).all())
        assert 0.1 < float((y > 0).float().mean()) < 0.9
    if "mask_bits" in kw:
        assert bool((y[act.to(dev) == 0] == 0).all())
    return name


DIRECT_FULL = [
    # (N, H, W, Cin, Cout, k, stride, pad), kind, the kernel the DEFAULT dispatch picks
    ((4, 50, 84, 256, 1024, 1, 1, 0), "f3", "igemm<bf16,128,64,4,1,pipe,tap,direct+res>"),       # res4 conv3
    ((4, 100, 168, 128, 512, 1, 1, 0), "f3", "igemm_ws<bf16,32,256,k128>"),                      # res3 conv3 (>= 40 000 pixels: the weight-stationary persistent kernel, igemm_ws.h)
    ((2, 25, 42, 512, 2048, 1, 1, 0), "f3", "igemm<bf16,128,64,4,1,pipe,tap,direct+res>"),       # res5 conv3, teacher
    ((4, 50, 84, 1024, 256, 1, 1, 0), "f1", "igemm<bf16,64,64,2,2,flat,tap,k64,direct>"),        # res4 conv1
    ((4, 100, 168, 512, 128, 1, 1, 0), "f1", "igemm_ws<bf16,16,128,k512>"),                      # res3 conv1
    ((4, 50, 84, 256, 256, 3, 1, 1), "f1", "igemm<bf16,128,64,4,1,flat,halo,direct>"),           # res4 conv2
    ((4, 100, 168, 512, 1024, 1, 2, 0), "sc", "igemm<bf16,128,64,4,1,pipe,tap,direct>"),         # res4 shortcut (stride 2)
    ((4, 50, 84, 256, 1024, 1, 1, 0), "d1", "igemm<bf16,128,64,4,1,pipe,tap,direct+res>"),       # res4 conv1 dgrad (+ skip gradient, mask of the block input)
    ((4, 100, 168, 128, 512, 1, 1, 0), "d1", "igemm_ws<bf16,32,256,k128>"),                      # res3 conv1 dgrad
    ((4, 100, 168, 512, 128, 1, 1, 0), "d3", "igemm_ws<bf16,16,128,k512>"),                      # res3 conv3 dgrad
    ((4, 50, 84, 1024, 256, 1, 1, 0), "d3", "igemm<bf16,64,64,2,2,flat,tap,k64,direct>"),        # res4 conv3 dgrad
    ((4, 50, 84, 256, 256, 3, 1, 1), "d3", "igemm<bf16,128,64,4,1,flat,halo,direct>"),           # res4 conv2 dgrad
    ((4, 100, 168, 512, 256, 1, 1, 0), "up", "igemm<bf16,128,64,4,1,pipe,tap,direct+res>"),      # FPN lateral 3 (+ upsampled top-down map)
    ((4, 25, 42, 2048, 256, 1, 1, 0), "b", "igemm<bf16,64,64,2,2,flat,tap,k64,direct>"),         # FPN lateral 5 (bias only)


    ((4, 200, 336, 256, 256, 3, 1, 1), "f1x", "igemm<bf16,256,256,4,2,halo64,direct>"),        # RPN conv on p2 (bias + ReLU, no bits)
    ((2, 200, 336, 256, 256, 3, 1, 1), "b", "igemm<bf16,256,256,4,2,halo64,direct>"),          # FPN output conv on p2, teacher (bias only)
    ((4, 200, 336, 256, 256, 3, 1, 1), "n", "igemm<bf16,256,256,4,2,halo64,direct>"),          # their data gradients (no epilogue operand)
]


@pytest.mark.parametrize("case,kind,expect", DIRECT_FULL)
def test_direct_epilogue_default_dispatch_fullsize(case, kind, expect):
    _check_direct(case, kind, expect)


DIRECT_SMALL = [
    (2, 25, 42, 64, 96, 3, 1, 1),      # halo, ragged M (2100), Cout = one and a half tiles
    (1, 19, 23, 128, 256, 3, 1, 1),    # halo, tiny image
    (2, 24, 40, 256, 72, 1, 1, 0),     # 1x1, Cout 72: the second channel tile holds 8 channels
    (2, 25, 41, 256, 136, 1, 2, 0),    # strided 1x1, ragged everything
    (3, 7, 9, 64, 200, 1, 1, 0),       # two K slabs (64-channel slabs: one), M = 189
    (1, 5, 6, 32, 64, 1, 1, 0),        # ONE K slab (the once-per-tile pieces are waited for after the loop), one partial tile
    (130, 1, 1, 2048, 136, 1, 1, 0),   # a linear layer
]


@pytest.mark.parametrize("kind", ["f3", "f1", "sc", "b", "d1", "d3"])
@pytest.mark.parametrize("case", DIRECT_SMALL)
def test_direct_epilogue_forced_templates(case, kind):
    """each direct arm forced onto ragged shapes: the 128x64 tap tile (with and without the residual prefetch), the 64x64 long-K tile, the
    128x64 halo tile"""
    N, H, W_, Cin, Cout, k, stride, pad = case
    res = kind in ("f3", "d1")
    # mask bits reach the direct epilogue by a 4-byte LDS-DMA at bit offset (m * Cout + channel): dword-aligned for Cout % 32 == 0 only --
    # other widths take the staged epilogue (byte loads)
    direct = not (kind in ("d1", "d3") and Cout % 32)
    if not direct and res:
        pytest.skip("staged epilogue with a residual rounds the product before the add: _check_direct's single-rounding bound does not apply (test_dgrad_* cover it)")
    if k == 3:
        if res:
            pytest.skip("the halo tile has no residual prefetch (no 3x3 layer of the step has a residual)")
        _check_direct(case, kind, "igemm<bf16,128,64,4,1,flat,halo%s>" % (",direct" if direct else ""), force=2)
        return
    _check_direct(case, kind, "igemm<bf16,128,64,4,1,pipe,tap%s>" % ((",direct" + ("+res" if res else "")) if direct else ""), force=2)
    if not res and stride == 1:                  # (the 64-channel-slab tile takes plain 1x1 / linear layers only)
        from aldi_amd import _lib as L
        L.reset_tuning()
        _check_direct(case, kind, "igemm<bf16,64,64,2,2,flat,tap,k64%s>" % (",direct" if direct else ""), force=8)


WS_SMALL = [
    (3, 7, 9, 64, 256, 1, 1, 0),       # K = 64 (one 128-byte slab), M = 189: six pixel tiles of 32, the last one ragged
    (2, 25, 41, 128, 512, 1, 1, 0),    # K = 128, two channel groups, M = 2050
    (1, 19, 23, 256, 256, 1, 1, 0),    # K = 256 (the weights fill 128 registers per lane), M = 437
    (2, 13, 30, 512, 128, 1, 1, 0),    # K = 512: 16-pixel tiles, one channel group
    (130, 1, 1, 128, 768, 1, 1, 0),    # a linear layer, three channel groups
]


@pytest.mark.parametrize("kind", ["f3", "f1", "f1x", "sc", "b", "n", "d1", "d3", "up"])
@pytest.mark.parametrize("case", WS_SMALL + [(2, 26, 42, 256, 256, 1, 1, 0), (3, 6, 10, 512, 128, 1, 1, 0)])
def test_weight_stationary_kernel_forced_on_ragged_shapes(case, kind):
    if kind == "up" and (case[1] % 2 or case[2] % 2):
        pytest.skip("the upsampled residual needs even output sizes")
    """igemm_ws.h (weights in registers, pixel tiles through a 3-stage LDS ring, persistent workgroups) with every epilogue operand set of the
    bottleneck layers: more pixel-tile sequences than tiles, ragged last tiles, one to three channel groups"""
    K = case[3]
    _check_direct(case, kind, "igemm_ws<bf16,%d,%d,k%d>" % (32 if K <= 256 else 16, 128 if K >= 256 else 256, K), force=14)


@pytest.mark.parametrize("wgs", [8, 64, 4096])
def test_weight_stationary_kernel_workgroup_count_does_not_change_the_result(wgs):
    """igemm_ws_wgs: 8 workgroups walk 263 pixel tiles each ... one tile each; the same bits as the tile kernel's direct epilogue (same K order, one rounding)"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2, 50, 84, 128, device="cuda", generator=g).bfloat16()
    w = (torch.randn(512, 1, 1, 128, device="cuda", generator=g) / 11).bfloat16()
    r = torch.randn(2, 50, 84, 512, device="cuda", generator=g).bfloat16()
    sc, sh = torch.rand(512, device="cuda", generator=g) + 0.5, torch.randn(512, device="cuda", generator=g)
    kw = dict(scale=sc, shift=sh, res=r, res_mode=1, relu=True)
    L.set_tuning("igemm_ws", 0)
    b0 = torch.zeros(8400 * 64, dtype=torch.uint8, device="cuda")
    y0 = ops.conv2d(x, w, bits_out=b0, **kw)
    assert L.last_dispatch() == "igemm<bf16,128,64,4,1,pipe,tap,direct+res>"
    L.set_tuning("igemm_ws", 1)
    L.set_tuning("igemm_ws_min", 0)
    L.set_tuning("igemm_ws_wgs", wgs)
    b1 = torch.zeros(8400 * 64, dtype=torch.uint8, device="cuda")
    y1 = ops.conv2d(x, w, bits_out=b1, **kw)
    assert L.last_dispatch() == "igemm_ws<bf16,32,256,k128>"
    assert torch.equal(y0, y1) and torch.equal(b0, b1)


def test_weight_stationary_kernel_eligibility():
    """what it does not take stays on the tile kernels: small maps (igemm_ws_min), an upsampled residual, K = 1024, channel counts that are not whole groups"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    mk = lambda *s_: torch.randn(*s_, device="cuda", generator=g).bfloat16()
    ops.conv2d(mk(2, 100, 168, 128), mk(512, 1, 1, 128))                    # 33 600 pixels (the teacher's res3): below igemm_ws_min = 40 000
    assert not L.last_dispatch().startswith("igemm_ws")
    L.set_tuning("igemm_ws_min", 4096)
    ops.conv2d(mk(1, 40, 50, 128), mk(512, 1, 1, 128))                      # 2000 pixels
    assert not L.last_dispatch().startswith("igemm_ws")
    ops.conv2d(mk(2, 50, 84, 512), mk(256, 1, 1, 512), res=mk(2, 25, 42, 256), res_mode=2)
    assert not L.last_dispatch().startswith("igemm_ws")                     # an FPN lateral + the upsampled top-down map: from 4 x igemm_ws_min pixels
    ops.conv2d(mk(2, 100, 168, 256), mk(256, 1, 1, 256), res=mk(2, 50, 84, 256), res_mode=2)
    assert L.last_dispatch() == "igemm_ws<bf16,32,128,k256>"
    ops.conv2d(mk(2, 50, 84, 512), mk(256, 1, 1, 512), want_f32=True)
    assert not L.last_dispatch().startswith("igemm_ws")                     # fp32 output
    ops.conv2d(mk(2, 50, 84, 1024), mk(256, 1, 1, 1024))
    assert not L.last_dispatch().startswith("igemm_ws")
    ops.conv2d(mk(2, 50, 84, 128), mk(384, 1, 1, 128))
    assert not L.last_dispatch().startswith("igemm_ws")                     # 384 channels = one and a half groups of 256
    ops.conv2d(mk(2, 50, 84, 128), mk(512, 1, 1, 128))
    assert L.last_dispatch() == "igemm_ws<bf16,32,256,k128>"


HALO64_SMALL = [
    (2, 25, 42, 64, 96, 3, 1, 1),      # one 64-channel chunk per kernel row (9 taps), ragged M (2100 = 8 tiles + 52 pixels), Cout < the tile
    (1, 19, 23, 128, 256, 3, 1, 1),    # tiny image: every tile crosses many image rows
    (3, 9, 130, 64, 264, 3, 1, 1),     # wide rows, Cout = one tile + 8 channels
    (2, 13, 300, 192, 256, 3, 1, 1),   # three chunks per kernel row (odd group count: the slab stages flip between tiles), image rows longer than a tile
]


@pytest.mark.parametrize("kind", ["f1x", "b", "n", "sc", "f1", "d3"])
@pytest.mark.parametrize("case", HALO64_SMALL)
def test_halo64_forced_on_ragged_shapes(case, kind):
    """the 256 x 256 tile with 128-byte K slabs (igemm_halo64.h): direct epilogue for scale / shift / ReLU outputs, the staged one for mask bits"""
    direct = kind in ("f1x", "b", "n", "sc")
    _check_direct(case, kind, "igemm<bf16,256,256,4,2,halo64%s>" % (",direct" if direct else ""), force=11)


@pytest.mark.parametrize("kind", ["f1x", "n", "d3"])
@pytest.mark.parametrize("case", HALO64_SMALL)
def test_halo64_mid_tile_forced_on_ragged_shapes(case, kind):
    """the same loop on the 128 x 128 tile (4 waves, 64 x 64 per wave: two sub-phases per tap), igemm_force 13 / igemm_halo64_mid"""
    direct = kind in ("f1x", "n")
    _check_direct(case, kind, "igemm<bf16,128,128,2,2,halo64%s>" % (",direct" if direct else ""), force=13)


@pytest.mark.parametrize("kind", ["f1x", "n"])
@pytest.mark.parametrize("case", HALO64_SMALL[:3])
def test_halo64_four_wave_tile_forced_on_ragged_shapes(case, kind):
    """igemm_force 15: the 256 x 256 tile on FOUR waves of 128 pixels x 128 channels (accumulators in the AGPR half of a 512-register budget, two DMA
    pieces per quarter of a slab / tap); measured slower than the 8-wave form (one wave per SIMD issues an MFMA every 27 clocks, two every 17.5:
    profiles/r05_mfma_clock_probe.txt) -- kept as a tested arm"""
    _check_direct(case, kind, "igemm<bf16,256,256,2,2,halo64,direct>", force=15)


def test_halo64_mid_knob_selects_the_tile():
    from aldi_amd import _lib as L
    from aldi_amd import ops
    x = torch.randn(4, 50, 84, 256, device="cuda").bfloat16()
    w = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16()
    ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,128,64,4,1,flat,halo,direct>"            # default: off (measured slower on the mid-size layers)
    L.set_tuning("igemm_halo64_mid", 256)
    ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,128,128,2,2,halo64,direct>"


@pytest.mark.parametrize("case", HALO64_SMALL[:2])
def test_halo64_staged_epilogue_all_operands(case):
    _check_forward(case, torch.bfloat16, "igemm<bf16,256,256,4,2,halo64>", force=11, full=True)


def test_halo64_group_over_pyramid_levels():
    """the grouped launch of the step (one layer over the pyramid levels) on the halo64 tile == the single launches"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(5)
    dev = "cuda"
    w = (torch.randn(256, 3, 3, 256, generator=gen) / 48.0).to(dev, torch.bfloat16)
    sh = (torch.randn(256, generator=gen) * 0.1).to(dev)
    calls = []
    for (N, H, W_) in ((2, 200, 336), (2, 100, 168), (2, 50, 84), (2, 25, 42), (2, 13, 21)):
        calls.append((torch.randn(N, H, W_, 256, generator=gen).to(dev, torch.bfloat16), w, dict(pad=1, shift=sh, relu=True)))
    outs = ops.conv2d_group(calls)
    assert L.last_dispatch() == "igemm_group5<bf16,256,256,4,2,halo64,direct>", L.last_dispatch()
    L.set_tuning("igemm_bigtile", 4)
    refs = ops.conv2d_group(calls)
    assert L.last_dispatch() == "igemm_group5<bf16,256,128,4,2,flat,halo>", L.last_dispatch()
    torch.cuda.synchronize()
    for a, b in zip(outs, refs):
        assert (a.float() - b.float()).abs().max().item() <= 2e-2 * max(1.0, b.float().abs().max().item())
        assert bool(((a.float() - b.float()).abs() <= 2.0 ** -7 * b.float().abs() + 1e-3).all())


def test_direct_epilogue_upsampled_residual_ragged():
    _check_direct((2, 26, 42, 256, 256, 1, 1, 0), "up", "igemm<bf16,128,64,4,1,pipe,tap,direct+res>", force=2)


def test_direct_epilogue_off_restores_the_staged_kernels():
    from aldi_amd import _lib as L
    L.set_tuning("igemm_direct", 0)
    from aldi_amd import ops
    x = torch.randn(2, 25, 42, 256, device="cuda").bfloat16()
    w = (torch.randn(1024, 1, 1, 256, device="cuda") / 16).bfloat16()
    x = torch.randn(4, 50, 84, 256, device="cuda").bfloat16()
    ops.conv2d(x, w, relu=True)
    assert L.last_dispatch() == "igemm<bf16,128,64,4,1,pipe,tap>"


# ---------------------------------------------------------------------------------- r06: interleaved K loops (asm MFMAs tied in place, reads / DMA between them)
@pytest.mark.parametrize("case", [(4, 200, 336, 256, 256), (2, 100, 168, 256, 256), (1, 37, 41, 192, 256), (2, 13, 21, 64, 512)])
def test_halo64_interleaved_loop_equals_lockstep_bit_for_bit(case):
    """igemm_halo_ilv 1 (default) vs 0 on the 256 x 256 halo64 tile: same products in the same order per accumulator -> identical bits; also with the weight
    taps requested at the lockstep loop's DMA points (igemm_dbg 1024) and with the staged epilogue"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    N, H, W_, Cin, Cout = case
    gen = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W_, Cin, generator=gen).to("cuda", torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=gen) / (9 * Cin) ** 0.5).to("cuda", torch.bfloat16)
    sh = (torch.randn(Cout, generator=gen) * 0.1).to("cuda")
    outs = {}
    for direct in (8, 0):
        for ilv, dbg in ((1, 0), (0, 0), (1, 1024)):
            L.reset_tuning(); L.set_tuning("igemm_force", 11); L.set_tuning("igemm_halo_ilv", ilv); L.set_tuning("igemm_dbg", dbg); L.set_tuning("igemm_direct", direct)
            outs[(direct, ilv, dbg)] = ops.conv2d(x, w, pad=1, shift=sh, relu=True)
            name = L.last_dispatch()
            assert ("lockstep" in name) == (ilv == 0) and ("direct" in name) == bool(direct), name
        torch.cuda.synchronize()
        assert torch.equal(outs[(direct, 1, 0)], outs[(direct, 0, 0)])
        assert torch.equal(outs[(direct, 1, 0)], outs[(direct, 1, 1024)])
    assert float(outs[(8, 1, 0)].float().abs().max()) > 0
    L.reset_tuning()


def test_wgrad_interleaved_loop_equals_lockstep_bit_for_bit():
    """wgrad_ilv 1 vs 0 (default) on both LDS-DMA tiles (256 x 256 alone and grouped, 128 x 128 grouped), bias column included"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    dev = "cuda"
    cases = [(4, 50, 84, 256, 256, 3, 1, 1), (4, 50, 84, 1024, 256, 1, 1, 0), (2, 100, 168, 128, 128, 3, 1, 1), (2, 100, 168, 512, 128, 1, 1, 0), (2, 13, 21, 256, 256, 3, 1, 1)]
    outs = {}
    for ilv in (1, 0):
        L.reset_tuning(); L.set_tuning("wgrad_ilv", ilv)
        g2 = torch.Generator().manual_seed(9)
        probs = []
        for (N, H, W_, Cin, Cout, k, stride, pad) in cases:
            x = torch.randn(N, H, W_, Cin, generator=g2).to(dev, torch.bfloat16)
            g = (torch.randn(N, H, W_, Cout, generator=g2) * 0.1).to(dev, torch.bfloat16)
            probs.append((x, g, torch.zeros(Cout, k, k, Cin, device=dev), dict(KH=k, KW=k, stride=stride, pad=pad, db=torch.zeros(Cout, device=dev))))
        ops.conv_wgrad_group(probs)
        assert "64_group" in L.last_dispatch(), L.last_dispatch()
        single = torch.zeros(256, 3, 3, 256, device=dev)
        L.set_tuning("wgrad_big_min", 1); L.set_tuning("wgrad_big_slots", 8)
        ops.conv_wgrad(probs[0][0], probs[0][1], single, KH=3, KW=3, stride=1, pad=1)
        assert "big64" in L.last_dispatch(), L.last_dispatch()
        torch.cuda.synchronize()
        outs[ilv] = [(p[2], p[3]["db"]) for p in probs] + [(single, single)]
    for (a, ab), (b, bb) in zip(outs[1], outs[0]):
        assert torch.equal(a, b) and torch.equal(ab, bb)
        assert float(a.abs().max()) > 0
    L.reset_tuning()


@pytest.mark.parametrize("case,kind", [((4, 25, 42, 512, 512, 3, 1, 1), "f1"), ((2, 25, 42, 512, 512, 3, 1, 1), "d3"), ((1, 19, 23, 128, 256, 3, 1, 1), "f1x"), ((2, 25, 42, 64, 96, 3, 1, 1), "n")])
def test_halo_64x64_tile_forced(case, kind):
    """igemm_force 16 / igemm_halo_small: the 3x3 halo form on 64 x 64 tiles (four waves of 32 x 32), direct epilogue"""
    _check_direct(case, kind, "igemm<bf16,64,64,2,2,flat,halo,direct>", force=16)


def test_halo_small_knob_selects_the_64x64_tile():
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 25, 42, 512, generator=gen).to("cuda", torch.bfloat16)
    w = (torch.randn(512, 3, 3, 512, generator=gen) / 68.0).to("cuda", torch.bfloat16)
    a = ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,128,64,4,1,flat,halo,direct>", L.last_dispatch()
    L.set_tuning("igemm_halo_small", 320)
    b = ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,64,64,2,2,flat,halo,direct>", L.last_dispatch()
    torch.cuda.synchronize()
    assert torch.equal(a, b)                    # same K order per output element
    L.reset_tuning()


@pytest.mark.parametrize("case,kind", [((4, 50, 84, 256, 256, 3, 1, 1), "f1"), ((2, 50, 84, 256, 256, 3, 1, 1), "d3"), ((1, 19, 23, 128, 256, 3, 1, 1), "f1x"), ((2, 25, 42, 64, 96, 3, 1, 1), "n")])
def test_halo_96x64_three_wave_tile_forced(case, kind):
    """igemm_force 17 / igemm_halo96: the 3x3 halo form on 96 x 64 tiles (three waves of 32 x 64), direct epilogue"""
    _check_direct(case, kind, "igemm<bf16,96,64,3,1,flat,halo,direct>", force=17)


def test_halo96_knob_selects_the_three_wave_tile_bit_identically():
    from aldi_amd import _lib as L
    from aldi_amd import ops
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(4, 50, 84, 256, generator=gen).to("cuda", torch.bfloat16)
    w = (torch.randn(256, 3, 3, 256, generator=gen) / 48.0).to("cuda", torch.bfloat16)
    a = ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,128,64,4,1,flat,halo,direct>", L.last_dispatch()
    L.set_tuning("igemm_halo96", 1)
    b = ops.conv2d(x, w, pad=1, relu=True)
    assert L.last_dispatch() == "igemm<bf16,96,64,3,1,flat,halo,direct>", L.last_dispatch()
    torch.cuda.synchronize()
    assert torch.equal(a, b)                    # same K order per output element
    L.reset_tuning()
