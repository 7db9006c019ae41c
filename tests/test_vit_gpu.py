"""ViTDet trunk kernels (C ABI) vs plain PyTorch fp32 references of the same ops (floating-point kernels: tolerance stated
per test), and vs transformers' VitDet modules (the same algorithm as detectron2's vit.py that aldi/backbone.py drives)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("C", [100, 256, 768, 1028, 1536])     # 100 / 1028: rows that are not whole 16-B vectors (4-wide lane chunks in bf16 too)
def test_layernorm_fwd_bwd(dtype, tol, C):
    from aldi_amd import vit_ops as V
    torch.manual_seed(0)
    rows = 333
    x = (torch.randn(rows, C, device=DEV) * 2 + 0.5).to(dtype)
    gamma = torch.randn(C, device=DEV) * 0.5 + 1
    beta = torch.randn(C, device=DEV) * 0.1
    g = torch.randn(rows, C, device=DEV).to(dtype)
    res = torch.randn(rows, C, device=DEV).to(dtype)
    y, mean, rstd = V.layernorm_forward(x, gamma, beta, eps=1e-6)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-6)
    assert relerr(y, yr) < tol
    yr.backward(g.float())
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = V.layernorm_backward(g, x, gamma, mean, rstd, dgamma, dbeta, res=res)
    assert relerr(dx, xr.grad + res.float()) < tol
    assert relerr(dgamma, gr.grad) < max(tol, 1e-4) and relerr(dbeta, br.grad) < max(tol, 1e-4)


def _partition_map(N, gh, gw, ws):
    """window-order row map (detectron2 window_partition): out row -> source token or -1 (padding)"""
    ph, pw = (ws - gh % ws) % ws, (ws - gw % ws) % ws
    Hp, Wp = gh + ph, gw + pw
    idx = torch.full((N, Hp, Wp), -1, dtype=torch.int32)
    idx[:, :gh, :gw] = torch.arange(N * gh * gw, dtype=torch.int32).view(N, gh, gw)
    win = idx.view(N, Hp // ws, ws, Wp // ws, ws).permute(0, 1, 3, 2, 4).reshape(-1)
    return win, (Hp, Wp)


def test_layernorm_window_partition_map():
    from aldi_amd import vit_ops as V
    from transformers.models.vitdet.modeling_vitdet import window_partition
    torch.manual_seed(1)
    N, gh, gw, C, ws = 2, 9, 17, 256, 7
    x = torch.randn(N * gh * gw, C, device=DEV)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    win, (Hp, Wp) = _partition_map(N, gh, gw, ws)
    win = win.to(DEV)
    y, mean, rstd = V.layernorm_forward(x, gamma, beta, row_map=win)
    ref, _ = window_partition(F.layer_norm(x, (C,), gamma, beta, 1e-6).view(N, gh, gw, C), ws)
    assert y.shape[0] == ref.shape[0] * ws * ws
    assert relerr(y, ref.reshape(-1, C)) < 2e-5
    # backward through the same map: padded rows receive no gradient, every source row exactly one
    g = torch.randn_like(y)
    xr = x.clone().requires_grad_(True)
    yr, _ = window_partition(F.layer_norm(xr, (C,), gamma, beta, 1e-6).view(N, gh, gw, C), ws)
    yr.reshape(-1, C).backward(g)
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = V.layernorm_backward(g, x, gamma, mean, rstd, dgamma, dbeta, row_map=win)
    assert relerr(dx, xr.grad) < 2e-5
    # un-partition + residual through rows_add with the inverse map
    inv = torch.empty(N * gh * gw, dtype=torch.int32, device=DEV)
    valid = win >= 0
    inv[win[valid].long()] = torch.nonzero(valid).flatten().int()
    out = V.rows_add(x, y, rows=N * gh * gw, row_map=inv)
    assert relerr(out, x + F.layer_norm(x, (C,), gamma, beta, 1e-6)) < 2e-5


# ------------------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_gelu(dtype, tol):
    from aldi_amd import vit_ops as V
    torch.manual_seed(2)
    x = (torch.randn(1000, 64, device=DEV) * 2).to(dtype)
    g = torch.randn(1000, 64, device=DEV).to(dtype)
    xr = x.float().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(g.float())
    assert relerr(V.gelu(x), yr) < tol
    assert relerr(V.gelu_backward(x, g), xr.grad) < tol


def test_rows_add_scale():
    from aldi_amd import vit_ops as V
    torch.manual_seed(3)
    a, b = torch.randn(40, 64, device=DEV), torch.randn(40, 64, device=DEV)
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0], device=DEV)
    out = V.rows_add(a, b, rows=40, scale=scale, rows_per_sample=10)
    ref = a + b * scale.repeat_interleave(10)[:, None]
    assert torch.allclose(out, ref, atol=1e-6)
    out = V.rows_add(None, b.bfloat16(), rows=40)
    assert torch.equal(out, b.bfloat16())


def test_patchify_matches_conv():
    from aldi_amd import vit_ops as V
    torch.manual_seed(4)
    N, Hs, Ws, P = 2, 64, 96, 16
    img = torch.randint(0, 256, (N, 3, Hs, Ws), dtype=torch.uint8, device=DEV)
    hw = torch.tensor([[64, 96], [50, 70]], dtype=torch.int32, device=DEV)
    img[1, :, 50:, :] = 0
    img[1, :, :, 70:] = 0
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    rows = V.patchify(img, hw.flatten(), P, mean, std, torch.float32)
    w = torch.randn(32, 3, P, P, device=DEV)
    xn = (img.float() - torch.tensor(mean, device=DEV).view(1, 3, 1, 1)) / torch.tensor(std, device=DEV).view(1, 3, 1, 1)
    xn[1, :, 50:, :] = 0
    xn[1, :, :, 70:] = 0
    ref = F.conv2d(xn, w, stride=P).permute(0, 2, 3, 1).reshape(-1, 32)
    assert relerr(rows @ w.view(32, -1).t(), ref) < 1e-4


@pytest.mark.parametrize("L0,L1", [(127, 99), (127, 167), (27, 27), (5, 11)])
def test_linear_resize(L0, L1):
    from aldi_amd import vit_ops as V
    torch.manual_seed(5)
    t = torch.randn(L0, 64, device=DEV)
    tr = t.clone().requires_grad_(True)
    ref = F.interpolate(tr.reshape(1, L0, -1).permute(0, 2, 1), size=L1, mode="linear").reshape(-1, L1).permute(1, 0)
    out = V.linear_resize(t, L1)
    assert torch.allclose(out, ref, atol=5e-5)      # the tap weight is a difference of O(100) fp32 coordinates
    g = torch.randn(L1, 64, device=DEV)
    ref.backward(g)
    dt = torch.zeros_like(t)
    V.linear_resize_backward(g, dt)
    assert torch.allclose(dt, tr.grad, atol=2e-4)


@pytest.mark.parametrize("S0,gh,gw", [(14, 50, 84), (64, 50, 84), (8, 5, 7)])
def test_bicubic_resize(S0, gh, gw):
    from aldi_amd import vit_ops as V
    torch.manual_seed(6)
    C = 32
    p = torch.randn(S0, S0, C, device=DEV)
    pr = p.clone().requires_grad_(True)
    ref = F.interpolate(pr.permute(2, 0, 1)[None], size=(gh, gw), mode="bicubic", align_corners=False)[0].permute(1, 2, 0)
    out = V.bicubic_resize(p, gh, gw)
    assert torch.allclose(out, ref, atol=1e-4)
    g = torch.randn(gh, gw, C, device=DEV)
    ref.backward(g)
    dp = torch.zeros_like(p)
    V.bicubic_resize_backward(g, dp)
    assert torch.allclose(dp, pr.grad, atol=1e-3, rtol=1e-4)


def test_add_pos_sum_batch():
    from aldi_amd import vit_ops as V
    torch.manual_seed(7)
    N, Tk, C = 3, 35, 64
    x = torch.randn(N, Tk, C, device=DEV).bfloat16()
    pos = torch.randn(Tk, C, device=DEV)
    y = V.add_pos(x, pos, N)
    assert torch.equal(y, (x.float() + pos).bfloat16())
    s = V.sum_batch(x, N)
    assert torch.allclose(s.view(Tk, C), x.float().sum(0), atol=1e-5)


def test_adamw_matches_torch():
    from aldi_amd import vit_ops as V
    torch.manual_seed(8)
    n = 5000
    p0 = torch.randn(n, device=DEV)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pc = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 6):
        g = torch.randn(n, device=DEV)
        pr.grad = g.clone()
        opt.step()
        V.adamw_step(p, g, m, v, pc, lr=1e-3, weight_decay=0.1, step=step)
        assert torch.allclose(p, pr.detach(), atol=2e-6, rtol=1e-5), step
    assert torch.equal(pc, p.bfloat16())


# ------------------------------------------------------------------------------------------------ attention
def _attn_reference(qkv, rel_h, rel_w, nB, gh, gw, heads):
    """transformers VitDetAttention.forward minus the two linears, fp32"""
    from transformers.models.vitdet.modeling_vitdet import add_decomposed_relative_positions
    L = gh * gw
    t = qkv.reshape(nB, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t.reshape(3, nB * heads, L, 64).unbind(0)
    s = (q * 64 ** -0.5) @ k.transpose(-2, -1)
    if rel_h is not None:
        s = add_decomposed_relative_positions(s, q, rel_h, rel_w, (gh, gw), (gh, gw))
    p = s.softmax(dim=-1)
    o = (p @ v).view(nB, heads, gh, gw, 64).permute(0, 2, 3, 1, 4).reshape(nB * L, heads * 64)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("nB,gh,gw,heads,rel", [
    (6, 14, 14, 3, True),       # ViTDet window (L = 196: one block per window-head)
    (2, 10, 13, 2, True),       # small global grid, ragged tiles
    (1, 50, 84, 2, True),       # the cfg-4 global grid: 4200 tokens; tiled path (8x8 key blocks), Dq = 224 columns of Q'
    (2, 17, 23, 2, True),       # tiled path, ragged last row / column of key blocks, two images
    (1, 16, 24, 3, True),       # tiled path, grid an exact multiple of the 8x8 blocks
    (2, 33, 7, 1, True),        # tiled path, a single ragged column of blocks
    (2, 9, 20, 2, False),       # no relative positions
    (3, 4, 5, 1, True),         # L < 64
])
def test_attention_fwd_bwd(nB, gh, gw, heads, rel):
    from aldi_amd import vit_ops as V
    torch.manual_seed(nB * 100 + gh)
    L = gh * gw
    qkv = (torch.randn(nB * L, 3 * heads * 64, device=DEV) * 1.5).bfloat16()
    rel_h = (torch.randn(2 * gh - 1, 64, device=DEV) * 0.1) if rel else None
    rel_w = (torch.randn(2 * gw - 1, 64, device=DEV) * 0.1) if rel else None
    dO = torch.randn(nB * L, heads * 64, device=DEV).bfloat16()

    qr = qkv.float().requires_grad_(True)
    rh = rel_h.clone().requires_grad_(True) if rel else None
    rw = rel_w.clone().requires_grad_(True) if rel else None
    o_ref, lse_ref = _attn_reference(qr, rh, rw, nB, gh, gw, heads)
    o_ref.backward(dO.float())

    att = V.Attention(nB, gh, gw, heads, DEV, rel=rel)
    O, lse = att.forward(qkv, rel_h, rel_w)
    # bf16 operands (Q' incl. the bias columns, P) with fp32 accumulation: 2e-2 of the output range
    assert relerr(O, o_ref) < 2e-2
    assert (lse - lse_ref).abs().max().item() < 5e-2       # the bias columns of Q' are bf16: |bias| * 2^-9 on a logit
    drh = torch.zeros_like(rel_h) if rel else None
    drw = torch.zeros_like(rel_w) if rel else None
    dqkv = att.backward(qkv, rel_h, rel_w, O, lse, dO, drh, drw, prepared=True)
    third = heads * 64
    for name, sl in (("dq", slice(0, third)), ("dk", slice(third, 2 * third)), ("dv", slice(2 * third, 3 * third))):
        assert relerr(dqkv[:, sl], qr.grad[:, sl]) < 3e-2, name
    if rel:
        assert relerr(drh, rh.grad) < 3e-2
        assert relerr(drw, rw.grad) < 3e-2


# ------------------------------------------------------------------------------------------------ whole trunk
def _hf_model(cfg, drop_path=0.0):
    from transformers import VitDetConfig, VitDetModel
    hc = VitDetConfig(hidden_size=cfg.embed, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads, mlp_ratio=cfg.mlp_ratio,
                      hidden_act="gelu", dropout_prob=0.0, layer_norm_eps=cfg.ln_eps, image_size=cfg.rel_input * cfg.patch,
                      pretrain_image_size=cfg.pretrain_grid * cfg.patch, patch_size=cfg.patch, num_channels=3, qkv_bias=True,
                      drop_path_rate=drop_path, window_block_indices=[i for i in range(cfg.depth) if i not in cfg.global_blocks],
                      residual_block_indices=[], use_absolute_position_embeddings=True, use_relative_position_embeddings=True,
                      window_size=cfg.window)
    torch.manual_seed(11)
    m = VitDetModel(hc).to(DEV).float()
    with torch.no_grad():       # HF initialises rel_pos / pos_embed to zeros: give every parameter a value that matters
        for n, p_ in m.named_parameters():
            if "rel_pos" in n or "position_embeddings" in n:
                p_.copy_(torch.randn_like(p_) * 0.2)
            elif n.endswith("bias"):
                p_.copy_(torch.randn_like(p_) * 0.1)
            elif "norm" in n and n.endswith("weight"):
                p_.copy_(1 + 0.2 * torch.randn_like(p_))
            else:
                p_.copy_(torch.randn_like(p_) * 0.05)
            p_.copy_(p_.bfloat16().float())           # both sides compute with the same (bf16-representable) weights
    return m


def _hf_to_d2(m, cfg):
    sd, out = m.state_dict(), {}
    P = cfg.prefix
    out[P + "pos_embed"] = sd["embeddings.position_embeddings"]
    out[P + "patch_embed.proj.weight"] = sd["embeddings.projection.weight"]
    out[P + "patch_embed.proj.bias"] = sd["embeddings.projection.bias"]
    for i in range(cfg.depth):
        for a, b in (("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"), ("attn.rel_pos_h", "attention.rel_pos_h"),
                     ("attn.rel_pos_w", "attention.rel_pos_w"), ("attn.qkv.weight", "attention.qkv.weight"),
                     ("attn.qkv.bias", "attention.qkv.bias"), ("attn.proj.weight", "attention.proj.weight"),
                     ("attn.proj.bias", "attention.proj.bias"), ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
                     ("mlp.fc1.weight", "mlp.fc1.weight"), ("mlp.fc1.bias", "mlp.fc1.bias"), ("mlp.fc2.weight", "mlp.fc2.weight"),
                     ("mlp.fc2.bias", "mlp.fc2.bias")):
            out[f"{P}blocks.{i}.{a}"] = sd[f"encoder.layer.{i}.{b}"]
    return out


def test_vit_trunk_matches_transformers_vitdet():
    """forward tokens and every parameter gradient of a 4-block ViTDet trunk (2 windowed + 2 global blocks, ragged window
    padding, resized abs-pos and rel-pos tables) vs transformers' VitDetModel in fp32.  bf16 activations: 4e-2 of range."""
    from aldi_amd.vit import ViT, VitConfig, VitParams
    cfg = VitConfig(embed=128, depth=4, heads=2, patch=16, window=7, global_blocks=(1, 3), pretrain_grid=4, rel_input=10,
                    drop_path_rate=0.0)
    m = _hf_model(cfg)
    params = VitParams(cfg, DEV)
    params.load_state_dict(_hf_to_d2(m, cfg))
    vit = ViT(params)
    torch.manual_seed(12)
    N, Hs, Ws = 2, 96, 144                      # 6 x 9 tokens: padded to 7 x 14 for the windows
    img = torch.randint(0, 256, (N, 3, Hs, Ws), dtype=torch.uint8, device=DEV)
    sizes = [(96, 144), (80, 130)]
    img[1, :, 80:, :] = 0
    img[1, :, :, 130:] = 0
    mean = torch.tensor(cfg.pixel_mean, device=DEV).view(1, 3, 1, 1)
    std = torch.tensor(cfg.pixel_std, device=DEV).view(1, 3, 1, 1)
    xn = (img.float() - mean) / std
    xn[1, :, 80:, :] = 0
    xn[1, :, :, 130:] = 0
    ref = m(xn).last_hidden_state.permute(0, 2, 3, 1).reshape(-1, cfg.embed)     # [N*gh*gw, E]
    g = torch.randn_like(ref).bfloat16()
    ref.backward(g.float())

    params.zero_grad()
    ctx = vit.forward(img, sizes, save=True)
    assert relerr(ctx.out, ref) < 4e-2
    vit.backward(ctx, g)
    grads = _hf_to_d2(type("G", (), {"state_dict": lambda self: {n: p_.grad for n, p_ in m.named_parameters()}})(), cfg)
    worst = {}
    for name in params.spec:
        mine = params._view(params.grad, name)
        r = grads[name]
        if name.endswith("pos_embed"):
            mine, r = mine[:, 1:], r[:, 1:]          # the cls slot gets no gradient
        worst[name] = relerr(mine, r)
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, bad


def test_vit_drop_path_and_adamw_step():
    """stochastic depth with host-drawn masks: a dropped branch leaves its block parameters without gradient, kept branches are
    scaled by 1/keep; one AdamW step moves only what has gradient or decay."""
    from aldi_amd.vit import ViT, VitConfig, VitParams
    cfg = VitConfig(embed=128, depth=2, heads=2, patch=16, window=7, global_blocks=(1,), pretrain_grid=4, rel_input=10, drop_path_rate=0.5)
    params = VitParams(cfg, DEV)
    params.init_random(3)
    vit = ViT(params)
    img = torch.randint(0, 256, (2, 3, 64, 64), dtype=torch.uint8, device=DEV)
    ds = torch.ones(2, 2, 2)
    ds[1, 0, :] = 0.0                # attention branch of block 1 dropped for both samples
    ds[1, 1, :] = 2.0
    params.zero_grad()
    ctx = vit.forward(img, [(64, 64), (64, 64)], drop_scales=ds)
    vit.backward(ctx, torch.randn_like(ctx.out))
    assert params.g("blocks.1.attn.qkv.weight").abs().max().item() == 0.0
    assert params.g("blocks.1.mlp.fc1.weight").abs().max().item() > 0.0
    assert params.g("blocks.0.attn.qkv.weight").abs().max().item() > 0.0
    before = params.master.clone()
    params.adamw_step(1e-3)
    moved = (params.master - before).abs()
    assert moved[params.off[cfg.prefix + "blocks.0.attn.qkv.weight"]].item() > 0
    assert torch.equal(params.compute.float(), params.master.bfloat16().float())


# ------------------------------------------------------------------------------------------------ SimpleFeaturePyramid
def test_maxpool2_first_max_wins():
    from aldi_amd import vit_ops as V
    torch.manual_seed(20)
    x = torch.randint(-3, 4, (2, 6, 8, 16), device=DEV).float().bfloat16()      # many ties
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    g = torch.randn_like(yr)
    yr.backward(g)
    y, idx = V.maxpool2(x)
    assert torch.equal(y.float(), yr.permute(0, 2, 3, 1))
    dx = V.maxpool2_backward(g.permute(0, 2, 3, 1).contiguous().bfloat16(), idx, 6, 8)
    assert torch.equal(dx.float(), (xr.grad.permute(0, 2, 3, 1) != 0) * dx.float())       # same taps
    assert torch.allclose(dx.float(), xr.grad.permute(0, 2, 3, 1).bfloat16().float())


def _chan_ln(x, w, b, eps=1e-6):       # detectron2 layers/batch_norm.py LayerNorm (channels-first)
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def _sfp_reference(x, sd):
    q = "backbone."

    def conv_ln(t, name, k):
        return _chan_ln(F.conv2d(t, sd[name + ".weight"], padding=k // 2), sd[name + ".norm.weight"], sd[name + ".norm.bias"])
    t = F.conv_transpose2d(x, sd[q + "simfp_2.0.weight"], sd[q + "simfp_2.0.bias"], stride=2)
    t = F.gelu(_chan_ln(t, sd[q + "simfp_2.1.weight"], sd[q + "simfp_2.1.bias"]))
    t = F.conv_transpose2d(t, sd[q + "simfp_2.3.weight"], sd[q + "simfp_2.3.bias"], stride=2)
    p2 = conv_ln(conv_ln(t, q + "simfp_2.4", 1), q + "simfp_2.5", 3)
    t = F.conv_transpose2d(x, sd[q + "simfp_3.0.weight"], sd[q + "simfp_3.0.bias"], stride=2)
    p3 = conv_ln(conv_ln(t, q + "simfp_3.1", 1), q + "simfp_3.2", 3)
    p4 = conv_ln(conv_ln(x, q + "simfp_4.0", 1), q + "simfp_4.1", 3)
    p5 = conv_ln(conv_ln(F.max_pool2d(x, 2, 2), q + "simfp_5.1", 1), q + "simfp_5.2", 3)
    p6 = F.max_pool2d(p5, kernel_size=1, stride=2)
    return [p2, p3, p4, p5, p6]


def test_simple_feature_pyramid_fwd_bwd():
    """SimpleFeaturePyramid (two deconv stages, channel LN, GELU, 1x1 + 3x3 conv+LN, max pool, p6) vs torch fp32 with the same
    bf16-representable weights; bf16 activations: 4e-2 of range on maps, 6e-2 on gradients."""
    from aldi_amd.vit import SimpleFeaturePyramid, VitConfig, VitParams
    cfg = VitConfig(embed=256, depth=0, heads=4, global_blocks=(), pretrain_grid=2, sfp=True, fpn_channels=64)
    params = VitParams(cfg, DEV)
    torch.manual_seed(21)
    sd = {}
    for name, (shape, _) in params.spec.items():
        if name.endswith("norm.weight") or name.endswith("simfp_2.1.weight"):
            t = 1 + 0.2 * torch.randn(shape)
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape)
        else:
            t = torch.randn(shape) * (2.0 / max(shape[-1] * shape[-2] * shape[1], 1) if len(shape) == 4 else 0.05) ** 0.5
        sd[name] = t.bfloat16().float()
    params.load_state_dict(sd)
    rt = params.state_dict()
    assert all(torch.equal(rt[k], sd[k]) for k in sd)            # NHWC storage round-trips to the detectron2 layouts
    sfp = SimpleFeaturePyramid(params)
    N, gh, gw = 2, 6, 10
    x = torch.randn(N, gh, gw, cfg.embed, device=DEV).bfloat16()
    sdd = {k: v.to(DEV).requires_grad_(True) for k, v in sd.items() if "simfp" in k}
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = _sfp_reference(xr, sdd)
    gs = [torch.randn_like(r).bfloat16() for r in ref]
    torch.autograd.backward(ref, [g.float() for g in gs])

    params.zero_grad()
    c = sfp.forward(x)
    for l, (mine, r) in enumerate(zip(c.P, ref)):
        assert relerr(mine, r.permute(0, 2, 3, 1)) < 4e-2, l
    gP = [g.permute(0, 2, 3, 1).contiguous() for g in gs]
    g5 = gP[3].float()
    g5[:, ::2, ::2] += gP[4].float()                                # LastLevelMaxPool: p6 = p5[:, ::2, ::2]
    dx = sfp.backward(c, [gP[0], gP[1], gP[2], g5.bfloat16()])
    assert relerr(dx.view(N, gh, gw, -1), xr.grad.permute(0, 2, 3, 1)) < 6e-2
    bad = {}
    for name, r in sdd.items():
        mine = params.state_dict_like(params.grad)[name].to(DEV)
        e = relerr(mine, r.grad)
        if e > 6e-2:
            bad[name] = e
    assert not bad, bad


def test_vit_geometry_cache_is_bounded():
    """multi-scale inputs: per-geometry attention workspaces are kept for the most recent few shapes only, and a pass whose geometry
    was evicted and rebuilt still produces the same tokens"""
    from aldi_amd.vit import ViT, VitConfig, VitParams
    cfg = VitConfig(embed=128, depth=2, heads=2, window=7, global_blocks=(1,), pretrain_grid=4, rel_input=10, drop_path_rate=0.0)
    params = VitParams(cfg, DEV)
    params.init_random(1)
    vit = ViT(params)
    torch.manual_seed(0)
    first = torch.randint(0, 256, (1, 3, 64, 96), dtype=torch.uint8, device=DEV)
    ref = vit.forward(first, [(64, 96)], save=False).out.clone()
    for h, w in ((64, 64), (96, 64), (96, 96), (128, 64), (64, 128), (128, 96)):
        vit.forward(torch.randint(0, 256, (1, 3, h, w), dtype=torch.uint8, device=DEV), [(h, w)], save=True)
    assert len(vit._geom) <= ViT.MAX_GEOMETRIES and (1, 4, 6) not in vit._geom
    again = vit.forward(first, [(64, 96)], save=False).out
    assert torch.equal(again, ref)
