"""The Deformable-DETR detector after its backbone on the HIP library (aldi_amd/detr/model.py: forward pass) against
oracle/deformable_detr.py (pinned to transformers' implementation in tests/test_oracle_detr_cpu.py), at the reference configuration's
widths (d_model 256, 8 heads x 32, 4 levels x 4 points, FFN 1024; configs/Base-DETR.yaml:13-26) on small maps, one image padded."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(gen, d=256, L=4, enc=2, dec=2, M=8, P=4, ffn=1024, Nq=100, K=20, chans=(512, 1024, 2048)):
    def rn(*s, std=0.05):
        return torch.randn(*s, generator=gen) * std
    p = {}
    for l in range(L):
        cin = chans[l] if l < len(chans) else (chans[-1] if l == len(chans) else d)
        k = 1 if l < len(chans) else 3
        p[f"input_proj.{l}.0.weight"] = rn(d, cin, k, k, std=(cin * k * k) ** -0.5)
        p[f"input_proj.{l}.0.bias"] = rn(d)
        p[f"input_proj.{l}.1.weight"] = 1 + rn(d, std=0.1)
        p[f"input_proj.{l}.1.bias"] = rn(d, std=0.1)
    p["transformer.level_embed"] = rn(L, d, std=0.5)
    def attn(pre):
        p[pre + ".sampling_offsets.weight"] = rn(M * L * P * 2, d, std=0.02)
        p[pre + ".sampling_offsets.bias"] = rn(M * L * P * 2, std=1.5)                 # offsets of a few pixels
        p[pre + ".attention_weights.weight"] = rn(M * L * P, d)
        p[pre + ".attention_weights.bias"] = rn(M * L * P, std=0.5)
        for n in ("value_proj", "output_proj"):
            p[pre + f".{n}.weight"] = rn(d, d, std=d ** -0.5)
            p[pre + f".{n}.bias"] = rn(d)
    def ffn_norms(pre, norms):
        p[pre + ".linear1.weight"], p[pre + ".linear1.bias"] = rn(ffn, d, std=d ** -0.5), rn(ffn)
        p[pre + ".linear2.weight"], p[pre + ".linear2.bias"] = rn(d, ffn, std=ffn ** -0.5), rn(d)
        for n in norms:
            p[pre + f".{n}.weight"], p[pre + f".{n}.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    for i in range(enc):
        attn(f"transformer.encoder.layers.{i}.self_attn")
        ffn_norms(f"transformer.encoder.layers.{i}", ("norm1", "norm2"))
    for i in range(dec):
        pre = f"transformer.decoder.layers.{i}"
        attn(pre + ".cross_attn")
        p[pre + ".self_attn.in_proj_weight"], p[pre + ".self_attn.in_proj_bias"] = rn(3 * d, d, std=d ** -0.5), rn(3 * d)
        p[pre + ".self_attn.out_proj.weight"], p[pre + ".self_attn.out_proj.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        ffn_norms(pre, ("norm1", "norm2", "norm3"))
    p["transformer.reference_points.weight"], p["transformer.reference_points.bias"] = rn(2, d, std=0.3), rn(2, std=0.3)
    p["query_embed.weight"] = rn(Nq, 2 * d, std=1.0)
    p["class_embed.weight"], p["class_embed.bias"] = rn(K, d, std=d ** -0.5), rn(K)
    for j, (o, i) in enumerate(((d, d), (d, d), (4, d))):
        p[f"bbox_embed.layers.{j}.weight"], p[f"bbox_embed.layers.{j}.bias"] = rn(o, i, std=i ** -0.5), rn(o, std=0.3)
    return p


def test_forward_matches_the_oracle():
    from aldi_amd.detr.model import DeformableTransformer
    from oracle import deformable_detr as D
    gen = torch.Generator().manual_seed(0)
    cfg = dict(d_model=256, num_levels=4, enc_layers=2, dec_layers=2, n_heads=8, enc_points=4, dec_points=4)
    p = _params(gen)
    B, H, W = 2, 192, 256
    feats = [torch.randn(B, c, H // s, W // s, generator=gen) for c, s in ((512, 8), (1024, 16), (2048, 32))]
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, 160:, :] = True
    mask[1, :, 200:] = True
    ref_logits, ref_boxes = D.forward(p, feats, mask, **cfg)
    model = DeformableTransformer(p, **cfg)
    logits, boxes = model.forward([f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats], mask)
    torch.cuda.synchronize()
    logits, boxes = logits.cpu(), boxes.cpu()
    assert logits.shape == ref_logits.shape and boxes.shape == ref_boxes.shape
    e_l = (logits - ref_logits).abs().max().item() / max(1.0, ref_logits.abs().max().item())
    e_b = (boxes - ref_boxes).abs().max().item()
    assert e_l <= 2e-3 and e_b <= 1e-3, (e_l, e_b)
    # the padded image really exercised the masks, and the outputs are not degenerate
    assert (ref_logits[:, 0] - ref_logits[:, 1]).abs().max().item() > 1e-2 and 0.05 < float(ref_boxes.std())
    # a second call reuses the cached tables and gives the same bits
    l2, b2 = model.forward([f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats], mask)
    assert torch.equal(l2.cpu(), logits) and torch.equal(b2.cpu(), boxes)


def test_group_norm_and_small_attention_against_torch():
    from aldi_amd import _lib as L
    from aldi_amd.detr.model import group_norm
    from aldi_amd.ops import _p, stream_ptr
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 37, 53, 256, generator=g) * 3 + 1
    gamma, beta = torch.randn(256, generator=g), torch.randn(256, generator=g)
    y = group_norm(x.cuda(), gamma.cuda(), beta.cuda()).cpu()
    ref = torch.nn.functional.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5).permute(0, 2, 3, 1)
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    for (B, Q, H, Dh) in ((2, 300, 8, 32), (1, 77, 4, 16), (3, 64, 2, 64)):
        q, k, v = (torch.randn(B, Q, H * Dh, generator=g) for _ in range(3))
        out = torch.empty(B, Q, H * Dh, device="cuda")
        lse = torch.empty(B, H, Q, device="cuda")
        qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
        L.call("aldi_mha_small_forward", _p(qc), _p(kc), _p(vc), _p(out), _p(lse), B, Q, H, Dh, H * Dh, H * Dh, H * Dh, Dh ** -0.5, 0.0, 0, stream_ptr())
        qh, kh, vh = (t.view(B, Q, H, Dh).transpose(1, 2).double() for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) * Dh ** -0.5
        ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Q, H * Dh)
        assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        assert (lse.cpu().double() - torch.logsumexp(s, -1)).abs().max().item() <= 1e-5 * max(1.0, s.abs().max().item())


def test_backward_matches_the_oracles_autograd():
    """d(loss)/d(every parameter) and d(loss)/d(backbone maps) of the taped backward == torch autograd through the oracle, for a random
    linear functional of all decoder layers' logits and boxes (one image padded)"""
    from aldi_amd.detr.model import DeformableTransformer
    from oracle import deformable_detr as D
    gen = torch.Generator().manual_seed(3)
    cfg = dict(d_model=256, num_levels=4, enc_layers=2, dec_layers=2, n_heads=8, enc_points=4, dec_points=4)
    p = _params(gen, Nq=60, K=12)
    B, H, W = 2, 128, 160
    feats = [torch.randn(B, c, H // s, W // s, generator=gen) for c, s in ((512, 8), (1024, 16), (2048, 32))]
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, 100:, :] = True
    mask[1, :, 130:] = True
    pr = {k: v.clone().double().requires_grad_(True) for k, v in p.items()}
    fr = [f.clone().double().requires_grad_(True) for f in feats]
    lo, bo = D.forward(pr, fr, mask, **cfg)
    R1, R2 = torch.randn(lo.shape, generator=gen), torch.randn(bo.shape, generator=gen)
    (lo * R1.double()).sum().add((bo * R2.double()).sum()).backward()
    model = DeformableTransformer(p, **cfg)
    model.P.zero_grad()
    f_dev = [f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats]
    logits, boxes = model.forward(f_dev, mask, record=True, feats_need_grad=True)
    gf = model.backward(R1.cuda(), R2.cuda())
    torch.cuda.synchronize()
    assert (logits.cpu().double() - lo.detach()).abs().max().item() <= 2e-3 * max(1.0, lo.abs().max().item())
    got = model.P.state_dict(model.P.grad)
    worst = []
    for k, v in pr.items():
        ref = v.grad
        e = (got[k].cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        worst.append((e, k))
        assert ref.abs().max().item() > 0, k
    worst.sort(reverse=True)
    assert worst[0][0] <= 5e-3, worst[:8]
    for g, f in zip(gf, fr):
        ref = f.grad.permute(0, 2, 3, 1)
        assert (g.cpu().double() - ref).abs().max().item() <= 5e-3 * ref.abs().max().item()


@pytest.mark.parametrize("offset_px,valid,cap_overflow", [(2.0, 1.0, False), (9.0, 1.0, False), (3.0, 0.8, False), (0.4, 1.0, True)])
def test_self_attention_value_gradient_gathered(offset_px, valid, cap_overflow):
    """aldi_ms_deform_attn_backward_self (queries = the pyramid's positions: a workgroup per 8 x 8 tile of value pixels gathers the samples
    that reach it, the far ones go through the atomic scatter) == the general scatter kernel -- offsets inside the 5-pixel neighbourhood,
    well outside it, reference points of a padded image (valid ratio 0.8: the queries' samples drift away from their pixel's position),
    and samples piled onto few pixels (a tile's list overflows: its own atomic path); ragged level shapes (tiles cut by the border)"""
    import numpy as np
    from aldi_amd import _lib as L
    from aldi_amd.ops import _p, stream_ptr
    g = torch.Generator().manual_seed(21)
    shapes = [(37, 53), (19, 27), (10, 14), (5, 7)]
    N, M, D, Lv, P = 2, 8, 32, 4, 4
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, dtype=torch.int32)
    ls = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1]), dtype=torch.int32)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in shapes]) * valid
    if cap_overflow:
        ref = ref * 0 + 0.31                     # every query samples around one point: > 8192 (query, corner) pairs on a few pixels
    off = torch.randn(N, S, M, Lv, P, 2, generator=g) * offset_px / torch.tensor([[w, h] for h, w in shapes]).view(1, 1, 1, Lv, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + off).contiguous()
    aw = torch.softmax(torch.randn(N, S, M, Lv * P, generator=g), -1).view(N, S, M, Lv, P).contiguous()
    value, gout = torch.randn(N, S, M, D, generator=g), torch.randn(N, S, M * D, generator=g)
    dev = [t.cuda() for t in (value, sh, ls, loc, aw, gout)]
    host = np.ascontiguousarray(sh.numpy())
    outs = []
    # None: the general scatter; then the gather with its default level set (the three finest: the coarsest keeps the atomics), all four
    # levels (1 x 1 tiles with the list split over 64 threads on the 5 x 7 map), the finest only, and switched off (the general form)
    need = int(L.lib.aldi_ms_deform_attn_backward_self_workspace(host.ctypes.data, N, S, M, Lv, P))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    # None: the general scatter; "bin": the lists built by one pass over the samples (workspace), also with tiny lists (capacity overflow ->
    # atomics for the surplus); then the walk form (no workspace) with its default level set (the three finest: the coarsest keeps the
    # atomics), all four levels (1 x 1 tiles, the list split over 64 threads on the 5 x 7 map), the finest only, and switched off
    for mask in (None, "bin", "bin-small", -1, 15, 1, 0):
        gv, gl, ga = torch.full_like(dev[0], 7.0), torch.empty_like(dev[3]), torch.empty_like(dev[4])
        L.reset_tuning()
        if isinstance(mask, str):
            if mask == "bin-small":
                L.set_tuning("msda_bin_list", 8)
                need2 = int(L.lib.aldi_ms_deform_attn_backward_self_workspace(host.ctypes.data, N, S, M, Lv, P))
                assert need2 != need
                ws2 = torch.empty(need2, dtype=torch.uint8, device="cuda")
            else:
                ws2 = ws
            L.call("aldi_ms_deform_attn_backward_self", _p(dev[0]), _p(dev[1]), _p(dev[2]), host.ctypes.data, _p(dev[3]), _p(dev[4]), _p(dev[5]),
                   _p(gv), _p(gl), _p(ga), _p(ws2), ws2.numel(), N, S, M, D, Lv, P, stream_ptr())
            assert L.last_dispatch() == "msda_bwd_value_binned", L.last_dispatch()
            with pytest.raises(Exception):         # a workspace that is too small is refused
                L.call("aldi_ms_deform_attn_backward_self", _p(dev[0]), _p(dev[1]), _p(dev[2]), host.ctypes.data, _p(dev[3]), _p(dev[4]), _p(dev[5]),
                       _p(gv.clone()), _p(gl), _p(ga), _p(ws2), 1024, N, S, M, D, Lv, P, stream_ptr())
        elif mask is not None:
            if mask >= 0:
                L.set_tuning("msda_gather", mask)
            L.call("aldi_ms_deform_attn_backward_self", _p(dev[0]), _p(dev[1]), _p(dev[2]), host.ctypes.data, _p(dev[3]), _p(dev[4]), _p(dev[5]),
                   _p(gv), _p(gl), _p(ga), None, 0, N, S, M, D, Lv, P, stream_ptr())
            assert (L.last_dispatch() == "msda_bwd_value_gather") == (mask != 0), L.last_dispatch()
        else:
            L.call("aldi_ms_deform_attn_backward", _p(dev[0]), _p(dev[1]), _p(dev[2]), _p(dev[3]), _p(dev[4]), _p(dev[5]), _p(gv), _p(gl), _p(ga), N, S, M, D, S, Lv, P, stream_ptr())
        torch.cuda.synchronize()
        outs.append((gv.cpu(), gl.cpu(), ga.cpu()))
    L.reset_tuning()
    gv0, gl0, ga0 = outs[0]
    assert gv0.abs().max().item() > 1.0
    for gv1, gl1, ga1 in outs[1:]:
        assert torch.equal(gl0, gl1) and torch.equal(ga0, ga1)
        assert (gv0 - gv1).abs().max().item() <= 2e-5 * gv0.abs().max().item(), (gv0 - gv1).abs().max().item()
    with pytest.raises(Exception):             # level shapes that do not add up to S
        L.call("aldi_ms_deform_attn_backward_self", _p(dev[0]), _p(dev[1]), _p(dev[2]), host.ctypes.data, _p(dev[3]), _p(dev[4]), _p(dev[5]),
               _p(gv), _p(gl), _p(ga), None, 0, N, S - 1, M, D, Lv, P, stream_ptr())


def _keep_mask(seed, shape, p):
    """the keep/(1-p) factors aldi_dropout_add applies for (seed, element index)"""
    from aldi_amd import _lib as L
    from aldi_amd.ops import _p, stream_ptr
    ones = torch.ones(shape, device="cuda")
    out = torch.empty_like(ones)
    L.call("aldi_dropout_add", _p(ones), None, _p(out), ones.numel(), p, seed, stream_ptr())
    return out


def test_dropout_add_keep_rate_scale_and_seeds():
    from aldi_amd import _lib as L
    from aldi_amd.ops import _p, stream_ptr
    g = torch.Generator().manual_seed(2)
    x, r = torch.randn(1 << 20, generator=g).cuda(), torch.randn(1 << 20, generator=g).cuda()
    for p in (0.1, 0.5):
        m = _keep_mask(77, x.shape, p)
        kept = m != 0
        assert abs(kept.float().mean().item() - (1 - p)) < 3e-3                                 # 1M draws: sigma < 5e-4
        assert torch.allclose(m[kept], torch.full_like(m[kept], 1 / (1 - p)), rtol=1e-6)
        out = torch.empty_like(x)
        L.call("aldi_dropout_add", _p(x), _p(r), _p(out), x.numel(), p, 77, stream_ptr())
        assert torch.allclose(out, r + x * m, rtol=1e-6, atol=1e-6)
        assert torch.equal(m, _keep_mask(77, x.shape, p))                                       # stateless: the same seed, the same mask
        m2 = _keep_mask(78, x.shape, p)
        agree = ((m != 0) == (m2 != 0)).float().mean().item()
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 5e-3                                       # a different seed is an independent mask
        # no structure along the index: neighbours are independent
        k = kept.float()
        assert abs(((k[1:] * k[:-1]).mean() - (1 - p) ** 2).item()) < 3e-3
    out = torch.empty_like(x)
    L.call("aldi_dropout_add", _p(x), None, _p(out), x.numel(), 0.0, 5, stream_ptr())
    assert torch.equal(out, x)
    with pytest.raises(Exception):
        L.call("aldi_dropout_add", _p(x), None, _p(out), x.numel(), 1.0, 5, stream_ptr())


def test_dropout_forward_backward_equals_the_oracle_with_the_same_masks():
    """TRANSFORMER.DROPOUT 0.1 (the reference's shipped value, configs/Base-DETR.yaml): the recorded forward draws its masks from
    (seed, element); the oracle run with exactly those masks at the authors' sites (dropout1..4 of every layer and the decoder self
    attention's probabilities) gives the same outputs and the same gradients; an unrecorded forward (the teacher) has no dropout"""
    from aldi_amd.detr.model import DeformableTransformer
    from oracle import deformable_detr as D
    gen = torch.Generator().manual_seed(13)
    cfg = dict(d_model=256, num_levels=4, enc_layers=2, dec_layers=2, n_heads=8, enc_points=4, dec_points=4)
    p = _params(gen, Nq=60, K=12)
    B, H, W = 2, 128, 160
    feats = [torch.randn(B, c, H // s, W // s, generator=gen) for c, s in ((512, 8), (1024, 16), (2048, 32))]
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    mask[1, 100:, :] = True
    model = DeformableTransformer(p, dropout=0.1, seed=4, **cfg)
    model.P.zero_grad()
    f_dev = [f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats]
    logits, boxes = model.forward(f_dev, mask, record=True, feats_need_grad=True)
    sites = dict(model.drop_sites)
    want = {f"transformer.encoder.layers.{i}.dropout{j}" for i in range(2) for j in (1, 2, 3)} | \
           {f"transformer.decoder.layers.{i}.dropout{j}" for i in range(2) for j in (1, 2, 3, 4)} | \
           {f"transformer.decoder.layers.{i}.self_attn.attn" for i in range(2)}
    assert set(sites) == want and len({sd for sd, _ in sites.values()}) == len(sites)
    R1, R2 = torch.randn(logits.shape, generator=gen), torch.randn(boxes.shape, generator=gen)
    gf = model.backward(R1.cuda(), R2.cuda())
    torch.cuda.synchronize()
    masks = {k: _keep_mask(sd, shape, 0.1).cpu().double() for k, (sd, shape) in sites.items()}
    used = set()

    def drop(name, t):
        used.add(name)
        return t * masks[name].view(t.shape)
    pr = {k: v.clone().double().requires_grad_(True) for k, v in p.items()}
    fr = [f.clone().double().requires_grad_(True) for f in feats]
    lo, bo = D.forward(pr, fr, mask, drop=drop, **cfg)
    assert used == want
    (lo * R1.double()).sum().add((bo * R2.double()).sum()).backward()
    assert (logits.cpu().double() - lo.detach()).abs().max().item() <= 2e-3 * max(1.0, lo.abs().max().item())
    assert (boxes.cpu().double() - bo.detach()).abs().max().item() <= 1e-3
    got = model.P.state_dict(model.P.grad)
    worst = sorted(((got[k].cpu().double() - v.grad).abs().max().item() / max(v.grad.abs().max().item(), 1e-6), k) for k, v in pr.items())[::-1]
    assert worst[0][0] <= 5e-3, worst[:8]
    for g, f in zip(gf, fr):
        ref = f.grad.permute(0, 2, 3, 1)
        assert (g.cpu().double() - ref).abs().max().item() <= 5e-3 * ref.abs().max().item()
    # dropout really acted, the next recorded forward draws other masks, and evaluation has none
    l_eval, _ = model.forward(f_dev, mask)
    lo0, _ = D.forward(p, feats, mask, **cfg)
    assert not model.drop_sites and (l_eval.cpu() - lo0).abs().max().item() <= 2e-3 * max(1.0, lo0.abs().max().item())
    assert (logits.cpu() - lo0).abs().max().item() > 0.05
    l2, _ = model.forward(f_dev, mask, record=True)
    assert (l2 - logits).abs().max().item() > 0.05


def test_set_criterion_matches_the_oracle():
    """matching, the weighted loss entries of every decoder layer and d(total)/d(logits, boxes) == the oracle's criterion + autograd;
    one image without targets"""
    from aldi_amd.detr.criterion import SetCriterion
    from oracle import deformable_detr as D
    g = torch.Generator().manual_seed(9)
    Ld, B, Nq, K = 3, 3, 50, 9
    logits = torch.randn(Ld, B, Nq, K, generator=g) * 2
    cxcy = torch.rand(Ld, B, Nq, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(Ld, B, Nq, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1)
    targets = []
    for n in (4, 0, 7):
        targets.append({"labels": torch.randint(0, K, (n,), generator=g), "boxes": torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.3 + 0.05], -1)})
    lr, br = logits.clone().double().requires_grad_(True), boxes.clone().double().requires_grad_(True)
    tr = [{"labels": t["labels"], "boxes": t["boxes"].double()} for t in targets]
    ref, total = D.criterion(lr, br, tr, weights=(2.0, 5.0, 2.0))
    total.backward()
    crit = SetCriterion()
    out, gl, gb = crit(logits.cuda(), boxes.cuda(), targets)
    torch.cuda.synchronize()
    w = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0}
    assert list(out.keys()) == ["loss_ce_0", "loss_bbox_0", "loss_giou_0", "loss_ce_1", "loss_bbox_1", "loss_giou_1", "loss_ce", "loss_bbox", "loss_giou"]
    for k, v in out.items():
        base = k.split("_")[0] + "_" + k.split("_")[1]
        assert abs(float(v) - w[base] * float(ref[k])) <= 1e-4 * max(1.0, abs(w[base] * float(ref[k]))), (k, float(v), float(ref[k]))
    assert abs(sum(float(v) for v in out.values()) - float(total)) <= 1e-4 * float(total)
    assert (gl.cpu().double() - lr.grad).abs().max().item() <= 1e-4 * lr.grad.abs().max().item()
    assert (gb.cpu().double() - br.grad).abs().max().item() <= 1e-4 * br.grad.abs().max().item()
    # the assignment: one query per target, the oracle's pairs
    m = crit.last_match.cpu().view(Ld, B, Nq)
    for l in range(Ld):
        idx = D.hungarian_match(logits[l], boxes[l], targets)
        for b, (qi, gi) in enumerate(idx):
            got = {(int(q), int(m[l, b, q])) for q in range(Nq) if m[l, b, q] >= 0}
            assert got == set(zip(qi.tolist(), gi.tolist())), (l, b)


def _detr_cfg(**over):
    import os
    from aldi_amd.config import add_aldi_config, get_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(root, "configs", "Base-DETR.yaml"))
    cfg.merge_from_list(["MODEL.DEFORMABLE_DETR.NUM_CLASSES", 8, "MODEL.DEFORMABLE_DETR.TRANSFORMER.NUM_QUERIES", 40, "MODEL.DEFORMABLE_DETR.TRANSFORMER.ENC_LAYERS", 2,
                         "MODEL.DEFORMABLE_DETR.TRANSFORMER.DEC_LAYERS", 2, "SEED", 3])
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg


def _detr_batch(gen, sizes=((160, 224), (128, 192)), counts=(3, 2), K=8):
    from aldi_amd.structures import Boxes, Instances
    data = []
    for (h, w), n in zip(sizes, counts):
        inst = Instances((h, w))
        x0, y0 = torch.rand(n, generator=gen) * w * 0.5, torch.rand(n, generator=gen) * h * 0.5
        bw, bh = 20 + torch.rand(n, generator=gen) * w * 0.4, 20 + torch.rand(n, generator=gen) * h * 0.4
        inst.gt_boxes = Boxes(torch.stack([x0, y0, (x0 + bw).clamp(max=w), (y0 + bh).clamp(max=h)], -1))
        inst.gt_classes = torch.randint(0, K, (n,), generator=gen)
        data.append({"image": torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8), "instances": inst, "height": h, "width": w})
    return data


def test_detector_training_forward_backward_vs_oracle():
    """META_ARCHITECTURE DeformableDETR built from configs/Base-DETR.yaml (fewer layers / queries): the weighted loss dict of model(batch)
    and, after `sum(losses).backward()`, gradients of the transformer AND of the R50 trunk == the oracle (its R50 + oracle/deformable_detr.py
    + criterion, torch autograd); one image smaller than the other (padding mask)"""
    from aldi_amd.model import build_aldi
    from oracle import d2_rcnn as d2
    from oracle import deformable_detr as D
    cfg = _detr_cfg(**{"MODEL.DEFORMABLE_DETR.TRANSFORMER.DROPOUT": 0.0})       # dropout against the oracle: the transformer-level test above
    model = build_aldi(cfg)
    gen = torch.Generator().manual_seed(5)
    data = _detr_batch(gen)
    # the authors' initial sampling offsets are whole pixels: in the encoder every same-level sample then sits exactly on a pixel centre,
    # a kink of the bilinear interpolation where one-sided derivatives (and fp32 rounding of the location) decide the gradient -- jitter them
    for k in model.weights.tr.spec:
        if k.endswith("sampling_offsets.bias"):
            model.weights.tr.p(k).add_(torch.randn(model.weights.tr.p(k).shape, generator=gen).cuda() * 0.11)
    model.weights.zero_grad()
    ld = model(data)
    total = sum(ld.values())
    total.backward()
    torch.cuda.synchronize()
    # ---- the oracle on the same weights
    W = model.weights
    sd_b = {k: v.detach().cpu().double() for k, v in W.backbone.state_dict().items()}
    train_b = [k for k in sd_b if k.startswith(("backbone.bottom_up.res3", "backbone.bottom_up.res4", "backbone.bottom_up.res5")) and k.endswith(".weight") and ".norm." not in k]
    for k in train_b:
        sd_b[k].requires_grad_(True)
    p_t = {k: v.detach().cpu().double().requires_grad_(True) for k, v in W.tr.state_dict().items()}
    ocfg = d2.make_cfg(pixel_mean=list(cfg.MODEL.PIXEL_MEAN), pixel_std=list(cfg.MODEL.PIXEL_STD))
    x, sizes = d2.preprocess(ocfg, [d["image"] for d in data])
    stages = []
    d2.resnet_fpn(ocfg, sd_b, x.double(), stages)
    mask = torch.ones(len(data), x.shape[2], x.shape[3], dtype=torch.bool)
    for i, (h, w) in enumerate(sizes):
        mask[i, :h, :w] = False
    dims = dict(d_model=256, num_levels=4, enc_layers=2, dec_layers=2, n_heads=8, enc_points=4, dec_points=4)
    lo, bo = D.forward(p_t, stages[1:4], mask, **dims)
    targets = [{"labels": t["labels"], "boxes": t["boxes"].double()} for t in model._targets([d["instances"] for d in data], sizes)]
    ref, ref_total = D.criterion(lo, bo, targets, weights=(2.0, 5.0, 2.0))
    ref_total.backward()
    w = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0}
    assert set(ld.keys()) == set(ref.keys())
    for k, v in ld.items():
        base = "_".join(k.split("_")[:2])
        assert abs(float(v) - w[base] * float(ref[k])) <= 2e-3 * max(1.0, abs(w[base] * float(ref[k]))), (k, float(v), w[base] * float(ref[k]))
    got_t = W.tr.state_dict(W.tr.grad)
    worst = sorted(((got_t[k].cpu().double() - v.grad).abs().max().item() / max(v.grad.abs().max().item(), 1e-9), k) for k, v in p_t.items() if v.grad is not None and v.grad.abs().max() > 0)
    assert worst[-1][0] <= 2e-2, worst[-6:]
    for k in ("backbone.bottom_up.res5.2.conv3.weight", "backbone.bottom_up.res4.0.conv1.weight", "backbone.bottom_up.res3.0.conv1.weight", "backbone.bottom_up.res3.3.conv2.weight"):
        name = k[: -len(".weight")]
        g = W.backbone.gw(name).view(W.backbone.layout.t[name].wshape).permute(0, 3, 1, 2).cpu().double()
        r = sd_b[k].grad
        assert (g - r).abs().max().item() <= 2e-2 * r.abs().max().item(), (k, (g - r).abs().max().item(), r.abs().max().item())


def test_aldi_iterations_with_the_hard_distiller():
    """configs/cityscapes/ALDI-Best-DETR-Cityscapes.yaml (fewer layers / queries, small synthetic images) through ALDITrainer: EMA teacher,
    pseudo labels of the target images, student on source + pseudo-labelled target, AdamW with the parameter groups and the full-model clip --
    finite losses, weights move, the teacher is the EMA of the student with `query_embed` copied (aldi/ema.py:17,39-41)"""
    import os
    import random
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(root, "configs", "cityscapes", "ALDI-Best-DETR-Cityscapes.yaml"))
    cfg.merge_from_list(["MODEL.DEFORMABLE_DETR.TRANSFORMER.NUM_QUERIES", 40, "MODEL.DEFORMABLE_DETR.TRANSFORMER.ENC_LAYERS", 2,
                         "MODEL.DEFORMABLE_DETR.TRANSFORMER.DEC_LAYERS", 2, "SEED", 3, "SOLVER.IMS_PER_BATCH", 4, "SOLVER.IMS_PER_GPU", 2, "SOLVER.WARMUP_ITERS", 0,
                         "SYNTHETIC.HEIGHT", 160, "SYNTHETIC.WIDTH", 224, "EMA.ALPHA", 0.9, "DOMAIN_ADAPT.TEACHER.THRESHOLD", 0.011, "SOLVER.BASE_LR", 1e-3])
    random.seed(0)
    torch.manual_seed(1)
    tr = ALDITrainer(cfg)
    W, T = tr.model.weights, tr.ema.model.weights
    w0 = W.master.clone()
    losses = []
    for it in range(3):
        tr.iter = it
        tr.before_step()
        at_ema = W.master.clone()                    # the EMA tick of `before_step` saw these student weights
        tr.run_step(); tr.after_step()
        losses.append({k: float(v) for k, v in tr._trainer.last_loss_dict.items()})
    torch.cuda.synchronize()
    assert all(v == v and abs(v) < 1e4 for d in losses for v in d.values()), losses
    keys = set(losses[-1])
    assert {"loss_ce", "loss_bbox", "loss_giou", "loss_ce_0"} <= {k.split("_source")[0].split("_pseudo")[0].split("_distill")[0] for k in keys} or any("loss_ce" in k for k in keys), keys
    moved = (W.master - w0).abs()
    nb = W.nb
    assert float(moved[nb:].max()) > 0 and float(moved[:nb].max()) > 0                       # transformer and trunk both stepped
    frozen = W.backbone.layout.ranges(["backbone.bottom_up.stem.conv1", "backbone.bottom_up.res2.0.conv1"])
    assert all(float(moved[a:b].max()) == 0.0 for a, b in frozen)                          # FREEZE_AT = 2
    (a, b), = W.ranges(["query_embed.weight"])
    assert torch.equal(T.master[a:b], at_ema[a:b])                                          # copied, not averaged
    (a, b), = W.ranges(["class_embed.weight"])
    assert not torch.equal(T.master[a:b], at_ema[a:b])
    # the pseudo labels the teacher produced at this threshold were consumed by the student step
    assert tr.ema.model._last_inference.pseudo["count"].sum().item() > 0


def test_fused_student_pass_equals_the_sequential_schedule():
    """SOLVER.FUSED_STEP on the Deformable-DETR detector (trainer.fused_run_model_detr / DeformableDETR.forward_fused): the source chunk and the
    pseudo-labelled target chunk through ONE trunk + transformer pass and ONE backward, the set criterion per chunk == the reference's
    sequential micro-steps (model(source) + backward, distiller(weak, strong) + backward): the same loss keys and values, the same gradient,
    the same weights after the AdamW step.  Dropout off (its masks are drawn per site over the pass's shapes: another stream for the fused batch)."""
    import os
    import random
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for fused in (True, False):
        cfg = get_cfg()
        add_aldi_config(cfg)
        cfg.merge_from_file(os.path.join(root, "configs", "cityscapes", "ALDI-Best-DETR-Cityscapes.yaml"))
        cfg.merge_from_list(["MODEL.DEFORMABLE_DETR.TRANSFORMER.NUM_QUERIES", 40, "MODEL.DEFORMABLE_DETR.TRANSFORMER.ENC_LAYERS", 2,
                             "MODEL.DEFORMABLE_DETR.TRANSFORMER.DEC_LAYERS", 2, "MODEL.DEFORMABLE_DETR.TRANSFORMER.DROPOUT", 0.0, "SEED", 3, "SOLVER.IMS_PER_BATCH", 4,
                             "SOLVER.IMS_PER_GPU", 2, "SOLVER.WARMUP_ITERS", 0, "SYNTHETIC.HEIGHT", 160, "SYNTHETIC.WIDTH", 224, "EMA.ALPHA", 0.9,
                             "DOMAIN_ADAPT.TEACHER.THRESHOLD", 0.011, "SOLVER.BASE_LR", 1e-4])
        cfg.SOLVER.FUSED_STEP = fused
        random.seed(0)
        torch.manual_seed(1)
        tr = ALDITrainer(cfg)
        W = tr.model.weights
        steps = []
        for it in range(2):
            tr.iter = it
            tr.before_step()
            tr.run_step()
            g = W.grad.clone()
            tr.after_step()
            torch.cuda.synchronize()
            steps.append(({k: float(v) for k, v in tr._trainer.last_loss_dict.items()}, g, W.master.clone()))
        assert int(getattr(tr._trainer, "_detr_fused_steps", 0)) == (2 if fused else 0)
        assert tr.ema.model._last_inference.pseudo["count"].sum().item() > 0
        out[fused] = steps
    for (la, ga, wa), (lb, gb, wb) in zip(out[True], out[False]):
        assert list(la) == list(lb) and any(k.endswith("_distill") for k in la) and any(k.endswith("_source_strong") for k in la)
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-4 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
        # (fp32 sums in another order: the fused pass adds both chunks' weight gradients in one launch)
        assert float((ga - gb).norm() / gb.norm()) <= 2e-3, float((ga - gb).norm() / gb.norm())
        assert float((wa - wb).abs().max()) <= 1e-4 * float(wb.abs().max())


def test_fused_student_pass_steps_aside_for_unequal_canvases_and_foreign_distillers():
    from aldi_amd.distill import HardDistiller

    class T:
        pass
    t = T()
    from aldi_amd import trainer as TR
    t.model_batch_size, t.fused, t.backward_at_end = 2, True, False
    t.model = T(); t.model.detr = True
    t.model.img_align = t.model.ins_align = None
    t.distiller = HardDistiller.__new__(HardDistiller)
    t.distiller.do_hard_cls = True
    t.distiller.do_hard_obj = t.distiller.do_hard_rpn_reg = t.distiller.do_hard_roi_reg = False
    t.distiller.student = t.distiller.teacher = T()
    img = lambda h, w: {"image": torch.zeros(3, h, w, dtype=torch.uint8)}
    can = lambda data: TR._ALDITrainer._can_fuse_detr(t, data)
    same = [img(160, 224), img(150, 200)]
    assert can((None, same, same, same))
    assert not can((None, same, same, [img(160, 224), img(200, 300)]))            # the target chunk's canvas is larger than the source chunk's
    assert not can((same, same, same, same))                                        # labeled_weak rows are not part of this schedule
    t.backward_at_end = True
    assert not can((None, same, same, same))
    t.backward_at_end = False

    class Mine(HardDistiller):
        pass
    t.distiller.__class__ = Mine
    assert not can((None, same, same, same))
