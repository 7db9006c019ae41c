"""HIP kernels (through the C ABI) vs the reference's golden vectors and vs the CPU oracle."""
import math
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------------
# ALDI-owned losses vs the reference's own outputs (golden fixtures)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["r256k8", "r64k80", "nofg"])
def test_roih_distill_vs_reference_golden(golden_dir, tag):
    from aldi_amd import ops
    g = np.load(os.path.join(golden_dir, "g2_roih_losses.npz"))
    sl, tl, sd, td = (T(g[f"{tag}_{n}"]) for n in ("s_logits", "t_logits", "s_deltas", "t_deltas"))
    R, K1 = sl.shape
    K = K1 - 1
    Cp = (5 * K + 1 + 15) // 16 * 16
    sp = torch.zeros(R, Cp)
    tp = torch.zeros(R, Cp)
    sp[:, :K1], sp[:, K1:K1 + 4 * K] = sl, sd
    tp[:, :K1], tp[:, K1:K1 + 4 * K] = tl, td
    spd, tpd = sp.to(DEV), tp.to(DEV)
    for lt in ("CE", "KL"):
        for Tm in (1.0, 0.5):
            loss = torch.zeros(2, device=DEV)
            grad = torch.zeros(R, Cp, device=DEV)
            ops.roih_distill_loss(spd, tpd, Cp, K, R, Tm, lt == "KL", True, True, 1.0, grad, loss)
            key = f"{tag}_{lt}_T{Tm}"
            assert abs(float(loss[0]) - float(g[key + "_loss_cls_ce"])) < 2e-5 * max(1.0, abs(float(g[key + "_loss_cls_ce"])))
            assert abs(float(loss[1]) - float(g[key + "_loss_roih_l1"])) < 2e-5 * max(1.0, abs(float(g[key + "_loss_roih_l1"])))
            gc = grad.cpu()
            np.testing.assert_allclose(gc[:, :K1].numpy(), g[key + "_dlogits"], rtol=1e-4, atol=2e-7)
            np.testing.assert_allclose(gc[:, K1:K1 + 4 * K].numpy(), g[key + "_ddeltas"], rtol=1e-5, atol=1e-8)
            assert float(gc[:, K1 + 4 * K:].abs().max()) == 0.0 if Cp > K1 + 4 * K else True


@pytest.mark.parametrize("tag", ["mix", "zerofg"])
def test_rpn_distill_index_quirk_vs_reference_golden(golden_dir, tag):
    """The mask built in (N, sumA) order is applied to the level-major RAW (N,A,H,W) flattening."""
    from aldi_amd import ops
    g = np.load(os.path.join(golden_dir, "g3_rpn_losses.npz"))
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    A, N, C = 3, 2, 16
    geom = ops.make_geom(shapes, A, C)

    def pack(lo, de):
        out = []
        for l, (h, w) in enumerate(shapes):
            t = torch.zeros(N, h, w, C)
            t[..., :A] = T(lo[l]).permute(0, 2, 3, 1)
            t[..., A:5 * A] = T(de[l]).permute(0, 2, 3, 1)
            out.append(t.to(DEV))
        return out
    nl = len(shapes)
    s_head = pack([g[f"{tag}_s_logits{l}"] for l in range(nl)], [g[f"{tag}_s_deltas{l}"] for l in range(nl)])
    t_head = pack([g[f"{tag}_t_logits{l}"] for l in range(nl)], [g[f"{tag}_t_deltas{l}"] for l in range(nl)])
    labels = T(g[f"{tag}_labels"]).to(torch.int32)
    n_valid, n_fg = int((labels >= 0).sum()), int((labels == 1).sum())
    ld = labels.to(DEV)
    for Tm in (1.0, 0.5):
        grads = [torch.zeros_like(h) for h in s_head]
        loss = torch.zeros(2, device=DEV)
        ops.rpn_distill_loss(geom, s_head, t_head, grads, ld, N, Tm, n_valid, n_fg, True, True, 1.0, loss)
        key = f"{tag}_T{Tm}"
        assert abs(float(loss[0]) - float(g[key + "_loss_obj_bce"])) < 2e-6 * max(1.0, float(g[key + "_loss_obj_bce"]))
        assert abs(float(loss[1]) - float(g[key + "_loss_rpn_l1"])) < 2e-6 * max(1.0, float(g[key + "_loss_rpn_l1"]))
        for l in range(nl):
            gl = grads[l].cpu()
            np.testing.assert_allclose(gl[..., :A].permute(0, 3, 1, 2).numpy(), g[f"{key}_dlogits{l}"], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(gl[..., A:5 * A].permute(0, 3, 1, 2).numpy(), g[f"{key}_ddeltas{l}"], rtol=1e-5, atol=1e-9)


def test_discriminators_grl_bce_vs_reference_golden(golden_dir):
    """ConvDiscriminator / FCDiscriminator + gradient reversal + domain BCE, fwd and bwd (aldi/align.py:76-135)."""
    from aldi_amd import ops
    g = np.load(os.path.join(golden_dir, "g1_discriminators.npz"))
    f32 = torch.float32

    def pad_rows(w, rows=8):
        out = torch.zeros((rows,) + tuple(w.shape[1:]))
        out[: w.shape[0]] = w
        return out
    for labeled in (1, 0):
        # ---- conv discriminator
        x = nhwc(T(g["conv_x"])).to(DEV)
        w1 = T(g["conv_sd.model.0.weight"]).permute(0, 2, 3, 1).contiguous().to(DEV)
        b1 = T(g["conv_sd.model.0.bias"]).to(DEV)
        w2 = pad_rows(T(g["conv_sd.model.4.weight"])).view(8, 1, 1, -1).contiguous().to(DEV)
        b2 = pad_rows(T(g["conv_sd.model.4.bias"])).to(DEV)
        a1 = ops.conv2d(x, w1, shift=b1, relu=True)
        pooled = ops.avgpool(a1)
        logit = ops.conv2d(pooled, w2, shift=b2, want_f32=True)
        np.testing.assert_allclose(logit.view(-1, 8)[:, :1].cpu().numpy(), g["conv_preds"], rtol=1e-5, atol=1e-6)
        loss = torch.zeros(1, device=DEV)
        glog = torch.empty(logit.shape, dtype=f32, device=DEV)
        ops.domain_bce(logit, 8, x.shape[0], float(labeled), 0.01, 1.0, glog, loss)
        assert abs(float(loss) - float(g[f"conv_loss_l{labeled}"])) < 1e-7
        dw2 = torch.zeros_like(w2)
        ops.conv_wgrad(pooled, glog, dw2, KH=1, KW=1)
        np.testing.assert_allclose(dw2.view(8, -1)[:1].cpu().numpy(), g[f"conv_grad_l{labeled}.model.4.weight"], rtol=1e-4, atol=1e-9)
        g_pooled = ops.conv2d(glog, ops.dgrad_weights(w2, None, f32))
        g_a1 = ops.avgpool_bwd(g_pooled, a1)
        dw1 = torch.zeros_like(w1)
        ops.conv_wgrad(x, g_a1, dw1, KH=3, KW=3)
        np.testing.assert_allclose(dw1.cpu().permute(0, 3, 1, 2).numpy(), g[f"conv_grad_l{labeled}.model.0.weight"], rtol=2e-4, atol=1e-9)
        db1 = torch.zeros_like(b1)
        ops.bias_grad(g_a1, db1)
        np.testing.assert_allclose(db1.cpu().numpy(), g[f"conv_grad_l{labeled}.model.0.bias"], rtol=2e-4, atol=1e-9)
        neg = torch.full((w1.shape[0],), -1.0, device=DEV)
        dx = ops.conv2d(g_a1, ops.dgrad_weights(w1, neg, f32), pad=2)       # gradient reversal folded into the dgrad weights
        np.testing.assert_allclose(dx.cpu().permute(0, 3, 1, 2).numpy(), g[f"conv_dx_l{labeled}"], rtol=2e-4, atol=1e-10)
        # ---- fc discriminator
        xf = T(g["fc_x"]).view(-1, 1, 1, 64).contiguous().to(DEV)
        v1 = T(g["fc_sd.model.1.weight"]).view(64, 1, 1, 64).contiguous().to(DEV)
        c1 = T(g["fc_sd.model.1.bias"]).to(DEV)
        v2 = pad_rows(T(g["fc_sd.model.3.weight"])).view(8, 1, 1, 64).contiguous().to(DEV)
        c2 = pad_rows(T(g["fc_sd.model.3.bias"])).to(DEV)
        h = ops.conv2d(xf, v1, shift=c1, relu=True)
        logit = ops.conv2d(h, v2, shift=c2, want_f32=True)
        np.testing.assert_allclose(logit.view(-1, 8)[:, :1].cpu().numpy(), g["fc_preds"], rtol=1e-5, atol=1e-6)
        loss = torch.zeros(1, device=DEV)
        glog = torch.empty(logit.shape, dtype=f32, device=DEV)
        ops.domain_bce(logit, 8, xf.shape[0], float(labeled), 0.01, 1.0, glog, loss)
        assert abs(float(loss) - float(g[f"fc_loss_l{labeled}"])) < 1e-7
        g_h = ops.conv2d(glog, ops.dgrad_weights(v2, None, f32), mask=h)
        dv1 = torch.zeros_like(v1)
        ops.conv_wgrad(xf, g_h, dv1, KH=1, KW=1)
        np.testing.assert_allclose(dv1.view(64, 64).cpu().numpy(), g[f"fc_grad_l{labeled}.model.1.weight"], rtol=2e-4, atol=1e-10)
        neg = torch.full((64,), -1.0, device=DEV)
        dxf = ops.conv2d(g_h, ops.dgrad_weights(v1, neg, f32))
        np.testing.assert_allclose(dxf.view(-1, 64).cpu().numpy(), g[f"fc_dx_l{labeled}"], rtol=2e-4, atol=1e-10)


def test_ema_bit_exact_vs_reference_golden(golden_dir):
    from aldi_amd import ops
    g = np.load(os.path.join(golden_dir, "g5_ema.npz"))
    keys = [k[3:] for k in g.files if k.startswith("t0.") and g[k].dtype == np.float32]
    t0 = torch.cat([T(g["t0." + k]).reshape(-1) for k in keys]).to(DEV)
    s = torch.cat([T(g["s." + k]).reshape(-1) for k in keys]).to(DEV)
    exp = torch.cat([T(g["t_after_ema." + k]).reshape(-1) for k in keys if "query_embed" not in k])
    sel = torch.cat([torch.full((g["t0." + k].size,), "query_embed" not in k) for k in keys])
    ops.ema_update(t0, s, None, t0.numel(), float(g["alpha"]), False, torch.float32)
    assert torch.equal(t0.cpu()[sel], exp)                                   # bit-exact lerp
    ops.ema_update(t0, s, None, t0.numel(), float(g["alpha"]), True, torch.float32)
    assert torch.equal(t0.cpu(), s.cpu())                                    # iter <= start_iter: copy


def test_sgd_vs_torch_optim():
    from aldi_amd import ops
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(10000, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=0.05, momentum=0.9, weight_decay=1e-4)
    pd, buf = p0.to(DEV), torch.zeros(10000, device=DEV)
    pc = torch.zeros(10000, dtype=torch.bfloat16, device=DEV)
    for it in range(3):
        gr = torch.randn(10000, generator=g)
        p.grad = gr.clone()
        opt.step()
        ops.sgd_step(pd, gr.to(DEV), buf, pc, 10000, 0.05, 0.9, 1e-4, 1.0, it == 0, torch.bfloat16)
    assert (pd.cpu() - p.detach()).abs().max() < 1e-6
    assert torch.equal(pc.cpu(), pd.cpu().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------
# Detectron2-side kernels vs the CPU oracle
# ------------------------------------------------------------------------------------------------
def test_stem_and_pool_vs_oracle():
    from aldi_amd import ops, synthetic as syn
    from oracle import d2_rcnn as d2
    sd = syn.init_state_dict(8, seed=2)
    cfg = d2.make_cfg(num_classes=8)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, 75, 100), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 96, 90), generator=g, dtype=torch.uint8)]
    x, sizes = d2.preprocess(cfg, imgs)
    p = "backbone.bottom_up.stem.conv1"
    ref = F.max_pool2d(F.relu(d2.conv_bn(x, sd, p, 2, 3)), 3, 2, 1)
    st = torch.zeros(2, 3, x.shape[2], x.shape[3], dtype=torch.uint8)
    for i, im in enumerate(imgs):
        st[i, :, : im.shape[1], : im.shape[2]] = im
    scale = sd[p + ".norm.weight"] * (sd[p + ".norm.running_var"] + 1e-5).rsqrt()
    shift = sd[p + ".norm.bias"] - sd[p + ".norm.running_mean"] * scale
    w = sd[p + ".weight"].permute(0, 2, 3, 1).contiguous()
    y = ops.stem_forward(st.to(DEV), sizes, w.to(DEV), scale.to(DEV), shift.to(DEV), cfg["pixel_mean"], cfg["pixel_std"], torch.float32)
    y = ops.maxpool3s2(y)
    torch.cuda.synchronize()
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max() < 2e-5 * ref.abs().max()
    # bf16 mode runs the stem on the matrix cores (bf16 image / weights, fp32 accumulation): the oracle with its inputs
    # rounded to bf16 is the yardstick, up to accumulation order and the bf16 rounding of the output
    xb, wb = x.bfloat16().float(), sd[p + ".weight"].bfloat16().float()
    refb = F.relu(F.conv2d(xb, wb, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    yb = ops.stem_forward(st.to(DEV), sizes, w.to(DEV), scale.to(DEV), shift.to(DEV), cfg["pixel_mean"], cfg["pixel_std"], torch.bfloat16)
    torch.cuda.synchronize()
    err = (yb.float().cpu().permute(0, 3, 1, 2) - refb).abs().max()
    assert err < 8e-3 * refb.abs().max(), float(err / refb.abs().max())


def _rand_boxes(n, w, h, g, lo=4.0, hi=120.0):
    x1 = torch.rand(n, generator=g) * (w - lo)
    y1 = torch.rand(n, generator=g) * (h - lo)
    bw = lo + torch.rand(n, generator=g) * (hi - lo)
    bh = lo + torch.rand(n, generator=g) * (hi - lo)
    return torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)


@pytest.mark.parametrize("empty_gt", [False, True])
def test_matcher_and_sampling_lists_bit_exact(empty_gt):
    from aldi_amd import ops
    from aldi_amd.engine import GMAX, make_anchors
    from oracle import d2_rcnn as d2
    shapes = [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)]
    anchors = make_anchors(shapes, DEV)
    cfg = d2.make_cfg(num_classes=8)
    oa = d2.generate_anchors(cfg, shapes)
    assert torch.equal(torch.cat(oa), anchors.cpu())                         # anchors bit-identical
    g = torch.Generator().manual_seed(3)
    N, sumA = 2, anchors.shape[0]
    gts = [_rand_boxes(7, 208, 160, g), torch.zeros(0, 4) if empty_gt else _rand_boxes(3, 208, 160, g)]
    gts[0][1] = torch.cat(oa)[1234]                                           # an exact-anchor GT (IoU == 1) and a degenerate far-away GT
    gts[0][2] = torch.tensor([5000.0, 5000.0, 5010.0, 5010.0])
    gb = torch.zeros(N, GMAX, 4)
    cnt = torch.zeros(N, dtype=torch.int32)
    for i, b in enumerate(gts):
        gb[i, : len(b)] = b
        cnt[i] = len(b)
    best_iou = torch.empty(N, sumA, device=DEV)
    best_idx = torch.empty(N, sumA, dtype=torch.int32, device=DEV)
    labels = torch.empty(N, sumA, dtype=torch.int32, device=DEV)
    scratch = torch.empty(N, GMAX, dtype=torch.int32, device=DEV)
    ops.box_match(anchors, 0, None, sumA, gb.to(DEV), cnt.to(DEV), GMAX, N, 0.3, 0.7, True, best_iou, best_idx, scratch, labels)
    lists = torch.empty(N, 2, sumA, dtype=torch.int32, device=DEV)
    counts = torch.empty(N, 2, dtype=torch.int32, device=DEV)
    ops.compact_labels(labels, sumA, N, 0, lists, counts)
    for n in range(N):
        mqm = d2.pairwise_iou(gts[n], torch.cat(oa))
        midx, lab = d2.matcher(mqm, cfg["rpn_iou_thresholds"], [0, -1, 1], True)
        assert torch.equal(labels[n].cpu(), lab.to(torch.int32))
        if len(gts[n]):
            assert torch.equal(best_idx[n].cpu(), midx.to(torch.int32))
        pos = torch.nonzero(lab == 1).squeeze(1).to(torch.int32)
        neg = torch.nonzero(lab == 0).squeeze(1).to(torch.int32)
        c = counts[n].tolist()
        assert c == [len(pos), len(neg)]
        assert torch.equal(lists[n, 0, : c[0]].cpu(), pos) and torch.equal(lists[n, 1, : c[1]].cpu(), neg)
    # the degenerate GT (zero IoU with every anchor) labels everything positive in Detectron2's low-quality rule
    assert int((labels[0] == 1).sum()) == sumA


@pytest.mark.parametrize("shapes", [[(48, 64), (24, 32), (12, 16), (6, 8), (3, 4)],
                                    [(104, 152), (52, 76), (26, 38), (13, 19), (7, 10)]])   # p2: 2 / 6 radix-select chunks, ragged last chunk
@pytest.mark.parametrize("quantize", [False, True])
def test_rpn_proposals_bit_exact_vs_oracle(quantize, shapes):
    """top-k -> decode -> clip -> drop empty -> batched NMS -> top-k: identical ORDER and boxes; quantised logits force score ties."""
    from aldi_amd import ops
    from aldi_amd.engine import make_anchors
    from oracle import d2_rcnn as d2
    N, A, C = 2, 3, 16
    cfg = d2.make_cfg(num_classes=8)
    anchors = make_anchors(shapes, DEV)
    oa = d2.generate_anchors(cfg, shapes)
    g = torch.Generator().manual_seed(5)
    heads, lo, de = [], [], []
    for (h, w) in shapes:
        t = torch.randn(N, h, w, C, generator=g)
        t[..., A:] *= 0.5
        if quantize:
            t[..., :A] = (t[..., :A] * 2).round() / 2
        heads.append(t.to(DEV))
        lo.append(t[..., :A].reshape(N, -1))                                  # (N, HWA)
        de.append(t[..., A:5 * A].reshape(N, h * w * A, 4))
    sizes = [(shapes[0][0] * 4 - 12, shapes[0][1] * 4 - 6), (shapes[0][0] * 4, shapes[0][1] * 4)]
    geom = ops.make_geom(shapes, A, C)
    hw = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    for training, pre, post in ((True, 2000, 1000), (False, 1000, 1000)):
        ws = torch.empty(ops.rpn_proposals_workspace(N, 5), dtype=torch.uint8, device=DEV)
        boxes = torch.empty(N, post, 4, device=DEV)
        scores = torch.empty(N, post, device=DEV)
        count = torch.empty(N, dtype=torch.int32, device=DEV)
        err = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.rpn_proposals(geom, heads, anchors, hw, N, pre, post, 0.7, ws, boxes, scores, count, err)
        ref = d2.find_top_rpn_proposals(cfg, oa, lo, de, sizes, training)
        assert int(err) == 0
        for n in range(N):
            k = int(count[n])
            assert k == len(ref[n]["proposal_boxes"])
            assert torch.equal(scores[n, :k].cpu(), ref[n]["objectness_logits"])
            assert (boxes[n, :k].cpu() - ref[n]["proposal_boxes"]).abs().max() < 1e-3      # expf/ulps only; the ORDER is exact
            # size-independent properties: sorted by score, inside the image, non-empty
            assert bool((scores[n, 1:k] <= scores[n, : k - 1]).all())
            b = boxes[n, :k]
            assert bool((b[:, 0] >= 0).all() and (b[:, 2] <= sizes[n][1]).all() and (b[:, 3] <= sizes[n][0]).all())
            assert bool(((b[:, 2] - b[:, 0]) > 0).all() and ((b[:, 3] - b[:, 1]) > 0).all())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_roialign_fwd_bwd_vs_oracle(dtype, tol):
    from aldi_amd import ops
    from oracle import d2_rcnn as d2
    g = torch.Generator().manual_seed(7)
    N, C = 2, 256
    shapes = [(112, 160), (56, 80), (28, 40), (14, 20)]
    feats = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    if dtype == torch.bfloat16:
        feats = [f.to(dtype).float() for f in feats]
    boxes = torch.cat([_rand_boxes(30, 640, 448, g, lo=2.0, hi=100.0), _rand_boxes(30, 640, 448, g, lo=100.0, hi=640.0)])   # all four levels
    boxes[0] = torch.tensor([-20.0, -10.0, 30.0, 25.0])                       # partially outside
    boxes[1] = torch.tensor([600.0, 400.0, 700.0, 500.0])
    bidx = torch.randint(0, N, (60,), generator=g)
    per_img = [boxes[bidx == i] for i in range(N)]
    cfg = d2.make_cfg(num_classes=8)
    fr = [f.clone().requires_grad_(True) for f in feats]
    ref = d2.roi_pool(cfg, fr, per_img)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(per_img)]).contiguous().to(DEV)
    fd = [nhwc(f).to(DEV, dtype) for f in feats]
    grads = [torch.zeros(f.shape, dtype=torch.float32, device=DEV) for f in fd]
    R = rois.shape[0]
    pooled = torch.empty(R, 7, 7, C, dtype=dtype, device=DEV)
    ops.roialign(ops.make_roi_feats(fd, None, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, pooled, backward=False)
    assert (pooled.float().cpu().permute(0, 3, 1, 2) - ref.detach()).abs().max() < tol * 10
    gd = nhwc(go).to(DEV, dtype)
    ops.roialign(ops.make_roi_feats(fd, grads, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gd, backward=True)
    assert sorted(set(d2.assign_levels(torch.cat(per_img)).tolist())) == [0, 1, 2, 3]
    for l in range(4):
        e = (grads[l].cpu().permute(0, 3, 1, 2) - fr[l].grad).abs().max()
        assert e < tol * 30 * max(1.0, float(fr[l].grad.abs().max())), (l, float(e))
    # gather form of the backward (what the engine runs): overwrites uninitialised maps, no atomics, same values
    grads2 = [torch.full(f.shape, float("nan"), dtype=torch.float32, device=DEV) for f in fd]
    ops.roialign_backward(ops.make_roi_feats(fd, grads2, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gd, N)
    for l in range(4):
        e = (grads2[l].cpu().permute(0, 3, 1, 2) - fr[l].grad).abs().max()
        assert e < tol * 30 * max(1.0, float(fr[l].grad.abs().max())), (l, float(e))
        assert (grads2[l] - grads[l]).abs().max() < 2e-5 * max(1.0, float(grads[l].abs().max()))      # scatter vs gather
    again = [torch.empty_like(t) for t in grads2]
    ops.roialign_backward(ops.make_roi_feats(fd, again, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gd, N)
    assert all(torch.equal(a, b) for a, b in zip(again, grads2))                                       # deterministic
    # rows grouped by image (as the engine produces them): each workgroup scans its image's rows only -- same bits
    srt = [torch.empty_like(t) for t in grads2]
    ops.roialign_backward(ops.make_roi_feats(fd, srt, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gd, N, rois_sorted=True)
    assert all(torch.equal(a, b) for a, b in zip(srt, grads2))
    if dtype == torch.bfloat16:
        # bf16 gradient maps (what the bf16 step's FPN backward consumes): the same fp32 sums, rounded once
        lo = [torch.full(f.shape, float("nan"), dtype=torch.bfloat16, device=DEV) for f in fd]
        ops.roialign_backward(ops.make_roi_feats(fd, lo, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gd, N, rois_sorted=True, grad_dtype=torch.bfloat16)
        assert all(torch.equal(a, b.to(torch.bfloat16)) for a, b in zip(lo, grads2))
    # property: pooling a constant map returns the constant wherever the ROI lies inside the map
    const = [torch.full(f.shape, 3.0, dtype=dtype, device=DEV) for f in fd]
    inside = ((rois[:, 1] >= 0) & (rois[:, 2] >= 0) & (rois[:, 3] <= 640) & (rois[:, 4] <= 448)).nonzero().squeeze(1)
    rin = rois[inside].contiguous()
    pin = torch.empty(len(inside), 7, 7, C, dtype=dtype, device=DEV)
    ops.roialign(ops.make_roi_feats(const, None, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rin, len(inside), 7, pin, backward=False)
    assert len(inside) >= 50 and float((pin.float() - 3.0).abs().max()) < 1e-5


def test_roialign_backward_two_row_kernel_is_the_one_row_kernel_bit_for_bit():
    """aldi_roialign_backward on bf16 pooled gradients runs two feature rows per workgroup (knob roialign_bwd_rows = 2): same terms in the
    same order as the one-row kernel (= 1).  Odd map heights (a last row pair with one row), 700 ROIs per image (three scan chunks),
    ROIs on three levels, rows sorted by image and not, fp32 and bf16 gradient maps."""
    from aldi_amd import ops, _lib as L
    g = torch.Generator().manual_seed(11)
    N, C = 2, 256
    shapes = [(51, 67), (26, 34), (13, 17), (7, 9)]
    per = 700
    boxes = torch.cat([_rand_boxes(per * N - 200, 268, 204, g, lo=2.0, hi=60.0), _rand_boxes(200, 268, 204, g, lo=60.0, hi=268.0)])
    boxes[0] = torch.tensor([-20.0, -10.0, 30.0, 25.0])
    boxes[1] = torch.tensor([250.0, 190.0, 300.0, 230.0])
    perm = torch.randperm(per * N, generator=g)
    boxes = boxes[perm]
    img = (torch.arange(per * N) // per).float()
    rois_sorted = torch.cat([img[:, None], boxes], 1).contiguous().to(DEV)
    shuffle = torch.randperm(per * N, generator=g)
    gp = torch.randn(per * N, 7, 7, C, generator=g).to(DEV, torch.bfloat16)
    fd = [torch.zeros(N, h, w, C, dtype=torch.bfloat16, device=DEV) for h, w in shapes]
    R = per * N
    try:
        for srt in (True, False):
            rois = rois_sorted if srt else rois_sorted[shuffle.to(DEV)].contiguous()
            gpx = gp if srt else gp[shuffle.to(DEV)].contiguous()
            for gdt in (torch.float32, torch.bfloat16):
                out = {}
                for rows in (1, 2):
                    L.reset_tuning(); L.set_tuning("roialign_bwd_rows", rows)
                    maps = [torch.full(f.shape, float("nan"), dtype=gdt, device=DEV) for f in fd]
                    ops.roialign_backward(ops.make_roi_feats(fd, maps, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gpx, N, rois_sorted=srt, grad_dtype=gdt)
                    torch.cuda.synchronize()
                    out[rows] = maps
                for l in range(4):
                    assert bool(torch.isfinite(out[2][l].float()).all()), (srt, gdt, l)
                    assert torch.equal(out[1][l], out[2][l]), (srt, gdt, l)
                assert sum(float(m.float().abs().sum()) for m in out[2]) > 0
        # a smaller pooling grid (P = 5), an image without any ROI (its maps must come back zero), and no ROI at all
        N3 = 3
        fd3 = [torch.zeros(N3, h, w, C, dtype=torch.bfloat16, device=DEV) for h, w in shapes]
        gp5 = torch.randn(R, 5, 5, C, generator=g).to(DEV, torch.bfloat16)
        for R_ in (R, 0):
            out = {}
            for rows in (1, 2):
                L.reset_tuning(); L.set_tuning("roialign_bwd_rows", rows)
                maps = [torch.full(f.shape, float("nan"), dtype=torch.bfloat16, device=DEV) for f in fd3]
                ops.roialign_backward(ops.make_roi_feats(fd3, maps, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois_sorted, R_, 5, gp5, N3, rois_sorted=True, grad_dtype=torch.bfloat16)
                torch.cuda.synchronize()
                out[rows] = maps
            for l in range(4):
                assert torch.equal(out[1][l], out[2][l]), (R_, l)
                assert float(out[2][l][2].float().abs().max()) == 0.0            # the image nobody pooled from
                if R_ == 0:
                    assert float(out[2][l].float().abs().max()) == 0.0
            if R_:
                assert float(out[2][0][:2].float().abs().max()) > 0
    finally:
        L.reset_tuning()


def test_roialign_forward_and_backward_are_adjoint_at_benchmark_size():
    """size-independent property at BASELINE's full size (4 images of 1333 x 800: maps 200 x 336 ... 25 x 42 x 256 channels, 2048 ROIs):
    the backward is the transpose of the forward, <pool(f), g> == <f, pool^T(g)>.  f and g hold bf16-representable values; the forward runs
    in fp32 on them (no output rounding), the backward is the benchmark step's kernel (bf16 pooled gradients, two rows per workgroup) writing
    fp32 maps, and again the one-row kernel: both sides are fp32 sums of the same products."""
    from aldi_amd import ops, _lib as L
    g = torch.Generator().manual_seed(3)
    N, C, R = 4, 256, 2048
    Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
    w = torch.rand(R, generator=g) * 100 + 28
    h = torch.rand(R, generator=g) * 50 + 15
    big = torch.rand(R, generator=g) < 0.05                                       # a few ROIs for the coarse levels
    w = torch.where(big, w * 6, w); h = torch.where(big, h * 8, h)
    cx, cy = torch.rand(R, generator=g) * 1333, torch.rand(R, generator=g) * 800
    rois = torch.stack([(torch.arange(R) // (R // N)).float(), (cx - w / 2).clamp(0, 1332), (cy - h / 2).clamp(0, 799), (cx + w / 2).clamp(1, 1333),
                        (cy + h / 2).clamp(1, 800)], 1).contiguous().to(DEV)
    feats = [torch.randn(N, Hs[l], Ws[l], C, generator=g).to(torch.bfloat16).float().to(DEV) for l in range(4)]
    gp = torch.randn(R, 7, 7, C, generator=g).to(torch.bfloat16).to(DEV)
    pooled = torch.empty(R, 7, 7, C, dtype=torch.float32, device=DEV)
    ops.roialign(ops.make_roi_feats(feats, None, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, pooled, backward=False)
    lhs = float((pooled.double() * gp.double()).sum())
    scale = float(pooled.double().norm() * gp.double().norm())
    fb = [f.to(torch.bfloat16) for f in feats]
    try:
        for rows in (2, 1):
            L.reset_tuning(); L.set_tuning("roialign_bwd_rows", rows)
            maps = [torch.full(f.shape, float("nan"), dtype=torch.float32, device=DEV) for f in feats]
            ops.roialign_backward(ops.make_roi_feats(fb, maps, [1 / 4, 1 / 8, 1 / 16, 1 / 32]), rois, R, 7, gp, N, rois_sorted=True)
            rhs = sum(float((f.double() * m.double()).sum()) for f, m in zip(feats, maps))
            assert abs(lhs - rhs) <= 2e-6 * scale, (rows, lhs, rhs, scale)
            assert all(float(m.abs().sum()) > 0 for m in maps)                     # every level was pooled from
    finally:
        L.reset_tuning()


def test_roi_prepare_lists_equals_the_eight_launch_path():
    """aldi_roi_prepare_lists (one launch; the image's last workgroup writes the lists) against aldi_roi_prepare + aldi_compact_labels: candidates,
    matches, classes, the ordered foreground / background lists and their lengths, bit for bit -- images with no ground truth, with a full ground
    truth table, with fewer proposals than slots, with ground truth exactly equal to a proposal (IoU = 1) and an IoU tie; called three times on
    the same ticket words (they must come back zero)."""
    from aldi_amd import ops
    from aldi_amd.engine import GMAX
    g = torch.Generator().manual_seed(5)
    N, P, K = 5, 1000, 8
    Lc = P + GMAX
    props = torch.zeros(N, P, 4)
    pcount = torch.tensor([1000, 640, 1000, 3, 1000], dtype=torch.int32)
    gcount = torch.tensor([7, 0, GMAX, 2, 100], dtype=torch.int32)
    gtb = torch.zeros(N, GMAX, 4)
    gtc = torch.zeros(N, GMAX, dtype=torch.int32)
    for n in range(N):
        props[n, : int(pcount[n])] = _rand_boxes(int(pcount[n]), 1333, 800, g, lo=8.0, hi=400.0)
        gtb[n, : int(gcount[n])] = _rand_boxes(int(gcount[n]), 1333, 800, g, lo=16.0, hi=300.0)
        gtc[n, : int(gcount[n])] = torch.randint(0, K, (int(gcount[n]),), generator=g, dtype=torch.int32)
    gtb[0, 1] = props[0, 10]                       # IoU exactly 1
    gtb[0, 2] = gtb[0, 1]                          # ... and a tie: the first maximum wins
    props[2, 5] = gtb[2, 77]
    dev = DEV
    props, pcount, gcount, gtb, gtc = props.to(dev), pcount.to(dev), gcount.to(dev), gtb.to(dev), gtc.to(dev)

    def bufs():
        return dict(cand=torch.full((N, Lc, 4), -7.0, device=dev), ccount=torch.full((N,), -7, dtype=torch.int32, device=dev),
                    best_iou=torch.full((N, Lc), -7.0, device=dev), best_idx=torch.full((N, Lc), -7, dtype=torch.int32, device=dev),
                    labels=torch.full((N, Lc), -7, dtype=torch.int32, device=dev), cls=torch.full((N, Lc), -7, dtype=torch.int32, device=dev),
                    lists=torch.full((N, 2, Lc), -7, dtype=torch.int32, device=dev), counts=torch.full((N, 2), -7, dtype=torch.int32, device=dev))
    a = bufs()
    scratch = torch.empty((N, GMAX), dtype=torch.int32, device=dev)
    ops.roi_prepare(props, pcount, P, gtb, gtc, gcount, GMAX, N, K, 0.5, a["cand"], a["ccount"], a["best_iou"], a["best_idx"], scratch, a["labels"], a["cls"])
    ops.compact_labels(a["cls"], Lc, N, K, a["lists"], a["counts"])
    tickets = torch.zeros(N, dtype=torch.int32, device=dev)
    for rep in range(3):
        b = bufs()
        wide = torch.full((2 * N + 2,), -7, dtype=torch.int32, device=dev)                 # counts + two appended words (third repetition)
        if rep == 2:
            b["counts"] = wide[: 2 * N].view(N, 2)
        ta, tb = torch.tensor([41], dtype=torch.int32, device=dev), torch.tensor([42], dtype=torch.int32, device=dev)
        ops.roi_prepare_lists(props, pcount, P, gtb, gtc, gcount, GMAX, N, K, 0.5, b["cand"], b["ccount"], b["best_iou"], b["best_idx"], b["labels"], b["cls"],
                              b["lists"], b["counts"], tickets, *((ta, tb) if rep == 2 else ()))
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        assert rep != 2 or wide[2 * N:].tolist() == [41, 42]
        for k in ("cand", "ccount", "labels", "cls", "counts"):
            assert torch.equal(a[k], b[k]), (rep, k)
        for n in range(N):
            cnt = int(a["ccount"][n])
            assert cnt == int(pcount[n]) + int(gcount[n])
            assert torch.equal(a["best_iou"][n, :cnt], b["best_iou"][n, :cnt]) and torch.equal(a["best_idx"][n, :cnt], b["best_idx"][n, :cnt])
            for kind in (0, 1):
                m = int(a["counts"][n, kind])
                assert torch.equal(a["lists"][n, kind, :m], b["lists"][n, kind, :m]), (rep, n, kind)
    assert int(a["counts"][0, 0]) >= 7 and int(a["counts"][1, 0]) == 0 and int(a["counts"][1, 1]) == 640
    assert int(b["best_idx"][0, 10]) == 1                                      # the tie between ground truth 1 and 2


def test_box_losses_fused_equals_the_separate_launches():
    """aldi_box_losses_fused against aldi_box_loss per chunk + aldi_roih_distill_loss (once with one scale, once split into its two parts with
    different scales) + aldi_cast_from_f32: gradient rows bit for bit (fp32 and the bf16 copy), loss values to the order of the block sums'
    atomic adds."""
    from aldi_amd import ops
    g = torch.Generator().manual_seed(9)
    K, Cp = 8, 48
    chunks = [(0, 700), (700, 700 + 513), (1213, 1213 + 300)]
    R = chunks[-1][1]
    pred = (torch.randn(R, Cp, generator=g) * 2).to(DEV)
    tpred = (torch.randn(R, Cp, generator=g) * 2).to(DEV)
    rois = torch.cat([torch.zeros(R, 1), _rand_boxes(R, 1333, 800, g, lo=8.0, hi=300.0)], 1).contiguous().to(DEV)
    cls = torch.randint(0, K + 1, (R,), generator=g, dtype=torch.int32).to(DEV)
    gtb = _rand_boxes(R, 1333, 800, g, lo=8.0, hi=300.0).to(DEV)
    w4 = (10.0, 10.0, 5.0, 5.0)
    for split_scales in (False, True):
        spec = [dict(gs_cls=0.5, gs_box=0.5), dict(gs_cls=0.0, gs_box=0.25, distill=(0.5, 0.125 if split_scales else 0.5), kl=split_scales),
                dict(gs_cls=1.0 / 3, gs_box=0.0, distill=(0.0 if split_scales else 0.25, 0.0 if split_scales else 0.25))]
        ref_g = torch.zeros(R, Cp, device=DEV)
        ref_l = torch.zeros(len(chunks), 4, device=DEV)
        for i, ((r0, r1), sp) in enumerate(zip(chunks, spec)):
            ops.box_loss(pred[r0:r1], Cp, K, r1 - r0, rois[r0:r1], cls[r0:r1], gtb[r0:r1], w4, sp["gs_cls"], sp["gs_box"], ref_g[r0:r1], ref_l[i, 0:2])
            if "distill" in sp:
                a, b = sp["distill"]
                kl = sp.get("kl", False)
                if a == b:
                    ops.roih_distill_loss(pred[r0:r1], tpred[r0:r1], Cp, K, r1 - r0, 2.0, kl, True, True, a, ref_g[r0:r1], ref_l[i, 2:4])
                else:
                    ops.roih_distill_loss(pred[r0:r1], tpred[r0:r1], Cp, K, r1 - r0, 2.0, kl, True, False, a, ref_g[r0:r1], ref_l[i, 2:4])
                    ops.roih_distill_loss(pred[r0:r1], tpred[r0:r1], Cp, K, r1 - r0, 2.0, kl, False, True, b, ref_g[r0:r1], ref_l[i, 2:4])
        ref_lo = ops.cast_from_f32(ref_g, torch.bfloat16)
        got_g = torch.zeros(R, Cp, device=DEV)
        got_lo = torch.full((R, Cp), float("nan"), dtype=torch.bfloat16, device=DEV)
        got_l = torch.zeros(len(chunks), 4, device=DEV)
        desc = []
        for i, ((r0, r1), sp) in enumerate(zip(chunks, spec)):
            q = dict(r0=r0, r1=r1, gs_cls=sp["gs_cls"], gs_box=sp["gs_box"], loss_box=got_l[i, 0:2])
            if "distill" in sp:
                q.update(t_pred=tpred[r0:r1], cls_T=2.0, kl=sp.get("kl", False), do_cls=True, do_reg=True, gs_dcls=sp["distill"][0], gs_dreg=sp["distill"][1],
                         loss_d=got_l[i, 2:4])
            desc.append(q)
        ops.box_losses_fused(pred, Cp, K, rois, cls, gtb, w4, desc, got_g, got_lo)
        torch.cuda.synchronize()
        assert torch.equal(ref_g, got_g), split_scales
        assert torch.equal(ref_lo, got_lo), split_scales
        assert float(ref_g.abs().sum()) > 0 and float((ref_l - got_l).abs().max()) <= 2e-6 * float(ref_l.abs().max()), (ref_l, got_l)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_upsample2_bwd_chain_equals_three_launches(dtype):
    """aldi_upsample2_bwd_chain against three accumulating aldi_upsample2_bwd calls (the FPN's top-down backward): the same bits on all three maps."""
    from aldi_amd import ops
    g = torch.Generator().manual_seed(13)
    N, H5, W5, C = 2, 7, 11, 256
    maps = [torch.randn(N, H5 * s_, W5 * s_, C, generator=g).to(DEV, dtype) for s_ in (8, 4, 2, 1)]
    ref = [m.clone() for m in maps]
    for i in range(3):
        ops.upsample2_bwd(ref[i], ref[i + 1], accumulate=True)
    got = [m.clone() for m in maps]
    ops.upsample2_bwd_chain(got[0], got[1], got[2], got[3])
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert not torch.equal(got[3], maps[3])


def test_rpn_and_box_losses_vs_oracle():
    from aldi_amd import ops
    from aldi_amd.engine import GMAX, ROI_WEIGHTS, make_anchors
    from oracle import d2_rcnn as d2
    g = torch.Generator().manual_seed(9)
    cfg = d2.make_cfg(num_classes=8)
    shapes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    N, A, C = 2, 3, 16
    anchors = make_anchors(shapes, DEV)
    oa = d2.generate_anchors(cfg, shapes)
    sumA = anchors.shape[0]
    heads = [torch.randn(N, h, w, C, generator=g) for h, w in shapes]
    lo = [t[..., :A].reshape(N, -1).clone().requires_grad_(True) for t in heads]
    de = [t[..., A:5 * A].reshape(N, -1, 4).clone().requires_grad_(True) for t in heads]
    gts = [_rand_boxes(4, 128, 96, g, lo=10, hi=80), _rand_boxes(2, 128, 96, g, lo=10, hi=80)]
    labels = torch.full((N, sumA), -1, dtype=torch.int32)
    matched = torch.zeros(N, sumA, dtype=torch.int32)
    mgt = []
    for n in range(N):
        mqm = d2.pairwise_iou(gts[n], torch.cat(oa))
        midx, lab = d2.matcher(mqm, cfg["rpn_iou_thresholds"], [0, -1, 1], True)
        pos = torch.nonzero(lab == 1).squeeze(1)[:40]
        neg = torch.nonzero(lab == 0).squeeze(1)[::7][:100]
        labels[n, pos], labels[n, neg] = 1, 0
        matched[n] = midx.to(torch.int32)
        mgt.append(gts[n][midx])
    ref = d2.rpn_losses(cfg, oa, lo, de, [labels[n].to(torch.int8) for n in range(N)], mgt)
    (2.0 * ref["loss_rpn_cls"] + 0.5 * ref["loss_rpn_loc"]).backward()
    gb = torch.zeros(N, GMAX, 4)
    cnt = torch.zeros(N, dtype=torch.int32)
    for i, b in enumerate(gts):
        gb[i, : len(b)], cnt[i] = b, len(b)
    hd = [t.to(DEV) for t in heads]
    gr = [torch.zeros_like(t) for t in hd]
    loss = torch.zeros(2, device=DEV)
    ops.rpn_loss(ops.make_geom(shapes, A, C), hd, gr, anchors, labels.to(DEV), matched.to(DEV), gb.to(DEV), cnt.to(DEV), GMAX, N,
                 1.0 / (256 * N), 2.0, 0.5, loss)
    assert abs(float(loss[0]) - float(ref["loss_rpn_cls"])) < 1e-5 and abs(float(loss[1]) - float(ref["loss_rpn_loc"])) < 1e-5
    for l, (h, w) in enumerate(shapes):
        gl = gr[l].cpu()
        assert (gl[..., :A].reshape(N, -1) - lo[l].grad).abs().max() < 1e-7
        assert (gl[..., A:5 * A].reshape(N, -1, 4) - de[l].grad).abs().max() < 1e-7
    # ---- box losses
    K, R = 8, 300
    Cp = 48
    pred = torch.randn(R, Cp, generator=g)
    scores = pred[:, : K + 1].clone().requires_grad_(True)
    deltas = pred[:, K + 1: K + 1 + 4 * K].clone().requires_grad_(True)
    pb = _rand_boxes(R, 300, 200, g)
    gtb = _rand_boxes(R, 300, 200, g)
    cls = torch.randint(0, K + 1, (R,), generator=g)
    sampled = [{"gt_classes": cls, "proposal_boxes": pb, "gt_boxes": gtb}]
    refb = d2.roi_losses(cfg, scores, deltas, sampled)
    (refb["loss_cls"] * 0.5 + refb["loss_box_reg"] * 3.0).backward()
    rois = torch.cat([torch.zeros(R, 1), pb], 1).contiguous().to(DEV)
    grad = torch.zeros(R, Cp, device=DEV)
    loss = torch.zeros(2, device=DEV)
    ops.box_loss(pred.to(DEV), Cp, K, R, rois, cls.to(torch.int32).to(DEV), gtb.to(DEV), ROI_WEIGHTS, 0.5, 3.0, grad, loss)
    assert abs(float(loss[0]) - float(refb["loss_cls"])) < 2e-6 * float(refb["loss_cls"]) + 1e-6
    assert abs(float(loss[1]) - float(refb["loss_box_reg"])) < 2e-5 * float(refb["loss_box_reg"]) + 1e-6
    assert (grad[:, : K + 1].cpu() - scores.grad).abs().max() < 1e-7
    assert (grad[:, K + 1: K + 1 + 4 * K].cpu() - deltas.grad).abs().max() < 1e-7


def test_detections_and_pseudolabel_filter_vs_oracle():
    from aldi_amd import ops
    from aldi_amd.engine import ROI_WEIGHTS
    from oracle import aldi_ops as ao
    from oracle import d2_rcnn as d2
    g = torch.Generator().manual_seed(11)
    cfg = d2.make_cfg(num_classes=8)
    K, Cp, N, P = 8, 48, 2, 1000
    sizes = [(200, 300), (190, 280)]
    props = torch.stack([_rand_boxes(P, 300, 200, g, lo=8, hi=150) for _ in range(N)])
    pcount = torch.tensor([1000, 640], dtype=torch.int32)
    pred = torch.randn(N * P, Cp, generator=g)
    pred[:, : K + 1] *= 3.0
    pred[:, K + 1:] *= 0.5
    proposals = [{"proposal_boxes": props[n, : int(pcount[n])], "image_size": sizes[n]} for n in range(N)]
    sel = torch.cat([torch.arange(n * P, n * P + int(pcount[n])) for n in range(N)])
    ref = d2.fast_rcnn_inference(cfg, pred[sel, : K + 1], pred[sel, K + 1: K + 1 + 4 * K], proposals)
    ws = torch.empty(ops.detections_workspace(N), dtype=torch.uint8, device=DEV)
    out = {k: torch.empty((N, 100) + s, dtype=d, device=DEV) for k, s, d in (("db", (4,), torch.float32), ("ds", (), torch.float32), ("dc", (), torch.int32),
                                                                            ("pb", (4,), torch.float32), ("pc", (), torch.int32), ("ps", (), torch.float32))}
    dcount = torch.empty(N, dtype=torch.int32, device=DEV)
    pl_count = torch.empty(N, dtype=torch.int32, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    thr = 0.995
    ops.detections(pred.to(DEV), Cp, K, props.to(DEV), pcount.to(DEV), P, N, torch.tensor(sizes, dtype=torch.int32, device=DEV), ROI_WEIGHTS,
                   0.05, 0.5, 100, thr, ws, out["db"], out["ds"], out["dc"], dcount, out["pb"], out["pc"], out["ps"], pl_count, err)
    assert int(err) == 0
    for n in range(N):
        k = int(dcount[n])
        assert k == len(ref[n]["scores"]) == 100
        assert (out["ds"][n, :k].cpu() - ref[n]["scores"]).abs().max() < 1e-6
        assert torch.equal(out["dc"][n, :k].cpu().long(), ref[n]["pred_classes"])
        assert (out["db"][n, :k].cpu() - ref[n]["pred_boxes"]).abs().max() < 1e-3
        pl = ao.process_bbox(ref[n], thr)
        m = int(pl_count[n])
        assert m == len(pl["scores"]) and 0 < m < 100
        assert torch.equal(out["pc"][n, :m].cpu().long(), pl["gt_classes"])
        assert (out["pb"][n, :m].cpu() - pl["gt_boxes"]).abs().max() < 1e-3
        assert bool((out["ps"][n, :m] > thr).all())


@pytest.mark.parametrize("sizes", [[(64, 96), (64, 96)], [(61, 93), (37, 50)], [(250, 333)], [(800, 1333), (800, 1333)]])
def test_fused_stem_pool_equals_stem_then_maxpool(sizes):
    """stem conv + FrozenBN + ReLU + max_pool2d(3,2,1) in one kernel == the two kernels, bit for bit (ragged images in one padded batch,
    odd conv-map sizes, the benchmark size)"""
    from aldi_amd import ops
    from aldi_amd.arch import pad_to
    gen = torch.Generator().manual_seed(len(sizes) + sizes[0][0])
    Hs, Ws = pad_to(max(s[0] for s in sizes), 32), pad_to(max(s[1] for s in sizes), 32)
    img = torch.zeros(len(sizes), 3, Hs, Ws, dtype=torch.uint8)
    for i, (h, w) in enumerate(sizes):
        img[i, :, :h, :w] = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
    w = (torch.randn(64, 7, 7, 3, generator=gen) * 0.05).to(DEV)
    scale = (0.5 + torch.rand(64, generator=gen)).to(DEV)
    shift = (torch.randn(64, generator=gen) * 0.3).to(DEV)
    mean, std = (103.53, 116.28, 123.675), (1.0, 1.0, 1.0)
    imgd = img.to(DEV)
    ref = ops.maxpool3s2(ops.stem_forward(imgd, sizes, w, scale, shift, mean, std, torch.bfloat16))
    got = ops.stem_pool_forward(imgd, sizes, ops.stem_pack_weights(w), scale, shift, mean, std)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and float(ref.float().abs().max()) > 0
    assert torch.equal(got, ref)


def test_stage_images_equals_per_image_copies():
    """aldi_stage_images: images of different sizes (odd widths: unaligned rows) into one padded batch; padding untouched"""
    from aldi_amd import ops
    g = torch.Generator().manual_seed(0)
    sizes = [(37, 61), (40, 64), (1, 5), (33, 63)]
    imgs = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).to(DEV) for h, w in sizes]
    batch = torch.full((len(imgs), 3, 64, 64), 7, dtype=torch.uint8, device=DEV)
    ref = batch.clone()
    for i, im in enumerate(imgs):
        ref[i, :, : im.shape[1], : im.shape[2]] = im
    ops.stage_images(imgs, batch)
    torch.cuda.synchronize()
    assert torch.equal(batch, ref)


def test_ema_update_writes_the_compute_copy_in_the_same_pass():
    """aldi_ema_update with a bf16 compute copy of the first n_compute elements == the update followed by a cast; unaligned
    tails and the copy-only form included"""
    from aldi_amd import ops
    g = torch.Generator().manual_seed(5)
    n, nc = 100003, 70001
    for copy_only in (False, True):
        t = torch.randn(n, generator=g).to(DEV)
        s = torch.randn(n, generator=g).to(DEV)
        ref = t.clone()
        ops.ema_update(ref, s, None, n, 0.9996, copy_only, torch.float32)
        comp = torch.full((nc,), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.ema_update(t, s, comp, n, 0.9996, copy_only, torch.bfloat16, n_compute=nc)
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
        assert torch.equal(comp, ops.cast_from_f32(ref[:nc], torch.bfloat16))


@pytest.mark.parametrize("shape", [(2, 196, 332, 256), (1, 9, 11, 256), (3, 40, 50, 64)])
def test_avgpool_large_and_small_maps(shape):
    """AdaptiveAvgPool2d(1) of the image discriminator (aldi/align.py:113): the split form (large maps) and the one-pass form"""
    from aldi_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g).to(DEV, torch.bfloat16)
    y = ops.avgpool(x)
    ref = x.float().mean(dim=(1, 2)).view(shape[0], 1, 1, shape[3])
    assert (y.float() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
    xf = x.float()
    yf = ops.avgpool(xf)
    assert (yf - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("G", [0, 1, 7, 100, "ragged"])
def test_anchor_matcher_per_wave_lists_equal_per_workgroup_lists(G):
    """match_wave 1 (default, r06: every wave culls the GT list against ITS 64 anchors' union box) vs 0 (one list per 1024-anchor workgroup): labels,
    matched indices and IoUs bit for bit, at the benchmark's anchor count, incl. images without GT, a GT nothing overlaps (the low-quality rule's
    'claims every box' case) and different counts per image"""
    from aldi_amd import _lib as L
    from aldi_amd import ops
    from aldi_amd.engine import GMAX, make_anchors
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    anchors = make_anchors(shapes, DEV)
    N, sumA = 4, anchors.shape[0]
    g = torch.Generator().manual_seed(5)
    counts = [100, 0, 13, GMAX] if G == "ragged" else [G] * N
    gb = torch.zeros(N, GMAX, 4)
    for i, k in enumerate(counts):
        w = torch.rand(k, generator=g) * 200 + 4; h = torch.rand(k, generator=g) * 150 + 4
        x = torch.rand(k, generator=g) * 1300; y = torch.rand(k, generator=g) * 780
        gb[i, :k] = torch.stack([x, y, x + w, y + h], 1)
    if G == "ragged":
        gb[0, 3] = torch.tensor([5000.0, 5000.0, 5010.0, 5010.0])        # overlaps no anchor: best IoU 0
    gbd, cntd = gb.to(DEV), torch.tensor(counts, dtype=torch.int32, device=DEV)
    out = {}
    for mode in (2, 1, 0):                   # 2: per-wave lists + one 128-byte line per GT in the scratch (the engine's form), 1: per-wave lists, packed scratch
        L.reset_tuning(); L.set_tuning("match_wave", min(mode, 1))
        best_iou = torch.empty(N, sumA, device=DEV); best_idx = torch.empty(N, sumA, dtype=torch.int32, device=DEV)
        labels = torch.empty(N, sumA, dtype=torch.int32, device=DEV)
        scratch = ops.box_match_scratch(N, GMAX, DEV) if mode == 2 else torch.empty(N, GMAX, dtype=torch.int32, device=DEV)
        ops.box_match(anchors, 0, None, sumA, gbd, cntd, GMAX, N, 0.3, 0.7, True, best_iou, best_idx, scratch, labels)
        torch.cuda.synchronize()
        out[mode] = (best_iou, best_idx, labels, scratch.view(N, GMAX, -1)[:, :, 0].contiguous())
    L.reset_tuning()
    for m in (2, 1):
        for a, b in zip(out[m], out[0]):
            assert torch.equal(a, b)
    if G != 0:
        assert int((out[1][2] == 1).sum()) > 0
