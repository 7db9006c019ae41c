"""BASELINE.json sizes (1333x800, padded 1344x800, 268 569 anchors/img): size-independent properties
of the HIP path where the oracle would take minutes."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 8


@pytest.fixture(scope="module")
def full_ctx():
    from aldi_amd import synthetic as syn
    from aldi_amd.arch import ParamLayout
    from aldi_amd.engine import RCNN, Weights
    lay = ParamLayout(K)
    w = Weights(lay, torch.device("cuda"), torch.bfloat16, trainable=True)
    w.load_state_dict(syn.init_state_dict(K, seed=1))
    m = RCNN(w, K)
    _, data, uw, us = syn.make_batch(2, 2, 800, 1333, K, seed=3)
    torch.manual_seed(0)
    c = m.forward_train([d["image"] for d in data], [d["instances"] for d in data], roi_seed=1)
    t = m.inference([d["image"] for d in uw], 0.8)
    torch.cuda.synchronize()
    return m, c, t, data


def test_geometry_at_benchmark_size(full_ctx):
    m, c, t, _ = full_ctx
    assert [tuple(p.shape[1:3]) for p in c.P] == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert c.anchors.shape[0] == 268569 and c.rpn_labels.shape == (2, 268569)
    assert int(m.err) == 0


def test_sampling_invariants(full_ctx):
    m, c, t, data = full_ctx
    lab = c.rpn_labels
    for n in range(2):
        npos, nneg = int((lab[n] == 1).sum()), int((lab[n] == 0).sum())
        assert npos <= 128 and npos + nneg == 256                      # 256 per image, at most half positive
    assert c.R == 1024 and c.rows == [512, 512]
    for n in range(2):
        cls = c.r_cls[n * 512:(n + 1) * 512]
        assert int((cls < K).sum()) <= 128                              # <= 25% foreground
        assert bool(((cls >= 0) & (cls <= K)).all())
    # sampled ROIs are proposals or appended GT of the right image
    assert bool((c.rois[:512, 0] == 0).all() and (c.rois[512:, 0] == 1).all())


def test_proposal_invariants(full_ctx):
    m, c, t, _ = full_ctx
    for ctx, in ((c,), (t,)):
        for n in range(2):
            k = int(ctx.prop_count[n])
            assert 0 < k <= 1000
            b = ctx.props[n, :k]
            assert bool((b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] <= 1333).all() and (b[:, 3] <= 800).all())
            assert bool(((b[:, 2] - b[:, 0]) > 0).all() and ((b[:, 3] - b[:, 1]) > 0).all())
    s = c.prop_scores[0, : int(c.prop_count[0])]
    assert bool((s[1:] <= s[:-1]).all())                               # sorted by objectness


def test_detection_and_pseudolabel_invariants(full_ctx):
    m, c, t, _ = full_ctx
    for n in range(2):
        k = int(t.det.count[n])
        assert 0 < k <= 100
        sc = t.det.scores[n, :k]
        assert bool((sc[1:] <= sc[:-1]).all()) and bool((sc > 0.05).all())
        m_ = int(t.pseudo["count"][n])
        assert m_ == int((sc > 0.8).sum())                              # strict > threshold, order preserved
        assert torch.equal(t.pseudo["scores"][n, :m_], sc[sc > 0.8])
        # per-class NMS is idempotent: no two kept same-class detections overlap by more than 0.5
        b, cl = t.det.boxes[n, :k], t.det.classes[n, :k]
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(b[:, None, :2], b[None, :, :2])
        rb = torch.min(b[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(2)
        iou = inter / (area[:, None] + area[None] - inter)
        same = (cl[:, None] == cl[None]) & ~torch.eye(k, dtype=torch.bool, device=b.device)
        assert float((iou * same).max()) <= 0.5 + 1e-6


def test_gradient_linearity_in_loss_scale(full_ctx):
    """backward is linear in the loss coefficients (what makes 1 all-reduce/step == the reference's 2-3)."""
    m, c, t, _ = full_ctx
    keys = ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")
    m.wts.zero_grad()
    m.backward(c, {k: 0.5 for k in keys})
    torch.cuda.synchronize()
    g1 = m.wts.grad.clone()
    m.wts.zero_grad()
    m.backward(c, {k: 1.0 for k in keys})
    torch.cuda.synchronize()
    g2 = m.wts.grad.clone()
    assert torch.isfinite(g2).all() and float(g2.abs().max()) > 0
    rel = (g2 - 2 * g1).abs().max() / g2.abs().max()
    assert float(rel) < 2e-2                                            # bf16 rounding of the scaled gradients only


def test_ema_is_a_convex_combination_at_full_size():
    from aldi_amd import ops
    n = 41_400_000
    s = torch.randn(n, device="cuda")
    t = torch.randn(n, device="cuda")
    t0 = t.clone()
    ops.ema_update(t, s, None, n, 0.9996, False, torch.float32)
    lo, hi = torch.minimum(s, t0), torch.maximum(s, t0)
    assert bool(((t >= lo - 1e-6) & (t <= hi + 1e-6)).all())
    assert float((t - (s * (1 - 0.9996) + t0 * 0.9996)).abs().max()) < 1e-6
