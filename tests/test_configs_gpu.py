"""BASELINE.json configs[0] (Base-RCNN-FPN.yaml, R50 source-only, K = 80, 800x800) end to end, and the
benchmark configuration (configs[1], 1333x800, 2 + 2 images) in the benchmark dtype against the same step in
the fp32 parity mode (VERDICT r01 missing #3, next #1).

Reference path: aldi/trainer.py:28-117 with DATASETS.BATCH_CONTENTS = ("labeled_weak",) (the default of
aldi/config.py:12, not overridden by configs/Base-RCNN-FPN.yaml), EMA / distillation / alignment off."""
import os
import random
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def oracle_lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def _cfg1(h, w, bf16, lr=0.002):
    from aldi_amd.config import add_aldi_config, get_cfg
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "Base-RCNN-FPN.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 2, "SOLVER.AMP.ENABLED", bf16, "SOLVER.BASE_LR", lr, "SOLVER.WARMUP_ITERS", 0, "SEED", 1,
                         "SYNTHETIC.HEIGHT", h, "SYNTHETIC.WIDTH", w])
    return cfg


def test_cfg1_is_source_only_k80():
    cfg = _cfg1(800, 800, True)
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 80 and tuple(cfg.DATASETS.BATCH_CONTENTS) == ("labeled_weak",)
    assert not cfg.EMA.ENABLED and not cfg.DOMAIN_ADAPT.TEACHER.ENABLED


def test_cfg1_iterations_vs_oracle_fp32():
    """two source-only iterations (K = 80) at a size the oracle finishes in seconds: loss dict (keys, order, values to
    1e-3), sampled ROI indices bit-exact, SGD update."""
    from aldi_amd import synthetic as syn
    from aldi_amd.trainer import ALDITrainer
    from oracle import aldi_ops as ao
    from oracle import d2_rcnn as d2
    K, H, W = 80, 160, 160
    cfg = _cfg1(H, W, False)
    random.seed(0)
    torch.manual_seed(21)
    tr = ALDITrainer(cfg)
    assert tr.ema is None or not cfg.EMA.ENABLED
    lay = tr.model.layout
    off = dict(do_hard_cls=False, do_hard_obj=False, do_hard_rpn_reg=False, do_hard_roi_reg=False, do_cls_dst=False, do_obj_dst=False,
               do_rpn_reg_dst=False, do_roih_reg_dst=False, cls_temperature=1.0, obj_temperature=1.0, cls_loss_type="CE")
    orc = ao.OracleALDI(d2.make_cfg(num_classes=K), syn.init_state_dict(K, seed=1), lr=0.002, ims_per_gpu=2, backward_at_end=False, py_seed=0,
                        distill=off)
    loader = iter(ALDITrainer.build_train_loader(cfg))
    hip_idx, hip_props = [], []
    mfwd = type(tr.model).forward

    def fwd(self, *a, **kw):
        out = mfwd(self, *a, **kw)
        c_ = self._last.ctx
        hip_idx.append(c_.r_idx[: c_.R].cpu())
        hip_props.append([{"proposal_boxes": c_.props[i, :n].cpu(), "objectness_logits": c_.prop_scores[i, :n].cpu(), "image_size": c_.sizes[i]}
                          for i, n in enumerate(c_.prop_count.tolist())])
        return out
    type(tr.model).forward = fwd
    try:
        for it in range(2):
            hip_idx.clear()
            hip_props.clear()
            s_sd = tr.model.state_dict()
            mflat = torch.zeros(lay.n_total)
            mflat[: lay.n_train] = tr.model.weights.mom.cpu()
            mom = lay.unpack(mflat)
            rng = torch.get_rng_state()
            tr.iter = it
            tr.before_step()
            tr.run_step()
            tr.after_step()
            torch.cuda.synchronize()
            hip = {k: float(v) for k, v in tr._trainer.last_loss_dict.items()}
            orc.sd = {k: v.clone() for k, v in s_sd.items()}
            for k in orc.train_keys:
                orc.sd[k].requires_grad_(True)
            orc.bufs = {k: mom[k].clone() for k in orc.train_keys} if it > 0 else {}
            orc.iter = it
            orc.proposal_override = [list(p_) for p_ in hip_props]
            torch.set_rng_state(rng)
            ref = orc.step(*next(loader))
            assert list(ref.keys()) == list(hip.keys()) == ["loss_cls_source_weak", "loss_box_reg_source_weak", "loss_rpn_cls_source_weak",
                                                            "loss_rpn_loc_source_weak"]
            oidx = torch.cat([s_["sampled_idxs"] for s_ in orc.last["student_cap"]["sampled"]]).to(torch.int32)
            assert torch.equal(hip_idx[0], oidx)
            for k in ref:
                assert abs(ref[k] - hip[k]) < 1e-3 * max(1.0, abs(ref[k])), (it, k, ref[k], hip[k])
            hs = tr.model.state_dict()
            for k in ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.weight", "backbone.fpn_output2.weight",
                      "backbone.bottom_up.res4.2.conv1.weight"):
                upd = (orc.sd[k].detach() - s_sd[k]).abs().max()
                assert (hs[k] - orc.sd[k].detach()).abs().max() <= 2e-2 * float(upd) + 1e-9, (it, k)
    finally:
        type(tr.model).forward = mfwd
    assert int(tr.model.engine.err) == 0


def test_cfg1_fullsize_bf16_properties():
    """800x800, K = 80, 2 images, benchmark dtype: three iterations through the default dispatch; size-independent properties."""
    from aldi_amd.trainer import ALDITrainer
    cfg = _cfg1(800, 800, True, lr=1e-4)
    random.seed(0)
    torch.manual_seed(2)
    tr = ALDITrainer(cfg)
    w0 = tr.model.weights.master.clone()
    for it in range(3):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
    torch.cuda.synchronize()
    c = tr.model._last.ctx
    assert [tuple(p.shape[1:3]) for p in c.P] == [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    assert c.anchors.shape[0] == 159882                                     # anchors per image of cfg 1
    assert c.R == 1024 and c.rows == [512, 512]
    lab = c.rpn_labels
    for n in range(2):
        npos, nneg = int((lab[n] == 1).sum()), int((lab[n] == 0).sum())
        assert npos <= 128 and npos + nneg == 256
        cls = c.r_cls[n * 512:(n + 1) * 512]
        assert int((cls < 80).sum()) <= 128 and bool(((cls >= 0) & (cls <= 80)).all())
    ld = {k: float(v) for k, v in tr._trainer.last_loss_dict.items()}
    assert all(v == v and 0.0 <= v < 1e3 for v in ld.values()), ld
    assert 0.0 < ld["loss_cls_source_weak"] < 8.0 and ld["loss_rpn_cls_source_weak"] > 0.0
    lay = tr.model.layout
    w1 = tr.model.weights.master
    assert torch.isfinite(w1).all()
    assert not torch.equal(w1[: lay.n_train], w0[: lay.n_train]) and torch.equal(w1[lay.n_train:], w0[lay.n_train:])   # frozen part untouched
    assert int(tr.model.engine.err) == 0


def _bench_trainer(bf16, fused=True):
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "SOLVER.BASE_LR", 1e-4,
                         "SOLVER.AMP.ENABLED", bf16])
    cfg.SOLVER.FUSED_STEP = fused
    cfg.SOLVER.STEP_GRAPH = False
    random.seed(1234)
    torch.manual_seed(100)
    return ALDITrainer(cfg)


def test_benchmark_step_bf16_vs_fp32_parity_mode():
    """The benchmark step itself (configs[1]: 1333x800, 2 labeled + 2 unlabeled, fused schedule, default dispatch = the
    256x128 halo igemm, the long-K tiles, the direct 1x1 epilogue and the 256x256 wgrad) in bf16 -- the dtype bench.py's number is
    measured in -- against the SAME step in the fp32 parity mode, whose kernels are the ones held to north_star's 1e-3 on losses against
    the oracle (tests/test_engine_gpu.py, fp32).  Identical weights, inputs and RNG stream, and the bf16 run is GIVEN the fp32 run's
    inputs of the discrete stages (the student's proposals, the teacher's pseudo labels: `FusedStep.discrete_inputs_hook`), because
    matching / sampling are discontinuous in them: with those equal every sampled anchor and ROI index is identical and what remains
    is rounding, so the bounds are rounding-level:
      * sampled anchor labels and ROI indices of all four images bit-identical;
      * every loss within 2 % (bf16 activations through ~60 layers);
      * the weight gradients of every layer group: cosine >= 0.99 and relative L2 error <= 3e-2."""
    from aldi_amd import synthetic as syn
    out, rec = {}, {}

    def record(S, c, tc):
        rec.update(props=c.props.clone(), scores=c.prop_scores.clone(), count=c.prop_count.clone(), gt={k: v.clone() for k, v in c.gt.items()})

    def inject(S, c, tc):
        c.props.copy_(rec["props"]); c.prop_scores.copy_(rec["scores"]); c.prop_count.copy_(rec["count"])
        for k, v in c.gt.items():
            v.copy_(rec["gt"][k])
    for name, bf16 in (("fp32", False), ("bf16", True)):
        tr = _bench_trainer(bf16)
        data = syn.make_batch(2, 2, 800, 1333, 8, seed=100)
        random.seed(77)
        torch.manual_seed(5)
        tr.iter = 0
        tr.before_step()
        t = tr._trainer
        t.optimizer.zero_grad()
        from aldi_amd.fused_step import FusedStep
        t._fused_step = FusedStep(t)
        t._fused_step.discrete_inputs_hook = inject if bf16 else record
        ld = t.run_model(tuple(None if p is None else [dict(d) for d in p] for p in data))
        torch.cuda.synchronize()
        assert t._fused_done
        c = tr.model._last_fused
        out[name] = dict(losses={k: float(v) for k, v in ld.items()}, labels=c.rpn_labels.cpu(), R=c.R, rows=list(c.rows), r_idx=c.r_idx[: c.R].cpu(),
                         grad=tr.model.weights.grad.clone(), layout=tr.model.layout,
                         pl=tr.ema.model._last_inference.pseudo["count"].tolist(), err=int(tr.model.engine.err) | int(tr.ema.model.engine.err))
        del tr
        torch.cuda.empty_cache()
    a, b = out["fp32"], out["bf16"]
    assert a["err"] == 0 and b["err"] == 0
    assert torch.equal(a["labels"], b["labels"])                          # all four images: bit-identical anchor sampling
    assert a["R"] == b["R"] == 2048 and a["rows"] == b["rows"] and torch.equal(a["r_idx"], b["r_idx"])
    assert list(a["losses"]) == list(b["losses"])
    for k in a["losses"]:
        x, y = a["losses"][k], b["losses"][k]
        print("loss bf16 vs fp32:", k, x, y)
        assert abs(x - y) <= 0.02 * max(abs(x), 0.05), (k, x, y)
    assert all(abs(p - q) <= 2 for p, q in zip(a["pl"], b["pl"])), (a["pl"], b["pl"])     # (the teacher's OWN bf16 pseudo-label counts)
    lay = a["layout"]
    groups = {"box head": ["roi_heads.box_head.fc1", "roi_heads.box_head.fc2", "box_pred"],
              "rpn + fpn": ["proposal_generator.rpn_head.conv", "rpn_head_out"] + [f"backbone.fpn_output{l}" for l in (2, 3, 4, 5)] +
                           [f"backbone.fpn_lateral{l}" for l in (2, 3, 4, 5)]}
    for st in (3, 4, 5):
        groups[f"res{st}"] = [n for n in lay.t if f".res{st}." in n]
    worst = []
    for gname, names in groups.items():
        ga = torch.cat([a["grad"][lo:hi] for lo, hi in lay.ranges(names)])
        gb = torch.cat([b["grad"][lo:hi] for lo, hi in lay.ranges(names)])
        assert torch.isfinite(gb).all()
        cos = float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30))
        rel = float((ga - gb).norm() / (ga.norm() + 1e-30))
        print("grad bf16 vs fp32:", gname, "cosine", round(cos, 5), "rel-L2", round(rel, 5))
        worst.append((gname, cos, rel))
    assert all(cos >= 0.99 and rel <= 3e-2 for _, cos, rel in worst), worst


def test_cfg2_fullsize_alignment_bf16_properties():
    """BASELINE configs[2]'s per-GPU workload at full size: ALDI++ with image- and instance-level alignment on, 1333x800, 2 labeled +
    2 unlabeled images, benchmark dtype, fused + graph-replayed schedule (three eager steps, the capture, replays).  Size-independent
    properties: loss-dict keys of all three rows incl. the discriminator losses (reference aldi/trainer.py:99-111: target_weak keeps
    only "_da_" keys; aldi/align.py:84-100), per-micro-step normalisers, the gradient reaches both discriminators and -- reversed --
    the trunk, the teacher stays an EMA of the student, frozen layers do not move."""
    from aldi_amd import synthetic as syn
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "SOLVER.BASE_LR", 1e-4,
                         "SOLVER.AMP.ENABLED", True, "DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED", True, "DOMAIN_ADAPT.ALIGN.INS_DA_ENABLED", True])
    cfg.SOLVER.FUSED_STEP = True
    cfg.SOLVER.STEP_GRAPH = True
    random.seed(1234)
    torch.manual_seed(100)
    tr = ALDITrainer(cfg)
    t = tr._trainer
    w0 = tr.model.weights.master.clone()
    for it in range(6):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        assert t._fused_done
    torch.cuda.synchronize()
    st = t._fused_step.stats
    assert st["replays_a"] >= 2 and st["replays_b"] >= 2, st
    ld = {k: float(v) for k, v in t.last_loss_dict.items()}
    src = [f"{k}_source_strong" for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_da_img", "loss_da_ins")]
    tgt = ["loss_da_img_target_weak", "loss_da_ins_target_weak"]
    dst = [f"{k}_distill" for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "_da", "loss_obj_bce", "loss_rpn_l1", "loss_cls_ce", "loss_roih_l1")]
    assert list(ld) == src + tgt + dst, list(ld)
    assert all(v == v and 0.0 <= v < 1e3 for v in ld.values()), ld
    # a discriminator that cannot tell the domains yet: BCE of a near-zero logit = ln 2, times the 0.01 loss weight, over accum = 3
    for k in ("loss_da_img_source_strong", "loss_da_ins_source_strong", "loss_da_img_target_weak", "loss_da_ins_target_weak"):
        assert 0.0005 < ld[k] < 0.02, (k, ld[k])
    assert ld["_da_distill"] == 0.0 and ld["loss_cls_distill"] == 0.0        # masked hard losses are reported as zeros (aldi/distill.py:181-186)
    c = tr.model._last_fused
    assert c.N == 6 and [ch["name"] for ch in c.chunks] == ["source_strong", "target_weak", "distill"]
    assert c.R == 3 * 1024 and list(c.rows) == [512] * 6
    assert [tuple(p.shape[1:3]) for p in c.P] == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    lay = tr.model.layout
    g = tr.model.weights.grad
    for name in [n for n in lay.t if n.startswith(("img_align", "ins_align"))] + ["backbone.fpn_output2", "backbone.bottom_up.res3.0.conv2", "roi_heads.box_head.fc2"]:
        for lo, hi in lay.ranges([name]):
            assert torch.isfinite(g[lo:hi]).all() and float(g[lo:hi].abs().max()) > 0, name
    w1 = tr.model.weights.master
    assert torch.isfinite(w1).all()
    assert not torch.equal(w1[: lay.n_train], w0[: lay.n_train]) and torch.equal(w1[lay.n_train:], w0[lay.n_train:])
    # the teacher (incl. its own, unused discriminators: SURVEY B.9) trails the student: |teacher - initial| < |student - initial|
    tw = tr.ema.model.weights.master
    n = lay.n_train
    ds, dt = (w1[:n] - w0[:n]).norm(), (tw[:n] - w0[:n]).norm()
    assert 0 < float(dt) < float(ds)
    assert int(tr.model.engine.err) == 0 and int(tr.ema.model.engine.err) == 0
    pl = tr.ema.model._last_inference.pseudo["count"].tolist()
    assert len(pl) == 2 and all(0 <= v <= 100 for v in pl)


def test_cfg4_fullsize_deformable_detr_properties():
    """BASELINE configs[4] at full size, as `bench.py --workload detr` runs it: configs/cityscapes/ALDI-Best-DETR-Cityscapes.yaml (6 + 6 layers,
    300 queries, dropout 0.1, fp32), 1333 x 800, 2 labeled + 2 unlabeled images, HardDistiller through ALDITrainer.  Size-independent
    properties: the loss-dict keys of the source and the pseudo-labelled target pass for every decoder layer (reference configs/Base-DETR.yaml
    AUX_LOSS; aldi/distill.py:62-84 keeps the detector's own keys), finite values, the teacher's pseudo labels are consumed, trunk and
    transformer step while the frozen stem / res2 do not, the teacher trails the student with `query_embed` copied (aldi/ema.py:17,39-41),
    and the encoder's backward went through the gathered value gradient."""
    from aldi_amd import _lib as L
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-DETR-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SOLVER.IMS_PER_GPU", 2, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333,
                         "DOMAIN_ADAPT.TEACHER.THRESHOLD", 0.011, "SOLVER.WARMUP_ITERS", 0])
    assert cfg.MODEL.DEFORMABLE_DETR.TRANSFORMER.DROPOUT == 0.1 and cfg.MODEL.DEFORMABLE_DETR.TRANSFORMER.NUM_QUERIES == 300
    random.seed(1234)
    torch.manual_seed(100)
    tr = ALDITrainer(cfg)
    W, T = tr.model.weights, tr.ema.model.weights
    w0 = W.master.clone()
    seen = set()
    orig = L.call

    def spy(name, *a):
        seen.add(name)
        return orig(name, *a)
    L.call = spy
    try:
        for it in range(2):
            tr.iter = it
            tr.before_step()
            at_ema = W.master.clone()
            tr.run_step()
            tr.after_step()
    finally:
        L.call = orig
    torch.cuda.synchronize()
    assert "aldi_ms_deform_attn_backward_self" in seen and "aldi_dropout_add" in seen and "aldi_detr_set_loss" in seen
    ld = {k: float(v) for k, v in tr._trainer.last_loss_dict.items()}
    base = [f"{k}_{i}" for i in range(5) for k in ("loss_ce", "loss_bbox", "loss_giou")] + ["loss_ce", "loss_bbox", "loss_giou"]
    assert set(ld) == {f"{k}_source_strong" for k in base} | {f"{k}_distill" for k in base}, sorted(ld)
    assert all(v == v and 0.0 <= v < 1e3 for v in ld.values()), ld
    assert int(tr.ema.model._last_inference.pseudo["count"].sum()) > 0
    moved = (W.master - w0).abs()
    nb = W.nb
    assert torch.isfinite(W.master).all() and float(moved[nb:].max()) > 0 and float(moved[:nb].max()) > 0
    frozen = W.backbone.layout.ranges(["backbone.bottom_up.stem.conv1", "backbone.bottom_up.res2.0.conv1"])
    assert all(float(moved[a:b].max()) == 0.0 for a, b in frozen)
    (a, b), = W.ranges(["query_embed.weight"])
    assert torch.equal(T.master[a:b], at_ema[a:b])
    (a, b), = W.ranges(["class_embed.weight"])
    assert not torch.equal(T.master[a:b], at_ema[a:b])


def test_cfg3_fullsize_vitdet_b_properties():
    """BASELINE configs[3] at FULL size: ALDI++ on the ViTDet-B detector (Base-RCNN-VitDetB.yaml: ViT-B/16 -- 768 wide, 12 blocks, 12 heads, 14 x 14
    windows with four global blocks -- SimpleFeaturePyramid, AdamW with the layer-wise lr decay 0.7 the reference turns on for this backbone,
    aldi/backbone.py:38-43,66-84, aldi/trainer.py:200-209), per-GPU workload 1 labeled + 1 unlabeled 1333 x 800 image, bf16, fused student pass.
    Size-independent properties over three iterations: the reference's loss keys, finite values, pyramid / token geometry, every trainable
    tensor moves (nothing is frozen in this detector), deeper blocks move more than shallow ones under the layer decay, the teacher trails
    the student, pseudo labels are produced."""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer, EngineAdamW
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-VitDetB-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 2, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "DOMAIN_ADAPT.TEACHER.THRESHOLD", 0.05,
                         "SOLVER.WARMUP_ITERS", 0])
    assert cfg.MODEL.BACKBONE.NAME == "build_vitdet_b_backbone" and cfg.SOLVER.OPTIMIZER == "ADAMW" and cfg.SOLVER.IMS_PER_GPU == 1
    random.seed(1234)
    torch.manual_seed(100)
    tr = ALDITrainer(cfg)
    opt = tr._trainer.optimizer
    assert tr.model.vitdet and isinstance(opt, EngineAdamW) and opt.lr_decay_rate == 0.7 and opt.num_layers == 12
    W, T = tr.model.weights, tr.ema.model.weights
    sd0 = {k: v.clone() for k, v in tr.model.state_dict().items()}
    assert sd0["backbone.net.blocks.11.attn.qkv.weight"].shape == (3 * 768, 768) and sd0["backbone.net.patch_embed.proj.weight"].shape == (768, 3, 16, 16)
    assert sum(1 for k in sd0 if k.startswith("backbone.net.blocks.") and k.endswith("attn.qkv.weight")) == 12
    w0 = W.master.clone()
    for it in range(3):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        assert tr._trainer._fused_done
    torch.cuda.synchronize()
    ld = {k: float(v) for k, v in tr._trainer.last_loss_dict.items()}
    src = [f"{k}_source_strong" for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")]
    dst = [f"{k}_distill" for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_obj_bce", "loss_rpn_l1", "loss_cls_ce", "loss_roih_l1")]
    assert set(src + dst) <= set(ld), sorted(ld)
    assert all(v == v and 0.0 <= v < 1e3 for v in ld.values()), ld
    assert int(tr.model.engine.err) == 0 and int(tr.ema.model.engine.err) == 0
    c = tr.model._last_fused
    assert [tuple(p.shape[1:3]) for p in c.P] == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]          # 800 x 1344 padded input, strides 4 .. 64
    assert int(tr.ema.model._last_inference.pseudo["count"].sum()) > 0
    sd1 = tr.model.state_dict()
    assert set(sd1) == set(sd0)
    still = [k for k in sd1 if torch.equal(sd1[k], sd0[k])]
    assert not still, still[:8]                                          # nothing frozen: every tensor of the state moved
    assert all(torch.isfinite(v).all() for v in sd1.values())
    # layer-wise lr decay 0.7: with AdamW's normalised steps the update of a block scales with its lr multiplier 0.7 ** (12 - i)
    def moved(k):
        return float((sd1[k].float() - sd0[k].float()).abs().mean())
    m0, m11 = moved("backbone.net.blocks.0.mlp.fc1.weight"), moved("backbone.net.blocks.11.mlp.fc1.weight")
    assert m11 > 10 * m0 > 0, (m0, m11)                                  # (0.7 ** 11 = 0.02)
    # the teacher (EMA 0.9996) trails: it has moved, and far less than the student
    dt, ds = float((T.master - w0).abs().max()), float((W.master - w0).abs().max())
    assert 0 < dt < 0.05 * ds, (dt, ds)
