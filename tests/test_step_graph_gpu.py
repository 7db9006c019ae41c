"""SOLVER.STEP_GRAPH: the fused iteration replayed as two hipGraphs (aldi_amd/fused_step.py) computes what the same
iteration computes when every launch is issued from Python: same loss dicts step by step (fresh images / ground truth
every step, so the fixed input buffers are really rewritten), same weights and EMA teacher afterwards, same host RNG stream."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 192, 256
ITERS = 7


def _trainer(graph, align, bf16=False):
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SOLVER.AMP.ENABLED", bf16, "SOLVER.BASE_LR", 0.002, "SOLVER.WARMUP_ITERS", 0, "SEED", 1,
                         "EMA.ALPHA", 0.9, "SYNTHETIC.HEIGHT", H, "SYNTHETIC.WIDTH", W,
                         "DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED", align, "DOMAIN_ADAPT.ALIGN.INS_DA_ENABLED", align])
    cfg.SOLVER.FUSED_STEP = True
    cfg.SOLVER.STEP_GRAPH = graph
    random.seed(4)
    torch.manual_seed(17)
    return ALDITrainer(cfg)


def _step(tr, it):
    tr.iter = it
    tr.before_step()
    tr.run_step()
    tr.after_step()
    torch.cuda.synchronize()
    return {k: float(v) for k, v in tr._trainer.last_loss_dict.items()}


def _copy_state(src, dst):
    """dst starts its next iteration from exactly src's state: weights, momentum, teacher, host RNG streams"""
    for a, b in ((src.model, dst.model), (src.ema.model, dst.ema.model)):
        b.weights.master.copy_(a.weights.master)
        b.weights.refresh()
    dst.model.weights.mom.copy_(src.model.weights.mom)
    dst.model.weights.first_step = src.model.weights.first_step
    dst._trainer.distiller.seeder.seed = src._trainer.distiller.seeder.seed
    dst.scheduler.last_iter = src.scheduler.last_iter
    dst.scheduler._apply()


def _compare(align, bf16, loss_tol, grad_tol):
    """Two trainers on the same data stream, one issuing every launch from Python, one replaying the captured graphs.  Every
    iteration starts from IDENTICAL state (copied over, host RNG included), so the comparison is not blurred by the
    run-to-run noise of the fp32 atomics accumulating over iterations (which can flip a near-tied proposal or sample)."""
    e, g = _trainer(False, align, bf16), _trainer(True, align, bf16)
    for it in range(ITERS):
        _copy_state(e, g)
        rng, py = torch.get_rng_state(), random.getstate()
        le = _step(e, it)
        ge = e.model.weights.grad.clone()
        rng_e, py_e = torch.get_rng_state(), random.getstate()
        torch.set_rng_state(rng)
        random.setstate(py)
        lg = _step(g, it)
        assert torch.equal(torch.get_rng_state(), rng_e) and random.getstate() == py_e      # the host drew the same numbers
        assert list(le) == list(lg)
        for k in le:
            assert abs(le[k] - lg[k]) <= loss_tol * max(1.0, abs(le[k])), (it, k, le[k], lg[k])
        ce, cg = e.model._last_fused, g.model._last_fused
        assert torch.equal(ce.rpn_labels, cg.rpn_labels) and torch.equal(ce.r_idx[: ce.R], cg.r_idx[: cg.R]) and ce.rows == cg.rows
        gg = g.model.weights.grad
        assert (ge - gg).abs().max().item() <= grad_tol * max(1e-6, ge.abs().max().item()), (it, (ge - gg).abs().max().item(), ge.abs().max().item())
    fs = g._trainer._fused_step
    assert e._trainer._fused_step.stats["captures"] == 0 and e._trainer._fused_step.stats["eager"] == ITERS
    ncap = 4 if (fs.pipeline and g.model.engine.prefix_pipelinable()) else 2     # (pipelined steps alternate between two sets of graphs)
    assert fs.stats["captures"] == ncap and fs.stats["replays_a"] == ITERS - 3 and fs.stats["replays_b"] == ITERS - 3, fs.stats
    assert int(g.model.engine.err) == 0 and int(g.ema.model.engine.err) == 0
    w = g.model.weights.master
    assert torch.isfinite(w).all()
    d = (w - e.model.weights.master).abs().max().item()
    assert d <= 1e-5 * max(1.0, w.abs().max().item()), d


@pytest.mark.parametrize("align", [False, True])
def test_graph_replay_equals_eager_fp32(align):
    _compare(align, False, 2e-6, 2e-5)


def test_graph_replay_bf16_runs_the_benchmark_dtype():
    _compare(False, True, 2e-6, 2e-3)


def test_env_switch_disables_graphs(monkeypatch):
    monkeypatch.setenv("ALDI_STEP_GRAPH", "0")
    tr = _trainer(True, False)
    for it in range(5):
        tr.iter = it
        tr.before_step(); tr.run_step(); tr.after_step()
    torch.cuda.synchronize()
    assert tr._trainer._fused_step.stats["captures"] == 0


def test_paired_student_teacher_forward_equals_separate_passes():
    """phase A with the student's and the teacher's trunk / RPN head sharing one launch per layer (aldi_conv_igemm_group) == the
    two separate passes: pseudo-labels, sampled indices and losses of the same step from the same state"""
    a, b = _trainer(False, False), _trainer(False, False)
    for it in range(3):
        _copy_state(a, b)
        rng, py = torch.get_rng_state(), random.getstate()
        os.environ["ALDI_PAIR_FORWARD"] = "1"                 # read when the trainer creates its FusedStep (first iteration)
        try:
            la = _step(a, it)
        finally:
            os.environ.pop("ALDI_PAIR_FORWARD", None)
        assert a._trainer._fused_step.pair_forward
        torch.set_rng_state(rng)
        random.setstate(py)
        lb = _step(b, it)
        assert not b._trainer._fused_step.pair_forward
        ta, tb = a.ema.model._last_inference, b.ema.model._last_inference
        assert torch.equal(ta.pseudo["count"], tb.pseudo["count"]) and torch.equal(ta.pseudo["classes"], tb.pseudo["classes"])
        assert (ta.pseudo["boxes"] - tb.pseudo["boxes"]).abs().max().item() <= 1e-3
        ca, cb = a.model._last_fused, b.model._last_fused
        assert torch.equal(ca.rpn_labels, cb.rpn_labels) and torch.equal(ca.r_idx[: ca.R], cb.r_idx[: cb.R])
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-5 * max(1.0, abs(la[k])), (it, k, la[k], lb[k])


def test_reference_yaml_runs_the_fused_graph_step_by_default(monkeypatch):
    """`ALDITrainer(cfg)` on the reference's own YAML (configs/cityscapes/ALDI-Best-Cityscapes.yaml = the reference's file of that name
    without the keys that name files / datasets absent here -- MODEL.WEIGHTS, DATASETS.UNLABELED, AUG.*, OUTPUT_DIR;
    tests/test_host_logic_cpu.py holds the two files to the same config otherwise; reference tools/train_net.py:83-85) plus only the batch
    size / synthetic image size: NO aldi_amd extension key is set, and the iteration it runs is the fused step replayed from its two
    hipGraphs -- the step bench.py measures."""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    for k in ("ALDI_FUSED_STEP", "ALDI_STEP_GRAPH", "ALDI_FUSED_LEGACY"):
        monkeypatch.delenv(k, raising=False)
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SYNTHETIC.HEIGHT", H, "SYNTHETIC.WIDTH", W])
    random.seed(4)
    torch.manual_seed(17)
    tr = ALDITrainer(cfg)
    assert tr._trainer.fused
    for it in range(6):
        ld = _step(tr, it)
    fs = tr._trainer._fused_step
    assert fs is not None and fs.graph_enabled
    assert fs.stats["captures"] in (2, 4) and fs.stats["replays_a"] >= 2 and fs.stats["replays_b"] >= 2, fs.stats
    assert all(v == v for v in ld.values()) and "loss_cls_source_strong" in ld and "loss_roih_l1_distill" in ld
    # ... and the keys / the environment turn it off again
    monkeypatch.setenv("ALDI_FUSED_STEP", "0")
    tr2 = ALDITrainer(cfg)
    assert not tr2._trainer.fused
    _step(tr2, 0)
    assert getattr(tr2._trainer, "_fused_step", None) is None


@pytest.mark.gpu
def test_fused_step_steps_aside_for_foreign_distillers_and_hooks(monkeypatch):
    """the fused driver replaces ALDIDistiller's losses and does not fire the hook points: a subclass that overrides a loss method, or a
    hook somebody else registered on the student (a SaveIO tap, reference aldi/helpers.py:7-20), must get the reference's sequential schedule"""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.distill import ALDIDistiller
    from aldi_amd.helpers import SaveIO
    from aldi_amd.trainer import ALDITrainer
    for k in ("ALDI_FUSED_STEP", "ALDI_STEP_GRAPH", "ALDI_FUSED_LEGACY"):
        monkeypatch.delenv(k, raising=False)
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SYNTHETIC.HEIGHT", H, "SYNTHETIC.WIDTH", W])
    random.seed(4)
    torch.manual_seed(17)
    tr = ALDITrainer(cfg)
    assert tr._trainer._fusable_distiller()
    # (a) a tap of the user's own on the student's box predictor
    tap = SaveIO()
    tr.model.roi_heads.box_predictor.register_forward_hook(tap)
    assert not tr._trainer._fusable_distiller()
    ld = _step(tr, 0)
    assert getattr(tr._trainer, "_fused_step", None) is None and tap.output is not None      # sequential schedule: the hook fired
    assert "loss_roih_l1_distill" in ld
    tr.model.roi_heads.box_predictor.hooks.remove(tap)
    assert tr._trainer._fusable_distiller()

    # (b) a subclass with its own RoI-head loss
    class Mine(ALDIDistiller):
        calls = 0

        def get_roih_losses(self, *a, **kw):
            Mine.calls += 1
            return super().get_roih_losses(*a, **kw)

    tr._trainer.distiller.__class__ = Mine
    assert not tr._trainer._fusable_distiller()
    _step(tr, 1)
    assert Mine.calls == 1 and getattr(tr._trainer, "_fused_step", None) is None
    # a subclass that only adds attributes keeps the fast path
    class Plain(ALDIDistiller):
        note = "same methods"

    tr._trainer.distiller.__class__ = Plain
    assert tr._trainer._fusable_distiller()


def _trainer_pipe(pipe):
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SOLVER.AMP.ENABLED", True, "SOLVER.BASE_LR", 0.002, "SOLVER.WARMUP_ITERS", 0, "SEED", 1,
                         "EMA.ALPHA", 0.9, "SYNTHETIC.HEIGHT", H, "SYNTHETIC.WIDTH", W, "SOLVER.PIPELINE_PREFIX", pipe])
    random.seed(4)
    torch.manual_seed(17)
    return ALDITrainer(cfg)


def test_pipelined_frozen_prefix_equals_the_in_order_schedule():
    """SOLVER.PIPELINE_PREFIX (built and measured, default OFF: DESIGN.md section 17): stem + res2 of batch k + 1 run inside step k's phase A, under its proposal chain, and step k + 1
    starts its student pass at res3 -- against the same trainer with the prefix computed in order, over steps with a CHANGING batch (the
    synthetic loader draws fresh images every iteration), eager steps and replayed ones: (1) the res2 output a pipelined step reads is, bit for
    bit, what stem + res2 give on THAT step's images (recomputed here), (2) the steps' indices are identical and their losses / gradients agree
    within the run-to-run noise of the float atomics (the bound of the graph-vs-eager test), (3) only the first step computed its prefix in
    order; nothing is carried over from an older batch."""
    a, b = _trainer_pipe(True), _trainer_pipe(False)
    for it in range(ITERS):
        _copy_state(a, b)
        rng, py = torch.get_rng_state(), random.getstate()
        la = _step(a, it)
        ga = a.model.weights.grad.clone()
        fs = a._trainer._fused_step
        S = next(reversed(fs.static.values()))
        eng = a.model.engine
        used = S.stu.out.clone()                                              # what this step's student pass started from
        ref = torch.empty_like(used)
        with torch.no_grad():
            eng._drive(eng.trunk_steps(S.stu.img, S.stu.sizes, False, prefix_out=ref))
        torch.cuda.synchronize()
        assert torch.equal(used, ref), it
        assert float(used.float().abs().max()) > 0
        torch.set_rng_state(rng)
        random.setstate(py)
        lb = _step(b, it)
        assert list(la) == list(lb)
        ca, cb = a.model._last_fused, b.model._last_fused
        assert torch.equal(ca.rpn_labels, cb.rpn_labels) and torch.equal(ca.r_idx[: ca.R], cb.r_idx[: cb.R]) and ca.rows == cb.rows
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(la[k])), (it, k, la[k], lb[k])
        gb = b.model.weights.grad
        assert (ga - gb).abs().max().item() <= 2e-3 * max(1e-6, ga.abs().max().item())
    fa, fb = a._trainer._fused_step, b._trainer._fused_step
    assert fa.pipeline and not fb.pipeline
    assert fa.stats.get("prefix_inline", 0) == 1 and fa.stats.get("prefix_ahead", 0) == ITERS - 1, fa.stats
    assert fa.stats["captures"] == 4 and fb.stats["captures"] == 2
    assert "prefix_ahead" not in fb.stats
