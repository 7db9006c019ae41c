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


def _run(graph, align, bf16=False):
    tr = _trainer(graph, align, bf16)
    out = []
    for it in range(ITERS):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        out.append({k: float(v) for k, v in tr._trainer.last_loss_dict.items()})
    torch.cuda.synchronize()
    fs = tr._trainer._fused_step
    c = tr.model._last_fused
    res = dict(losses=out, w=tr.model.weights.master.clone(), t=tr.ema.model.weights.master.clone(), stats=dict(fs.stats),
               rng=torch.get_rng_state(), py=random.random(), labels=c.rpn_labels.clone(), r_idx=c.r_idx[: c.R].clone(),
               err=int(tr.model.engine.err) | int(tr.ema.model.engine.err))
    return res


@pytest.mark.parametrize("align", [False, True])
def test_graph_replay_equals_eager_fp32(align):
    e = _run(False, align)
    g = _run(True, align)
    assert e["err"] == 0 and g["err"] == 0
    assert e["stats"]["captures"] == 0 and e["stats"]["eager"] == ITERS
    assert g["stats"]["captures"] == 2 and g["stats"]["replays_a"] == ITERS - 3 and g["stats"]["replays_b"] == ITERS - 3, g["stats"]
    for it, (a, b) in enumerate(zip(e["losses"], g["losses"])):
        assert list(a) == list(b)
        for k in a:
            # identical arithmetic; only the order of the fp32 atomics in the weight gradients differs from run to run
            assert abs(a[k] - b[k]) <= (1e-5 if it == 0 else 2e-4) * max(1.0, abs(a[k])), (it, k, a[k], b[k])
    assert torch.equal(e["rng"], g["rng"]) and e["py"] == g["py"]            # the host drew the same numbers
    assert torch.equal(e["labels"], g["labels"]) and torch.equal(e["r_idx"], g["r_idx"])
    n = (e["w"] - g["w"]).abs().max().item()
    assert n <= 2e-6 * max(1.0, e["w"].abs().max().item()), n
    assert (e["t"] - g["t"]).abs().max().item() <= 2e-6 * max(1.0, e["t"].abs().max().item())


def test_graph_replay_bf16_runs_the_benchmark_dtype():
    e = _run(False, False, bf16=True)
    g = _run(True, False, bf16=True)
    assert g["stats"]["captures"] == 2 and g["err"] == 0
    for a, b in zip(e["losses"], g["losses"]):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-2 * max(1.0, abs(a[k])), (k, a[k], b[k])
    assert torch.isfinite(g["w"]).all()


def test_env_switch_disables_graphs(monkeypatch):
    monkeypatch.setenv("ALDI_STEP_GRAPH", "0")
    tr = _trainer(True, False)
    for it in range(5):
        tr.iter = it
        tr.before_step(); tr.run_step(); tr.after_step()
    torch.cuda.synchronize()
    assert tr._trainer._fused_step.stats["captures"] == 0
