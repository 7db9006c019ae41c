"""Checkpoint semantics of the reference (aldi/checkpoint.py:8-32, fvcore Checkpointer file format) on plain objects."""
import os

import pytest
import torch

from aldi_amd.checkpoint import DetectionCheckpointer, DetectionCheckpointerWithEMA


class _Model:
    def __init__(self, v):
        self.sd = {"a.weight": torch.full((2, 3), float(v)), "b.bias": torch.full((4,), float(v) + 0.5)}

    def state_dict(self):
        return {k: v.clone() for k, v in self.sd.items()}

    def load_state_dict(self, sd):
        assert set(sd) == set(self.sd)
        self.sd = {k: v.clone() for k, v in sd.items()}


class _EMA:
    def __init__(self, v):
        self.model = _Model(v)

    def state_dict(self):
        return {"model." + k: v for k, v in self.model.state_dict().items()}

    def load_state_dict(self, sd):
        self.model.load_state_dict({k[len("model."):]: v for k, v in sd.items()})


def test_save_layout_and_resume(tmp_path):
    m, e = _Model(1), _EMA(2)
    ck = DetectionCheckpointer(m, str(tmp_path))
    ck.add_checkpointable("ema", e)
    with pytest.raises(KeyError):
        ck.add_checkpointable("ema", e)
    ck.save("model_0000009", iteration=9)
    assert open(tmp_path / "last_checkpoint").read() == "model_0000009.pth"
    raw = torch.load(tmp_path / "model_0000009.pth", weights_only=False)
    assert set(raw) == {"model", "ema", "iteration"} and set(raw["ema"]) == {"model.a.weight", "model.b.bias"}
    m2, e2 = _Model(7), _EMA(8)
    ck2 = DetectionCheckpointer(m2, str(tmp_path), ema=e2)
    ret = ck2.resume_or_load("ignored.pth", resume=True)                 # resume: last_checkpoint wins, checkpointables restored
    assert ret["iteration"] == 9
    assert torch.equal(m2.sd["a.weight"], m.sd["a.weight"]) and torch.equal(e2.model.sd["b.bias"], e.model.sd["b.bias"])


def test_ema_start_semantics(tmp_path):
    m, e = _Model(1), _EMA(2)
    ck = DetectionCheckpointer(m, str(tmp_path), ema=e)
    ck.save("burnin")
    path = os.path.join(str(tmp_path), "burnin.pth")
    # plain checkpointer, resume=False: the 'model' entry, checkpointables untouched
    m2, e2 = _Model(7), _EMA(8)
    DetectionCheckpointer(m2, "", ema=e2).resume_or_load(path, resume=False)
    assert float(m2.sd["a.weight"][0, 0]) == 1.0 and float(e2.model.sd["a.weight"][0, 0]) == 8.0
    # WithEMA, resume=False: the model starts from the file's EMA weights (prefix stripped)
    m3, e3 = _Model(7), _EMA(8)
    DetectionCheckpointerWithEMA(m3, "", ema=e3).resume_or_load(path, resume=False)
    assert float(m3.sd["a.weight"][0, 0]) == 2.0 and float(m3.sd["b.bias"][0]) == 2.5
    assert float(e3.model.sd["a.weight"][0, 0]) == 8.0
    # WithEMA, resume=True with no last_checkpoint in its save_dir: behaves like resume=False minus the EMA start
    m4 = _Model(7)
    DetectionCheckpointerWithEMA(m4, str(tmp_path / "empty")).resume_or_load(path, resume=True)
    assert float(m4.sd["a.weight"][0, 0]) == 1.0


def test_partial_and_mismatched_state_dicts(tmp_path):
    m = _Model(1)
    torch.save({"model": {"a.weight": torch.zeros(2, 3), "b.bias": torch.zeros(5), "c.extra": torch.zeros(1)}}, tmp_path / "x.pth")
    ck = DetectionCheckpointer(m, "")
    ck.load(str(tmp_path / "x.pth"))
    assert float(m.sd["a.weight"].sum()) == 0.0 and float(m.sd["b.bias"][0]) == 1.5      # wrong shape skipped, kept
    with pytest.raises(AssertionError):
        ck.load("R-50.pkl")                                  # a missing file, of either format
    with pytest.raises(AssertionError):
        ck.load("detectron2://ImageNetPretrained/MSRA/R-50.pkl")   # model-zoo URLs need a local copy (no network)
    assert ck.load("") == {}


# ---- trainer-level checkpointing (CHECKPOINT_PERIOD / model_final / optimizer + LR-schedule resume)
class _Weights:
    def __init__(self):
        self._mom = None
        self.first_step = True
        self.dev = torch.device("cpu")

    @property
    def mom(self):
        if self._mom is None:
            self._mom = torch.zeros(10)
        return self._mom

    def zero_grad(self):
        pass

    def sgd_step(self, lr, momentum, wd):
        self.mom.add_(lr)
        self.first_step = False


class _TModel(_Model):
    training = True

    def __init__(self, v):
        super().__init__(v)
        self.weights = _Weights()


def _mini_trainer(tmp_path, max_iter, period):
    """DefaultTrainer with its collaborators replaced by host-only stand-ins (no GPU): what is under test is the hook logic"""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd import trainer as T

    class Mini(T.DefaultTrainer):
        @classmethod
        def build_model(cls, cfg):
            return _TModel(1)

        @classmethod
        def build_train_loader(cls, cfg):
            return None

        def _create_trainer(self, cfg, model, data_loader, optimizer):
            class Step:
                def __init__(s):
                    s.model, s.optimizer = model, optimizer

                def run_step(s):
                    s.optimizer.step()
            return Step()
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_list(["OUTPUT_DIR", str(tmp_path), "SOLVER.MAX_ITER", max_iter, "SOLVER.CHECKPOINT_PERIOD", period, "SOLVER.BASE_LR", 0.1,
                         "SOLVER.WARMUP_ITERS", 4, "SOLVER.WARMUP_FACTOR", 0.1, "SOLVER.STEPS", (6,), "SOLVER.GAMMA", 0.5])
    return Mini(cfg)


def test_periodic_and_final_checkpoints_and_full_resume(tmp_path):
    tr = _mini_trainer(tmp_path, 10, 4)
    tr.resume_or_load(resume=False)
    lrs = []
    for tr.iter in range(0, 6):                                      # "crash" after iteration 5
        lrs.append(tr._trainer.optimizer.param_groups[0]["lr"])
        tr.before_step(); tr.run_step(); tr.after_step()
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth"))
    assert files == ["model_0000003.pth"] and open(tmp_path / "last_checkpoint").read() == "model_0000003.pth"
    raw = torch.load(tmp_path / "model_0000003.pth", weights_only=False)
    assert raw["iteration"] == 3 and raw["trainer"]["iteration"] == 3 and raw["trainer"]["hooks"]["LRScheduler"] == {"last_epoch": 4}
    assert raw["trainer"]["_trainer"]["optimizer"]["format"] == "aldi_amd.flat_sgd"
    # a fresh process resumes: iteration, LR schedule position and momentum continue
    tr2 = _mini_trainer(tmp_path, 10, 4)
    tr2.resume_or_load(resume=True)
    assert tr2.start_iter == 4 and tr2.scheduler.last_iter == 4
    assert tr2._trainer.optimizer.param_groups[0]["lr"] == lrs[4]   # not the warm-up start again
    assert torch.allclose(tr2.model.weights.mom, torch.full((10,), sum(lrs[:4])))
    assert tr2.model.weights.first_step is False
    tr2.train()
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth"))
    assert files == ["model_0000003.pth", "model_0000007.pth", "model_final.pth"]
    assert torch.load(tmp_path / "model_final.pth", weights_only=False)["iteration"] == 9
    # the schedule the resumed run followed == an uninterrupted run's
    ref = _mini_trainer(tmp_path / "ref", 10, 0)
    assert [ref.scheduler.lr_at(i) for i in range(10)][4] == lrs[4]
    assert tr2._trainer.optimizer.param_groups[0]["lr"] == ref.scheduler.lr_at(10)


def test_best_checkpoint_per_test_set(tmp_path):
    """two DATASETS.TEST entries: results are keyed by data set, one `<test_set>_model_best` each (reference aldi/trainer.py:186-194)"""
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd import trainer as T
    saved = []

    class Ck:
        def save(self, name, **kw):
            saved.append((name, kw.get("iteration")))

    class Sched:
        def step(self):
            pass
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_list(["OUTPUT_DIR", "", "TEST.EVAL_PERIOD", 2, "SOLVER.MAX_ITER", 100, "DATASETS.TEST", ("a_val", "b_val")])
    tr = T.ALDITrainer.__new__(T.ALDITrainer)
    tr.cfg, tr.checkpointer, tr.scheduler, tr.max_iter = cfg, Ck(), Sched(), 100
    tr.ema = None
    scores = iter([{"a_val": {"bbox": {"AP50": 10.0}}, "b_val": {"bbox": {"AP50": 30.0}}},
                   {"a_val": {"bbox": {"AP50": 12.0}}, "b_val": {"bbox": {"AP50": 29.0}}}])
    tr.test = lambda cfg_, model_: next(scores)
    type(tr).model = property(lambda self: None)
    try:
        for tr.iter in range(4):
            tr.after_step()
    finally:
        type(tr).model = T.DefaultTrainer.model
    assert saved == [("a_val_model_best", 1), ("b_val_model_best", 1), ("a_val_model_best", 3)]


# ---- model-zoo .pkl files (reference configs/Base-RCNN-FPN.yaml:3: detectron2://ImageNetPretrained/MSRA/R-50.pkl)
def test_c2_name_conversion_rules():
    from aldi_amd.checkpoint import convert_c2_names
    src = ["conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "res2_0_branch2a_w", "res2_0_branch2a_bn_s", "res2_0_branch2b_bn_b",
           "res2_0_branch1_w", "res3_1_branch2c_w", "fc1000_w", "fc1000_b", "res4_5_branch2b_bn_rm", "res4_5_branch2b_bn_riv"]
    assert convert_c2_names(src) == ["stem.conv1.weight", "stem.conv1.norm.weight", "stem.conv1.norm.bias", "res2.0.conv1.weight",
                                     "res2.0.conv1.norm.weight", "res2.0.conv2.norm.bias", "res2.0.shortcut.weight", "res3.1.conv3.weight",
                                     "fc1000.weight", "fc1000.bias", "res4.5.conv2.norm.running_mean", "res4.5.conv2.norm.running_var"]


def test_msra_style_pkl_loads_into_the_r50_layout(tmp_path):
    """a Caffe2-named, affine-only backbone file (the MSRA R-50 layout) built from known weights lands on the right detectron2
    keys; heads / FPN keep the model's own values; missing FrozenBN statistics become mean 0 / var 1"""
    import pickle
    import numpy as np
    from aldi_amd import synthetic as syn
    from aldi_amd.arch import ParamLayout
    lay = ParamLayout(8)
    sd = syn.init_state_dict(8, seed=3)

    class M:
        def __init__(self):
            self.sd = {k: torch.zeros_like(sd[k]) + 7.0 for k in lay.state_dict_keys()}

        def state_dict(self):
            return dict(self.sd)

        def load_state_dict(self, new):
            self.sd = dict(new)
    blobs = {}
    bu = "backbone.bottom_up."
    back = {"stem.conv1": "conv1", "shortcut": "branch1", "conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c"}
    for k, v in sd.items():
        if not k.startswith(bu) or "running_" in k:
            continue
        parts = k[len(bu):].split(".")
        if parts[0] == "stem":
            base = "conv1" if parts[2] == "weight" else "res_conv1_bn"
            name = base + {"weight": "_w" if base == "conv1" else "_s", "bias": "_b"}[parts[-1]]
        else:
            base = f"{parts[0]}_{parts[1]}_{back[parts[2]]}"
            name = base + ("_w" if parts[3] == "weight" else ("_bn_s" if parts[-1] == "weight" else "_bn_b"))
        blobs[name] = v.numpy()
    blobs["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    with open(tmp_path / "R-50.pkl", "wb") as f:
        pickle.dump({"model": blobs, "__author__": "Caffe2", "matching_heuristics": True}, f)
    m = M()
    ret = DetectionCheckpointer(m, "").load(str(tmp_path / "R-50.pkl"), checkpointables=[])
    assert ret["unmatched_checkpoint_keys"] == ["fc1000_w"]
    for k in lay.state_dict_keys():
        if k.startswith(bu) and k.endswith("running_mean"):
            assert float(m.sd[k].abs().max()) == 0.0
        elif k.startswith(bu) and k.endswith("running_var"):
            assert bool((m.sd[k] == 1).all())
        elif k.startswith(bu):
            assert torch.equal(m.sd[k], sd[k]), k
        else:
            assert bool((m.sd[k] == 7.0).all()), k                          # FPN / RPN / ROI heads untouched
