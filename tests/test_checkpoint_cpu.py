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
    with pytest.raises(NotImplementedError):
        ck.load("R-50.pkl")
    assert ck.load("") == {}
