"""Checkpoint I/O with the reference's semantics (aldi/checkpoint.py:8-32; detectron2 `DetectionCheckpointer` /
fvcore `Checkpointer` as reached from aldi/trainer.py:151-156).

File format = what detectron2 writes: `torch.save({"model": state_dict, <name>: obj.state_dict() for every checkpointable,
**extra}, "<save_dir>/<name>.pth")` plus a `last_checkpoint` text file naming the most recent one.  State-dict keys are
Detectron2's (`backbone.bottom_up.res2.0.conv1.weight`, ...): `ParamLayout.pack/unpack` translate to the engine's flat
state, so files written by the reference load here and vice versa.  The EMA checkpointable stores its keys with a
`model.` prefix exactly like the reference's `EMA` module.

`DetectionCheckpointerWithEMA.resume_or_load(path, resume=False)` additionally starts the model from the file's `ema`
entry (burn-in with EMA, `cfg.EMA.LOAD_FROM_EMA_ON_START`).

`.pkl` files (reference configs/Base-RCNN-FPN.yaml:3 `detectron2://ImageNetPretrained/MSRA/R-50.pkl`, docs/MODELS.md model
zoo) are read as detectron2's `DetectionCheckpointer._load_file` does: a pickle holding {"model": {name: ndarray}, "__author__",
["matching_heuristics"]}.  Files flagged `matching_heuristics` carry Caffe2 names (`res2_0_branch2a_w`, `res2_0_branch2a_bn_s`,
`conv1_w`, `res_conv1_bn_b`, `fc1000_w` ...): they go through the published renaming rules of detectron2's
`c2_model_loading.convert_basic_c2_names` and are then matched to the model's keys by longest dotted suffix
(`align_and_update_state_dicts`: `backbone.bottom_up.res2.0.conv1.weight` <- `res2.0.conv1.weight`).  FrozenBN statistics
absent from such a file (the MSRA weights have the affine only) load as mean 0 / var 1, detectron2's
`FrozenBatchNorm2d._load_from_state_dict` rule.  `detectron2://` URLs need a local copy (no network): pass the file path.
Parity of the renaming is unpinned (detectron2 absent); tests build a Caffe2-named file from known weights."""
from __future__ import annotations

import logging
import os
import pickle
import re
from typing import Any, Dict, List, Optional

import torch


class _IncompatibleKeys:
    def __init__(self, missing_keys, unexpected_keys, incorrect_shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing_keys, unexpected_keys, incorrect_shapes


def convert_c2_names(names: List[str]) -> List[str]:
    """Caffe2 / MSRA blob names -> detectron2 parameter-name suffixes (the basic rules: backbone + norm layers)."""
    out = []
    for k in names:
        k = k.replace("_", ".")
        k = re.sub(r"\.b$", ".bias", k)
        k = re.sub(r"\.w$", ".weight", k)
        k = re.sub(r"bn\.s$", "norm.weight", k)
        k = re.sub(r"bn\.bias$", "norm.bias", k)
        k = re.sub(r"bn\.rm$", "norm.running_mean", k)
        k = re.sub(r"bn\.running\.mean$", "norm.running_mean", k)
        k = re.sub(r"bn\.riv$", "norm.running_var", k)
        k = re.sub(r"bn\.running\.var$", "norm.running_var", k)
        k = re.sub(r"bn\.gamma$", "norm.weight", k)
        k = re.sub(r"bn\.beta$", "norm.bias", k)
        k = re.sub(r"gn\.s$", "norm.weight", k)
        k = re.sub(r"gn\.bias$", "norm.bias", k)
        k = re.sub(r"^res\.conv1\.norm\.", "conv1.norm.", k)          # the stem's norm is stored as res_conv1_bn_*
        k = re.sub(r"^conv1\.", "stem.conv1.", k)
        k = k.replace(".branch1.", ".shortcut.").replace(".branch2a.", ".conv1.").replace(".branch2b.", ".conv2.").replace(".branch2c.", ".conv3.")
        out.append(k)
    return out


def align_by_suffix(model_keys: List[str], ckpt_keys: List[str]) -> Dict[str, str]:
    """model key -> checkpoint key whose name is the LONGEST dotted suffix of it (detectron2 align_and_update_state_dicts)"""
    match: Dict[str, str] = {}
    for mk in model_keys:
        best = None
        for ck in ckpt_keys:
            if mk == ck or mk.endswith("." + ck):
                if best is None or len(ck) > len(best):
                    best = ck
        if best is not None:
            match[mk] = best
    return match


def load_pkl(path: str, model_keys: List[str]) -> Dict[str, Any]:
    """detectron2 model-zoo / Caffe2 `.pkl` -> {"model": {detectron2 key: tensor}, ...}"""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    if "model" in data and "__author__" in data:
        raw = data["model"]
    else:                                              # a bare Caffe2 blob dict
        raw = data["blobs"] if "blobs" in data else data
        data = {"model": raw, "__author__": "Caffe2", "matching_heuristics": True}
    raw = {k: torch.as_tensor(v) for k, v in raw.items() if not k.endswith("_momentum")}
    if data.get("matching_heuristics", False):
        names = list(raw.keys())
        conv = dict(zip(convert_c2_names(names), names))
        match = align_by_suffix(model_keys, list(conv.keys()))
        model = {mk: raw[conv[ck]] for mk, ck in match.items()}
        for mk in model_keys:                          # FrozenBN buffers absent from an affine-only file
            if mk not in model and mk.endswith(".norm.running_mean") and mk[:-len("running_mean")] + "weight" in model:
                model[mk] = torch.zeros_like(model[mk[:-len("running_mean")] + "weight"])
            if mk not in model and mk.endswith(".norm.running_var") and mk[:-len("running_var")] + "weight" in model:
                model[mk] = torch.ones_like(model[mk[:-len("running_var")] + "weight"])
        used = set(match.values())
        data["unmatched_checkpoint_keys"] = [conv[k] for k in conv if k not in used]
    else:
        model = raw
    out = {k: v for k, v in data.items() if k != "model"}
    out["model"] = model
    return out


class DetectionCheckpointer:
    def __init__(self, model, save_dir: str = "", *, save_to_disk: Optional[bool] = None, **checkpointables):
        self.model = model.module if hasattr(model, "module") else model
        self.save_dir = save_dir
        self.save_to_disk = True if save_to_disk is None else save_to_disk
        self.checkpointables: Dict[str, Any] = dict(checkpointables)
        self.logger = logging.getLogger(__name__)

    def add_checkpointable(self, key: str, checkpointable: Any) -> None:
        if key in self.checkpointables:
            raise KeyError(f"Key {key} already used in the Checkpointer")
        if not hasattr(checkpointable, "state_dict"):
            raise TypeError("add_checkpointable needs an object with 'state_dict()' method.")
        self.checkpointables[key] = checkpointable

    # ---- save ------------------------------------------------------------------------------------
    def save(self, name: str, **kwargs: Any) -> None:
        if not self.save_dir or not self.save_to_disk:
            return
        data: Dict[str, Any] = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        for key, obj in self.checkpointables.items():
            sd = obj.state_dict()
            data[key] = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()} if isinstance(sd, dict) else sd
        data.update(kwargs)
        basename = f"{name}.pth"
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, basename)
        self.logger.info("Saving checkpoint to %s", path)
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)

    # ---- load ------------------------------------------------------------------------------------
    def has_checkpoint(self) -> bool:
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self) -> str:
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ""

    def _load_model(self, state_dict: Dict[str, torch.Tensor]) -> _IncompatibleKeys:
        own = self.model.state_dict()
        shapes = [k for k, v in state_dict.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
        usable = {k: v for k, v in state_dict.items() if k in own and k not in shapes}
        missing = [k for k in own if k not in usable]
        unexpected = [k for k in state_dict if k not in own]
        merged = {k: (usable[k] if k in usable else own[k]) for k in own}
        self.model.load_state_dict(merged)
        return _IncompatibleKeys(missing, unexpected, shapes)

    def _log_incompatible_keys(self, inc: _IncompatibleKeys) -> None:
        for what, keys in (("missing", inc.missing_keys), ("unexpected", inc.unexpected_keys), ("shape-mismatched", inc.incorrect_shapes)):
            if keys:
                self.logger.warning("%s keys in checkpoint: %s", what, ", ".join(list(keys)[:8]) + (" ..." if len(keys) > 8 else ""))

    def load(self, path: str, checkpointables=None) -> Dict[str, Any]:
        if not path:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if path.startswith("detectron2://"):
            raise AssertionError(f"{path}: model-zoo URLs cannot be fetched here (no network); download the file and pass its path")
        if not os.path.isfile(path):
            raise AssertionError(f"Checkpoint {path} not found!")
        if path.endswith(".pkl"):
            ck = load_pkl(path, list(self.model.state_dict().keys()))
            if ck.get("unmatched_checkpoint_keys"):
                self.logger.warning("checkpoint keys not used by the model: %s", ", ".join(ck["unmatched_checkpoint_keys"][:8]))
        else:
            ck = torch.load(path, map_location="cpu", weights_only=False)
        if "model" not in ck:                        # a bare state_dict
            ck = {"model": ck}
        self._log_incompatible_keys(self._load_model(ck.pop("model")))
        for key in self.checkpointables if checkpointables is None else checkpointables:
            if key in ck:
                self.checkpointables[key].load_state_dict(ck[key])      # left IN the returned dict (the EMA start reads it)
        return ck

    def resume_or_load(self, path: str, *, resume: bool = True) -> Dict[str, Any]:
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])


class DetectionCheckpointerWithEMA(DetectionCheckpointer):
    """Start training from the EMA weights of a burned-in checkpoint (aldi/checkpoint.py:8-32)."""
    def resume_or_load(self, path: str, *, resume: bool = True) -> Dict[str, Any]:
        ret = super().resume_or_load(path, resume=resume)
        if (not resume) and path.endswith(".pth") and "ema" in ret.keys():
            self.logger.info("Loading EMA weights as model starting point.")
            ema_dict = {k.replace("model.", "", 1): v for k, v in ret["ema"].items()}
            self._log_incompatible_keys(self._load_model(ema_dict))
        return ret
