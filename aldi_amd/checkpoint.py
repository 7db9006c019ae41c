"""Checkpoint I/O with the reference's semantics (aldi/checkpoint.py:8-32; detectron2 `DetectionCheckpointer` /
fvcore `Checkpointer` as reached from aldi/trainer.py:151-156).

File format = what detectron2 writes: `torch.save({"model": state_dict, <name>: obj.state_dict() for every checkpointable,
**extra}, "<save_dir>/<name>.pth")` plus a `last_checkpoint` text file naming the most recent one.  State-dict keys are
Detectron2's (`backbone.bottom_up.res2.0.conv1.weight`, ...): `ParamLayout.pack/unpack` translate to the engine's flat
state, so files written by the reference load here and vice versa.  The EMA checkpointable stores its keys with a
`model.` prefix exactly like the reference's `EMA` module.

`DetectionCheckpointerWithEMA.resume_or_load(path, resume=False)` additionally starts the model from the file's `ema`
entry (burn-in with EMA, `cfg.EMA.LOAD_FROM_EMA_ON_START`).  Not implemented: the `.pkl` model-zoo / Caffe2 name
heuristics (`align_and_update_state_dicts`) -- a `.pkl` path raises."""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional

import torch


class _IncompatibleKeys:
    def __init__(self, missing_keys, unexpected_keys, incorrect_shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing_keys, unexpected_keys, incorrect_shapes


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
        if path.endswith(".pkl"):
            raise NotImplementedError("model-zoo .pkl checkpoints (Caffe2 / detectron2 name heuristics) are not supported; convert to .pth")
        if not os.path.isfile(path):
            raise AssertionError(f"Checkpoint {path} not found!")
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
