"""Minimal Boxes / Instances (the subset of detectron2.structures the reference touches:
aldi/pseudolabeler.py:51-73, aldi/dataloader.py:28-29, aldi/distill.py:202)."""
from __future__ import annotations

from typing import Any, Dict, Tuple

import torch


class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        b = self.tensor[item]
        return Boxes(b.view(1, -1) if b.dim() == 1 else b)

    @property
    def device(self):
        return self.tensor.device


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, device):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(device) if hasattr(v, "to") else v)
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    # dict-style access used by the engine
    def __getitem__(self, key):
        if key == "image_size":
            return self._image_size
        return self._fields[key]


def as_record(inst) -> dict:
    """Instances | dict -> {'image_size', 'gt_boxes' (Tensor), 'gt_classes'}"""
    if isinstance(inst, dict):
        b = inst["gt_boxes"]
        return {"image_size": inst.get("image_size"), "gt_boxes": b.tensor if hasattr(b, "tensor") else b, "gt_classes": inst["gt_classes"]}
    b = inst.gt_boxes
    return {"image_size": inst.image_size, "gt_boxes": b.tensor if hasattr(b, "tensor") else b, "gt_classes": inst.gt_classes}
