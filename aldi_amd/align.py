"""AlignMixin + discriminators: same registry, class names, constructor keys and loss names as
reference aldi/align.py:11-135.  The discriminator arithmetic (Conv/FC, gradient reversal, BCE
against the constant domain label, fwd+bwd) runs in the engine (aldi_amd.engine.RCNN.align_forward
/ backward); these classes are the plugin surface."""
from __future__ import annotations

from .model import GeneralizedRCNN
from .registry import Registry

ALIGN_MIXIN_REGISTRY = Registry("ALIGN_MIXIN")


class _DiscriminatorView:
    """What `model.img_align` / `model.ins_align` evaluate to (only tested for `is not None` by the
    step driver, aldi/trainer.py:48); exposes the D2 parameter names it owns."""
    def __init__(self, model, prefix, keys):
        self._model, self.prefix, self.keys = model, prefix, keys

    def state_dict(self):
        sd = self._model.state_dict()
        return {k[len(self.prefix) + 1:]: v for k, v in sd.items() if k.startswith(self.prefix + ".")}

    def parameters(self):
        return list(self.state_dict().values())


class ConvDiscriminator(_DiscriminatorView):
    """Conv2d(C, hidden, 3, padding=0) -> ReLU -> AdaptiveAvgPool2d(1) -> Flatten -> Linear(hidden, 1) (aldi/align.py:103-119)."""


class FCDiscriminator(_DiscriminatorView):
    """Flatten -> Linear(C, hidden) -> ReLU -> Linear(hidden, 1) (aldi/align.py:121-135)."""


@ALIGN_MIXIN_REGISTRY.register()
class AlignMixin(GeneralizedRCNN):
    def __init__(self, cfg):
        super().__init__(cfg)
        a = cfg.DOMAIN_ADAPT.ALIGN
        self.img_da_layer = a.IMG_DA_LAYER
        self.img_da_weight = a.IMG_DA_WEIGHT
        self.ins_da_weight = a.INS_DA_WEIGHT
        if a.IMG_DA_ENABLED:
            assert a.IMG_DA_LAYER == "p2" and list(a.IMG_DA_HIDDEN_DIMS) == [256] and a.IMG_DA_INPUT_DIM == 256, \
                "HIP path implements the reference default image discriminator (p2, 256 -> [256] -> 1)"
        if a.INS_DA_ENABLED:
            assert list(a.INS_DA_HIDDEN_DIMS) == [1024] and a.INS_DA_INPUT_DIM == 1024, \
                "HIP path implements the reference default instance discriminator (1024 -> [1024] -> 1)"
        self.img_align = ConvDiscriminator(self, "img_align", ("model.0", "model.4")) if a.IMG_DA_ENABLED else None
        self.ins_align = FCDiscriminator(self, "ins_align", ("model.1", "model.3")) if a.INS_DA_ENABLED else None

    def forward(self, *args, do_align=False, labeled=True, **kwargs):
        output = super().forward(*args, do_align=do_align, labeled=labeled, **kwargs)
        if self.training and not do_align and (self.img_align or self.ins_align):
            # reference aldi/align.py:91-100: a zero "_da" output so every parameter is "used"
            import torch
            output["_da"] = torch.zeros((), device=self.device)
        return output
