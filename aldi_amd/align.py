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
    """[Conv2d(., hidden_i, 3, padding=0) -> ReLU]* -> AdaptiveAvgPool2d(1) -> Flatten -> Linear(., 1) (aldi/align.py:103-119)."""


class FCDiscriminator(_DiscriminatorView):
    """Flatten -> [Linear(., hidden_i) -> ReLU]* -> Linear(., 1) (aldi/align.py:121-135)."""


@ALIGN_MIXIN_REGISTRY.register()
class AlignMixin(GeneralizedRCNN):
    def __init__(self, cfg):
        super().__init__(cfg)
        a = cfg.DOMAIN_ADAPT.ALIGN
        self.img_da_layer = a.IMG_DA_LAYER
        self.img_da_weight = a.IMG_DA_WEIGHT
        self.ins_da_weight = a.INS_DA_WEIGHT
        # any FPN level / hidden_dims list (aldi/align.py:22-52): the engine builds the layer chain from the layout's names
        self.img_align = ConvDiscriminator(self, "img_align", tuple(n[len("img_align."):] for n in self.engine.img_da_layers)) if a.IMG_DA_ENABLED else None
        self.ins_align = FCDiscriminator(self, "ins_align", tuple(n[len("ins_align."):] for n in self.engine.ins_da_layers)) if a.INS_DA_ENABLED else None

    def forward(self, *args, do_align=False, labeled=True, **kwargs):
        output = super().forward(*args, do_align=do_align, labeled=labeled, **kwargs)
        if self.training and not do_align and (self.img_align or self.ins_align):
            # reference aldi/align.py:91-100: a zero "_da" output so every parameter is "used"
            import torch
            output["_da"] = torch.zeros((), device=self.device)
        return output
