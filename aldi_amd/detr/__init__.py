"""Deformable-DETR on the HIP library (BASELINE configs[4]): the multi-scale deformable attention op
(ms_deform_attn.py), the detector after its backbone with its taped backward (model.py), the set criterion (criterion.py) and the
META_ARCHITECTURE "DeformableDETR" with its distill / align mixins (detector.py; reference aldi/detr/distill.py:6-7, aldi/detr/align.py:6-7)."""
from .ms_deform_attn import MSDeformAttnFunction, ms_deform_attn  # noqa: F401
from .detector import DeformableDETR  # noqa: F401
from ..align import ALIGN_MIXIN_REGISTRY
from ..distill import DISTILL_MIXIN_REGISTRY


@DISTILL_MIXIN_REGISTRY.register()
class DETRDistillMixin(DeformableDETR):
    pass


@ALIGN_MIXIN_REGISTRY.register()
class DETRAlignMixin(DeformableDETR):
    """(registered as the reference does; adversarial alignment itself is not implemented for this detector: forward(do_align=True) raises)"""
    img_align = ins_align = None
