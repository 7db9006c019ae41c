"""Deformable-DETR pieces of the ALDI path on the HIP library.  Only the multi-scale deformable
attention sampling op exists so far; the detector around it (reference `aldi/detr/`, an absent submodule) is not built."""
from .ms_deform_attn import MSDeformAttnFunction, ms_deform_attn  # noqa: F401
