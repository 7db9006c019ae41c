"""aldi_amd: the ALDI (justinkay/aldi) student+teacher training step, MI355X-native.

Hand-written gfx950 HIP kernels behind a C ABI (include/aldi_hip.h, aldi_amd/csrc) with a
thin Python host layer that mirrors the reference's plugin API (aldi.trainer / aldi.distill /
aldi.align / aldi.ema / aldi.model).  See DESIGN.md.
"""
__version__ = "0.1.0"
