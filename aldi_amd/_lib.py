"""ctypes binding of libaldi_hip.so (the C ABI declared in include/aldi_hip.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing the
import of this module raises, and every wrapper raises ``AldiHipError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaldi_hip.so")


class AldiHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C aldi_amd/csrc`). There is no CPU fallback for the ALDI HIP path.")

lib = C.CDLL(LIB_PATH)
lib.aldi_last_error.restype = C.c_char_p
lib.aldi_version.restype = C.c_int

F32, BF16 = 0, 1
c_void_p, c_int, c_float, c_long = C.c_void_p, C.c_int, C.c_float, C.c_long


def check(status: int, what: str = ""):
    if status != 0:
        raise AldiHipError(f"{what}: status {status}: {lib.aldi_last_error().decode()}")


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("y_f32", c_void_p),
        ("scale", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("mask", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
        ("Cout", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int),
        ("Ho", c_int), ("Wo", c_int),
        ("relu", c_int), ("res_mode", c_int), ("out_scale", c_int), ("OH", c_int), ("OW", c_int),
        ("dtype", c_int),
    ]


def _sig(name, *argtypes):
    fn = getattr(lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = c_int
    return fn


conv_igemm = _sig("aldi_conv_igemm", C.POINTER(ConvArgs), c_void_p)


class WgradArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("g", c_void_p), ("dw", c_void_p), ("scale", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int), ("KH", c_int), ("KW", c_int),
        ("stride", c_int), ("pad", c_int), ("Ho", c_int), ("Wo", c_int), ("dtype", c_int),
    ]


conv_wgrad = _sig("aldi_conv_wgrad", C.POINTER(WgradArgs), c_void_p)
bias_grad = _sig("aldi_bias_grad", c_void_p, c_void_p, c_int, c_int, c_int, c_void_p)
dgrad_weights = _sig("aldi_dgrad_weights", c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p)
