"""ctypes binding of libaldi_hip.so (the C ABI declared in include/aldi_hip.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing the
import of this module raises, and every wrapper raises ``AldiHipError`` on a non-zero status.

Function prototypes are parsed from the header itself, so the binding cannot drift from the ABI.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaldi_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "aldi_hip.h")


class AldiHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C aldi_amd/csrc`). There is no CPU fallback for the ALDI HIP path.")

lib = C.CDLL(LIB_PATH)

F32, BF16 = 0, 1
MAX_IMAGES, MAX_LEVELS = 16, 5
c_void_p, c_int, c_float, c_long = C.c_void_p, C.c_int, C.c_float, C.c_long


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl or decl.startswith("aldi_stream_t"):
        return C.c_void_p
    base = decl.rsplit(" ", 1)[0].replace("const", "").strip() if " " in decl else decl
    return {"int": C.c_int, "long": C.c_long, "float": C.c_float, "size_t": C.c_size_t, "unsigned": C.c_uint, "unsigned long long": C.c_ulonglong,
            "double": C.c_double}[base]


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long|size_t|const char\*)\s+(aldi_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        argtypes = [] if args in ("void", "") else [_ctype(a) for a in args.split(",")]
        restype = {"int": C.c_int, "long": C.c_long, "size_t": C.c_size_t, "const char*": C.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


PROTOS = parse_header()
for _name, (_res, _args) in PROTOS.items():
    _fn = getattr(lib, _name)        # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int, what: str = ""):
    if status != 0:
        raise AldiHipError(f"{what}: status {status}: {lib.aldi_last_error().decode()}")


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    check(getattr(lib, name)(*args), name)


def set_tuning(name: str, value: int):
    """aldi_set_tuning: select a dispatch arm / launch geometry at run time (include/aldi_hip.h lists the knobs)."""
    call("aldi_set_tuning", name.encode(), int(value))


def get_tuning(name: str) -> int:
    v = C.c_int(0)
    call("aldi_get_tuning", name.encode(), C.byref(v))
    return v.value


def reset_tuning():
    call("aldi_reset_tuning")


def last_dispatch() -> str:
    """kernel variant chosen by the most recent aldi_conv_igemm / aldi_conv_wgrad call of this thread"""
    return lib.aldi_last_dispatch().decode()


class BoxLossChunk(C.Structure):
    """aldi_box_loss_chunk (include/aldi_hip.h)"""
    _fields_ = [("r0", c_int), ("r1", c_int), ("grad_scale_cls", c_float), ("grad_scale_box", c_float), ("loss_box", c_void_p),
                ("teacher_pred", c_void_p), ("cls_temperature", c_float), ("kl", c_int), ("do_cls", c_int), ("do_reg", c_int),
                ("grad_scale_distill_cls", c_float), ("grad_scale_distill_reg", c_float), ("loss_distill", c_void_p)]


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("y_f32", c_void_p),
        ("scale", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("mask", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
        ("Cout", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int),
        ("Ho", c_int), ("Wo", c_int),
        ("relu", c_int), ("res_mode", c_int), ("out_scale", c_int), ("OH", c_int), ("OW", c_int),
        ("dtype", c_int), ("ws", c_void_p), ("ksplit", c_int), ("mask_bits", c_void_p), ("bits_out", c_void_p),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("g", c_void_p), ("dw", c_void_p), ("scale", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int), ("KH", c_int), ("KW", c_int),
        ("stride", c_int), ("pad", c_int), ("Ho", c_int), ("Wo", c_int), ("dtype", c_int), ("db", c_void_p),
        ("ws", c_void_p), ("ws_bytes", C.c_long),
    ]


class StemArgs(C.Structure):
    _fields_ = [
        ("img", c_void_p), ("w", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("y", c_void_p),
        ("N", c_int), ("Hs", c_int), ("Ws", c_int), ("Hc", c_int), ("Wc", c_int),
        ("h", c_int * MAX_IMAGES), ("w_img", c_int * MAX_IMAGES),
        ("mean", c_float * 3), ("std", c_float * 3), ("dtype", c_int),
    ]


class RpnGeom(C.Structure):
    _fields_ = [
        ("num_levels", c_int), ("A", c_int), ("C", c_int),
        ("H", c_int * MAX_LEVELS), ("W", c_int * MAX_LEVELS), ("off", c_int * (MAX_LEVELS + 1)),
    ]


class RoiFeats(C.Structure):
    _fields_ = [
        ("feat", c_void_p * 4), ("grad", c_void_p * 4), ("H", c_int * 4), ("W", c_int * 4),
        ("scale", c_float * 4), ("C", c_int),
    ]


class DgwItem(C.Structure):
    _fields_ = [("w_master", c_void_p), ("scale", c_void_p), ("wt", c_void_p),
                ("Cout", c_int), ("KH", c_int), ("KW", c_int), ("Cin", c_int), ("tile_begin", c_int), ("reserved", c_int)]


class FoldItem(C.Structure):
    _fields_ = [("w", c_void_p), ("scale", c_void_p), ("out", c_void_p), ("rows", c_int), ("cols", c_int), ("chunk_begin", c_int), ("reserved", c_int)]


class BottleneckArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("res", c_void_p), ("y", c_void_p), ("w1", c_void_p), ("w2", c_void_p), ("w3", c_void_p),
                ("b1", c_void_p), ("b2", c_void_p), ("b3", c_void_p),
                ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("mid", c_int), ("Cout", c_int)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("qkv", c_void_p), ("rel_h", c_void_p), ("rel_w", c_void_p),
        ("Qp", c_void_p), ("Kp", c_void_p), ("KpT", c_void_p), ("VT", c_void_p), ("QsT", c_void_p),
        ("O", c_void_p), ("lse", c_void_p), ("dO", c_void_p), ("dOT", c_void_p), ("dQp", c_void_p), ("delta", c_void_p),
        ("dqkv", c_void_p), ("drel_h", c_void_p), ("drel_w", c_void_p),
        ("nB", c_int), ("gh", c_int), ("gw", c_int), ("heads", c_int), ("Dq", c_int), ("scale", c_float),
    ]


PtrArray5 = c_void_p * MAX_LEVELS
