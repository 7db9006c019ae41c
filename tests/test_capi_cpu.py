"""The C-ABI library loads without a GPU and exports every symbol include/aldi_hip.h declares."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_all_header_symbols():
    import aldi_amd._lib as L
    src = re.sub(r"/\*.*?\*/", "", open(L.HEADER_PATH).read(), flags=re.S)
    declared = set(re.findall(r"\b(aldi_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 30
    assert declared == set(L.PROTOS), declared ^ set(L.PROTOS)
    exported = set(re.findall(r" T (aldi_\w+)", subprocess.check_output(["nm", "-D", L.LIB_PATH]).decode()))
    assert declared <= exported, declared - exported
    assert L.lib.aldi_version() >= 1
    assert isinstance(L.lib.aldi_last_error(), bytes)


def test_struct_layouts_match_header_field_order():
    import aldi_amd._lib as L
    src = open(L.HEADER_PATH).read()

    def fields(struct_name):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + struct_name + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"\[[^\]]*\]", "", decl)
            for part in decl.split(","):
                names.append(part.strip().split()[-1].lstrip("*"))
        return names
    for cname, cls in (("aldi_conv_args", L.ConvArgs), ("aldi_wgrad_args", L.WgradArgs), ("aldi_stem_args", L.StemArgs),
                       ("aldi_rpn_geom", L.RpnGeom), ("aldi_roi_feats", L.RoiFeats)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_argument_errors_are_reported_not_raised_in_c():
    import aldi_amd._lib as L
    a = L.ConvArgs()          # all null
    rc = L.lib.aldi_conv_igemm(ctypes.byref(a), None)
    assert rc == -2 and b"null" in L.lib.aldi_last_error()
    try:
        L.call("aldi_conv_igemm", ctypes.byref(a), None)
        assert False
    except L.AldiHipError as e:
        assert "conv_igemm" in str(e)
