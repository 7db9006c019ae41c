"""The generated gfx950 code of every kernel with hand-issued LDS reads (`asm volatile("ds_read_b128 ...")` + a later frag_wait) is checked for
instructions that touch such a register before the wait -- register copies the compiler may insert on control-flow edges because it takes an asm's
outputs for available at once (tools/isa_hazard_check.py; found in round 4 in the 128x64 tile's loop tail).  Cross-compiles with hipcc, no GPU."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src", ["igemm.hip", "bneck.hip", "wgrad.hip"])
def test_no_register_is_touched_before_its_lds_read_has_landed(src):
    import isa_hazard_check as H
    res = H.check_file(os.path.join(ROOT, "aldi_amd", "csrc", src))
    assert res, "no kernel with hand-issued LDS reads found: the parser lost track of the assembly format"
    bad = {k: v[:3] for k, v in res.items() if v}
    assert not bad, bad


def test_the_checker_sees_a_copy_between_read_and_wait():
    import isa_hazard_check as H
    asm = """
_Z6kernelv:
\tv_mov_b32_e32 v1, 0
\t;;#ASMSTART
\tds_read_b128 v[4:7], v1 offset:0
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
.LBB0_1:
\tv_mov_b32_e32 v9, v6
.LBB0_2:
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\t;;#ASMEND
\tv_mov_b32_e32 v10, v6
\ts_endpgm
"""
    (rep,) = H.parse_kernels(asm).values()
    found = H.check_kernel(rep)
    assert len(found) == 1 and found[0][1] == "v_mov_b32_e32 v9, v6"


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hot_kernels_do_not_spill():
    """the production instantiations of the kernels with counted `vmcnt` waits use no scratch: a spilled register is a vector-memory operation inside
    their loops that the counts do not include (and the 256 x 256 halo tile is within 30 registers of its budget: tools/isa_hazard_check.scratch_sizes)"""
    import isa_hazard_check as H
    sizes = H.scratch_sizes(os.path.join(ROOT, "aldi_amd", "csrc", "igemm.hip"))
    hot = {k: v for k, v in sizes.items() if ("igemm_halo64" in k and "Li256ELi256ELi4ELi2E" in k) or ("igemm_halo64" in k and "Li128ELi128E" in k)
           or "igemm_ws_kernel" in k}
    assert len(hot) >= 12, sorted(sizes)
    assert not {k: v for k, v in hot.items() if v}, {k: v for k, v in hot.items() if v}
    wg = H.scratch_sizes(os.path.join(ROOT, "aldi_amd", "csrc", "wgrad.hip"))
    hot = {k: v for k, v in wg.items() if "big64" in k or "lean64" in k}
    assert len(hot) >= 3 and not any(hot.values()), hot
