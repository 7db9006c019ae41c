"""aldi_torch_randperm_prefix reproduces torch.randperm(n)[:k] AND the generator state afterwards, bit for bit (CPU only)."""
import pytest
import torch


@pytest.mark.parametrize("seed", [0, 7, 2**31 + 5])
def test_randperm_prefix_is_torch_randperm(seed):
    from aldi_amd.engine import randperm_prefix
    import aldi_amd.engine as E
    cases = [(0, 4), (1, 1), (2, 2), (3, 1), (623, 9), (624, 624), (625, 3), (1000, 128), (159882, 256), (268569, 256), (268569, 0), (1132, 512)]
    for n, k in cases:
        torch.manual_seed(seed)
        torch.randperm(91)                      # shift the position inside the 624-word block
        a = torch.randperm(n)[:k]
        tail_a = (torch.randperm(33), torch.rand(3))
        torch.manual_seed(seed)
        torch.randperm(91)
        b = randperm_prefix(n, k)
        tail_b = (torch.randperm(33), torch.rand(3))
        assert E._FAST_RANDPERM is True
        assert torch.equal(a, b), (n, k)
        assert torch.equal(tail_a[0], tail_b[0]) and torch.equal(tail_a[1], tail_b[1]), (n, k)


def _script(ops_, threads):
    """run a draw / reseed script through aldi_torch_rng_script on the GLOBAL torch generator; returns the draws"""
    import ctypes as C
    from aldi_amd import _lib as L
    rows, off, outs = [], 0, []
    for op in ops_:
        if op[0] == "seed":
            rows += [1, op[1], 0, 0]
        else:
            _, n, k, keep = op
            rows += [0, n, k, off if keep else -1]
            if keep:
                outs.append((off, min(n, k)))
                off += min(n, k)
    script = (C.c_long * len(rows))(*rows)
    out = torch.zeros(max(off, 1), dtype=torch.int32)
    st = torch.get_rng_state()
    L.call("aldi_torch_rng_script", st.data_ptr(), script, len(ops_), out.data_ptr(), threads)
    torch.set_rng_state(st)
    return [out[o:o + k].clone() for o, k in outs]


@pytest.mark.parametrize("threads", [1, 8])
def test_rng_script_equals_torch_sequence(threads):
    """a step-shaped script (large negative lists, re-seeds in between, a discarded draw) == the same torch calls, including
    the generator state it leaves behind (normal-sample caches reset by manual_seed)"""
    ops_ = [("draw", 300, 128, True), ("draw", 268000, 128, True), ("draw", 17, 256, True), ("seed", 123456789), ("draw", 1100, 128, True),
            ("draw", 900, 384, True), ("seed", 4294967295), ("draw", 268569, 256, False), ("draw", 0, 10, True), ("draw", 1, 1, True),
            ("seed", 7), ("draw", 70001, 300, True)]
    torch.manual_seed(99)
    torch.randn(3)                                      # leaves a cached normal sample in the generator state
    start = torch.get_rng_state()
    ref = []
    for op in ops_:
        if op[0] == "seed":
            torch.manual_seed(op[1])
        else:
            p = torch.randperm(op[1])[: op[2]]
            if op[3]:
                ref.append(p.to(torch.int32))
    ref_state, ref_next = torch.get_rng_state(), torch.randperm(50)
    torch.set_rng_state(start)
    got = _script(ops_, threads)
    assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))
    assert torch.equal(torch.get_rng_state(), ref_state)
    assert torch.equal(torch.randperm(50), ref_next)
    # a script without any seed continues the incoming stream
    torch.set_rng_state(start)
    a = torch.randperm(5000)[:40].to(torch.int32)
    s1 = torch.get_rng_state()
    torch.set_rng_state(start)
    (b,) = _script([("draw", 5000, 40, True)], threads)
    assert torch.equal(a, b) and torch.equal(torch.get_rng_state(), s1)
