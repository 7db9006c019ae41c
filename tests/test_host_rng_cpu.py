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


def _prefetch(seeds, max_draws, with_state=True):
    import ctypes as C
    from aldi_amd import _lib as L
    st = torch.get_rng_state()
    arr = (C.c_long * max(len(seeds), 1))(*seeds)
    L.call("aldi_torch_rng_prefetch", st.data_ptr() if with_state else None, arr, len(seeds), max_draws)


def _hits():
    from aldi_amd import _lib as L
    return L.lib.aldi_torch_rng_prefetch_hits()


@pytest.mark.parametrize("case", ["all", "short", "wrong_seed", "stale_state", "mid_block", "no_draws"])
def test_rng_script_from_prefetched_streams(case):
    """the same script served from pre-generated Mersenne streams (aldi_torch_rng_prefetch): identical draws and final
    generator state; streams that are too short / belong to another seed / another engine state are not used"""
    ops_ = [("draw", 300, 128, True), ("draw", 268000, 128, True), ("draw", 17, 256, True), ("seed", 123456789), ("draw", 1100, 128, True),
            ("draw", 900, 384, True), ("seed", 77), ("draw", 268569, 256, False), ("draw", 0, 10, True), ("draw", 1, 1, True),
            ("seed", 77), ("draw", 70001, 300, True), ("draw", 5000, 300, True)]
    if case == "no_draws":
        ops_ = [("draw", 1, 5, True), ("seed", 123456789), ("draw", 0, 1, True), ("seed", 77)]
    torch.manual_seed(4242)
    if case == "mid_block":
        torch.randperm(1000)                            # the engine sits in the middle of a state block
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
    seeds, depth = [123456789, 77], 300000
    if case == "short":
        depth = 1000                                    # only the 1100 + 900 segment fits... not even that: (1099 + 899) > 1000
    if case == "wrong_seed":
        seeds = [5, 6]
    _prefetch(seeds, depth)
    if case == "stale_state":
        torch.randperm(3)                               # the engine moved after the prefetch: its stream must be ignored
        torch.set_rng_state(start)
        torch.randperm(3)
        ref = None
    h0 = _hits()
    got = _script(ops_, 2)
    served = _hits() - h0
    if case == "stale_state":
        assert served == 3                              # the seeded segments only
        return
    assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))
    assert torch.equal(torch.get_rng_state(), ref_state)
    assert torch.equal(torch.randperm(50), ref_next)
    assert served == {"all": 4, "mid_block": 4, "short": 0, "wrong_seed": 1, "no_draws": 3}[case]


@pytest.mark.parametrize("layout", ["source_only", "one_each", "three_each"])
def test_step_draws_equals_the_reference_sequence(layout):
    """aldi_step_draws (the fused step's host phase as one native call) == the reference's sequence of torch.manual_seed /
    torch.randperm calls, micro-step by micro-step (aldi/trainer.py:51-52,86-89 with IMS_PER_GPU-sized chunks):
    sampled positions, counts, ROI row offsets, normalisers, generator state.  "three_each" = the reference's shipped shape on
    8 GPUs (IMS_PER_BATCH 48, IMS_PER_GPU 2: three source and three distillation micro-steps per iteration)."""
    import ctypes as C
    from aldi_amd import _lib as L
    chunks = {"source_only": [(0, 0, 2)], "one_each": [(0, 0, 2), (1, 2, 4)],
              "three_each": [(0, 0, 2), (0, 2, 4), (0, 4, 6), (1, 6, 8), (1, 8, 10), (1, 10, 12)]}[layout]
    N = chunks[-1][2]
    RB, RP, OB, OP = 256, 128, 512, 128
    gen = torch.Generator().manual_seed(11)
    rpn = [[int(torch.randint(0, 300, (1,), generator=gen)), int(torch.randint(100, 250000, (1,), generator=gen))] for _ in range(N)]
    roi = [[int(torch.randint(0, 200, (1,), generator=gen)), int(torch.randint(0, 2100, (1,), generator=gen))] for _ in range(N)]
    rpn[0][0] = 0                                                           # an image without positives
    nk = sum(1 for c in chunks if c[0] == 1)
    seeds = [1234567, 4000000001, 17, 3999999999][: nk + 1]
    nd = max(sum(c[2] - c[1] for c in chunks if c[0] == 1), 1)
    # word layout of the upload buffer
    names = [("rsel", N * 2 * RB), ("rnsel", N * 2), ("osel", N * 2 * OB), ("onsel", N * 2), ("row_off", N), ("dsel", nd * 2 * RB), ("dnsel", nd * 2),
             ("nvf", 2 * max(nk, 1))]
    w0, off = {}, 0
    for k, n in names:
        w0[k] = off
        off += (n + 3) // 4 * 4
    # ---- reference: torch calls in the reference's order
    torch.manual_seed(77)
    torch.randperm(500)
    start = torch.get_rng_state()
    ref = torch.full((off,), -7, dtype=torch.int32)

    def sample(dst, ndst, row0, cnts, batch, cap):
        sums = []
        for i, (npos, nneg) in enumerate(cnts):
            num_pos = min(npos, cap)
            num_neg = min(nneg, batch - num_pos)
            p1, p2 = torch.randperm(npos)[:num_pos], torch.randperm(nneg)[:num_neg]
            if dst is not None:
                b = w0[dst] + (row0 + i) * 2 * batch
                ref[b: b + num_pos] = p1.to(torch.int32)
                ref[b + batch: b + batch + num_neg] = p2.to(torch.int32)
                ref[w0[ndst] + 2 * (row0 + i)], ref[w0[ndst] + 2 * (row0 + i) + 1] = num_pos, num_neg
            sums.append((num_pos, num_neg))
        return sums
    rows, k, d0 = [], 0, 0
    for kind, n0, n1 in chunks:
        if kind == 1:
            torch.manual_seed(seeds[k])            # the teacher's eval inference fires ManualSeed with the current seed ...
            k += 1                                 # ... then the distiller resets the seeder (aldi/distill.py:148-150)
        sample("rsel", "rnsel", n0, rpn[n0:n1], RB, RP)          # student RPN sampling
        torch.manual_seed(seeds[k])                # ManualSeed on the student's roi_heads
        rows += [a + b for a, b in sample("osel", "onsel", n0, roi[n0:n1], OB, OP)]
        if kind == 1:
            torch.manual_seed(seeds[k])            # the teacher's train-mode roi_heads: identical ROI draws (aldi/distill.py:160-162)
            sample(None, None, 0, roi[n0:n1], OB, OP)
            dh = sample("dsel", "dnsel", d0, rpn[n0:n1], RB, RP)  # get_rpn_losses' fresh sample (aldi/distill.py:200-202)
            ref[w0["nvf"] + 2 * (k - 1)], ref[w0["nvf"] + 2 * (k - 1) + 1] = sum(a + b for a, b in dh), sum(a for a, _ in dh)
            d0 += n1 - n0
    o = 0
    for i, r in enumerate(rows):
        ref[w0["row_off"] + i] = o
        o += r
    ref_state = torch.get_rng_state()
    # ---- the native call
    torch.set_rng_state(start)
    got = torch.full((off,), -7, dtype=torch.int32)
    counts = torch.tensor([v for p in rpn for v in p] + [v for p in roi for v in p], dtype=torch.int32)
    carr = (C.c_int * (3 * len(chunks)))(*[v for c in chunks for v in c])
    warr = (C.c_int * 8)(*[w0[k_] for k_, _ in names])
    rarr = (C.c_int * N)()
    sarr = (C.c_long * len(seeds))(*seeds)
    st = torch.get_rng_state()
    L.call("aldi_step_draws", st.data_ptr(), counts.data_ptr(), N, carr, len(chunks), sarr, len(seeds), RB, RP, OB, OP, got.data_ptr(), warr, rarr, 4)
    torch.set_rng_state(st)
    assert list(rarr) == rows
    assert torch.equal(got, ref)
    assert torch.equal(torch.get_rng_state(), ref_state)
