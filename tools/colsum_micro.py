"""Bias-gradient (column sum) micro-benchmark: us per call and effective HBM read rate for the token-matrix shapes of the ViTDet /
ConvNeXt trunks.  Knobs: ALDI_COLSUM_BLOCKS, ALDI_COLSUM_MINROWS."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops  # noqa: E402


def timed(fn, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = 0.0
for rows, C in ((268800, 192), (268800, 768), (67200, 384), (67200, 1536), (16800, 768), (16800, 2304), (16800, 3072), (4200, 1536), (4200, 6144), (2000, 1024)):
    g = torch.randn(rows, C, device="cuda").bfloat16()
    db = torch.zeros(C, device="cuda")
    ops.bias_grad(g, db)
    err = (db - g.float().sum(0)).abs().max().item() / (rows ** 0.5)
    t = timed(lambda: ops.bias_grad(g, db))
    tot += t
    print("rows %6d C %4d: %6.1f us (%.2f TB/s)  err/sqrt(rows) %.1e" % (rows, C, t, rows * C * 2 / t / 1e6, err))
print("total %.1f us" % tot)
