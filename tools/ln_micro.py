"""LayerNorm backward micro-benchmark: us per call and effective HBM rate for token-matrix shapes of the ViTDet / ConvNeXt trunks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import vit_ops as V  # noqa: E402


def timed(fn, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for rows, C in ((8400, 768), (16800, 768), (67200, 192), (134400, 256), (4200, 1536)):
    x = torch.randn(rows, C, device="cuda").bfloat16()
    g = torch.randn(rows, C, device="cuda").bfloat16()
    res = torch.randn(rows, C, device="cuda").bfloat16()
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y, mean, rstd = V.layernorm_forward(x, gamma, beta)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    t_f = timed(lambda: V.layernorm_forward(x, gamma, beta))
    t_b = timed(lambda: V.layernorm_backward(g, x, gamma, mean, rstd, dg, db, res=res))
    print("rows %6d C %4d: fwd %6.1f us (%.2f TB/s)  bwd %6.1f us (%.2f TB/s)" % (rows, C, t_f, rows * C * 4 / t_f / 1e6, t_b, rows * C * 8 / t_b / 1e6))
