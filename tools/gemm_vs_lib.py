"""Yardstick for the token-matrix linears of the ViTDet / ConvNeXt trunks: this repo's igemm / wgrad kernels vs the library GEMM torch
dispatches to (hipBLASLt / rocBLAS) at the same shapes.  Forward y = x W^T, data gradient dx = g W, weight gradient dW = g^T x."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops  # noqa: E402


def t(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M, K, N in ((8400, 768, 2304), (8400, 768, 3072), (8400, 3072, 768), (16800, 768, 3072), (16800, 3072, 768), (67200, 192, 768), (4200, 1536, 6144)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    g = torch.randn(M, N, device="cuda").bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    mine_f = t(lambda: ops.conv2d(x.view(M, 1, 1, K), w.view(N, 1, 1, K)))
    lib_f = t(lambda: torch.matmul(x, w.t()))
    mine_w = t(lambda: ops.conv_wgrad(x.view(M, 1, 1, K), g.view(M, 1, 1, N), dw, KH=1, KW=1))
    lib_w = t(lambda: torch.matmul(g.t(), x))
    print("M=%6d K=%5d N=%5d  fwd: igemm %6.1f us (%4.0f TF/s)  lib %6.1f us (%4.0f TF/s) | wgrad: ours %6.1f us (%4.0f)  lib %6.1f us (%4.0f)" %
          (M, K, N, mine_f, fl / mine_f / 1e6, lib_f, fl / lib_f / 1e6, mine_w, fl / mine_w / 1e6, lib_w, fl / lib_w / 1e6))
