"""Tile templates on the box head's FC shapes (tuning aid): the 64x64 128-byte-slab form is the fastest for FC1 (M = 2048, K = 12544)
at ~100 us / 500 TFLOP/s -- 512 workgroups moving 1.6 GB from L2; a split-K 128x128 form is the next step."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from aldi_amd import _lib as L, ops
g = torch.Generator(device="cuda").manual_seed(0)
for (M, K, N) in [(2048, 12544, 1024), (1024, 12544, 1024), (2048, 1024, 1024), (2048, 1024, 12544)]:
    x = torch.randn(M, 1, 1, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, 1, 1, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    sh = torch.rand(N, device="cuda")
    row = []
    for force in (0, 1, 2, 3, 4, 6, 7, 8):
        L.reset_tuning(); L.set_tuning("igemm_force", force)
        run = lambda: ops.conv2d(x, w, relu=True, shift=sh)
        run(); which = L.last_dispatch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("f%d %.0fus %.0fTF [%s]" % (force, us, 2.0 * M * K * N / us / 1e6, which.replace("igemm<bf16,", "<")))
    print((M, K, N), " | ".join(row), flush=True)
L.reset_tuning()
