"""box head FC1 (K = 12544 -> 1024) forward: split-K slices x tile of the split-K launch (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
w = (torch.randn(1024, 1, 1, 12544, device="cuda") * 0.01).bfloat16()
b = torch.randn(1024, device="cuda")
for M in (1024, 2000, 2048):
    x = torch.randn(M, 1, 1, 12544, device="cuda").bfloat16()
    row = []
    for tile in (0, 1):
        for ks in (0, 2, 4, 7, 8, 14):
            L.reset_tuning(); L.set_tuning("igemm_splitk_tile", tile)
            try:
                run = lambda: ops.conv2d(x, w, shift=b, relu=True, ksplit=ks)
                run()
            except Exception as e:
                row.append("t%d ks%d: -" % (tile, ks)); continue
            which = L.last_dispatch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            row.append("t%d ks%d %.0fus %.0fTF" % (tile, ks, us, 2.0 * M * 12544 * 1024 / us / 1e6))
    print(M, " | ".join(row), flush=True)
L.reset_tuning()
