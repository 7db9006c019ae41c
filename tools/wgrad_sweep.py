"""Isolated weight-gradient shapes of the R50 step under kernel variants / ablations (tuning aid, not product code):
lean vs LDS-DMA form, with and without the atomic epilogue, and across split counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [(4, 50, 84, 256, 256, 3), (4, 50, 84, 1024, 256, 1), (4, 50, 84, 256, 1024, 1), (4, 100, 168, 128, 128, 3), (4, 25, 42, 512, 512, 3),
          (4, 100, 168, 128, 512, 1), (4, 200, 336, 256, 256, 3), (4, 100, 168, 256, 256, 3), (2048, 1, 1, 12544, 1024, 1)]
VARIANTS = [("lean", {}), ("lean-noatom", {"wgrad_dbg": 1}), ("dma", {"wgrad_dma": 2}), ("dma-noatom", {"wgrad_dma": 2, "wgrad_dbg": 1}),
            ("lean s192", {"wgrad_slots": 192}), ("lean s768", {"wgrad_slots": 768}), ("dma s192", {"wgrad_dma": 2, "wgrad_slots": 192}),
            ("dma s768", {"wgrad_dma": 2, "wgrad_slots": 768}), ("nobig lean", {"wgrad_big_min": 0}), ("nobig dma", {"wgrad_big_min": 0, "wgrad_dma": 2})]
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    gy = torch.randn(N, H, W, Cout, device="cuda", generator=g).bfloat16()
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    row = []
    for name, knobs in VARIANTS:
        L.reset_tuning()
        for kn, v in knobs.items():
            L.set_tuning(kn, v)
        run = lambda: ops.conv_wgrad(x, gy, dw, KH=k, KW=k, stride=1, pad=k // 2)
        run(); which = L.last_dispatch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("%s %.0fus %.0fTF [%s]" % (name, us, 2.0 * N * H * W * Cin * Cout * k * k / us / 1e6, which.replace("wgrad_bf16_", "")))
    print((N, H, W, Cin, Cout, k), " | ".join(row), flush=True)
L.reset_tuning()
