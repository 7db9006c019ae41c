"""Isolated 1x1 / 3x3 conv shapes of the R50 step under each tile template (tuning aid): igemm_force 0 (heuristic) .. 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
# (N, H, W, Cin, Cout, k, res, relu, mask)
SHAPES = [(4, 50, 84, 256, 1024, 1, 1, 1, 0), (4, 50, 84, 1024, 256, 1, 0, 1, 0), (4, 50, 84, 1024, 256, 1, 0, 0, 1), (4, 50, 84, 256, 1024, 1, 1, 0, 1),
          (4, 100, 168, 128, 512, 1, 1, 1, 0), (4, 100, 168, 512, 128, 1, 0, 1, 0), (4, 200, 336, 64, 256, 1, 1, 1, 0), (4, 200, 336, 256, 64, 1, 0, 1, 0),
          (4, 25, 42, 512, 2048, 1, 1, 1, 0), (4, 25, 42, 2048, 512, 1, 0, 1, 0), (2, 50, 84, 256, 1024, 1, 1, 1, 0), (2, 50, 84, 1024, 256, 1, 0, 1, 0),
          (4, 50, 84, 256, 256, 3, 0, 1, 0), (2, 50, 84, 256, 256, 3, 0, 1, 0), (4, 100, 168, 128, 128, 3, 0, 1, 0), (4, 25, 42, 512, 512, 3, 0, 1, 0),
          (4, 200, 336, 256, 256, 1, 2, 0, 0)]
g = torch.Generator(device="cuda").manual_seed(0)
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for (N, H, W, Cin, Cout, k, res, relu, mask) in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(N, H // (2 if res == 2 else 1), W // (2 if res == 2 else 1), Cout, device="cuda", generator=g).bfloat16() if res else None
    m = torch.randn(N, H, W, Cout, device="cuda", generator=g).bfloat16() if mask else None
    sc = torch.rand(Cout, device="cuda") + 0.5
    row = []
    nby = 2 * (x.numel() + w.numel() + y.numel() + (r.numel() if res else 0) + (m.numel() if mask else 0))
    # a variant is igemm_force[:igemm_dbg] (igemm_dbg ablation bits of igemm_body: 4 = no epilogue, 8 = one K slab only, 16 = every DMA source inside one 4-KB window)
    for var in os.environ.get("SWEEP_FORCE", "0,1,2,3,4").split(","):
        force, dbg = (int(v) for v in (var + ":0").split(":")[:2])
        L.reset_tuning(); L.set_tuning("igemm_force", force); L.set_tuning("igemm_dbg", dbg)
        run = lambda: ops.conv2d(x, w, pad=k // 2, out=y, relu=bool(relu), res=r, res_mode=res, mask=m, scale=sc, shift=sc)
        run(); which = L.last_dispatch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("f%s %.1fus %.0fTF %.1fTB/s [%s]" % (var, us, 2.0 * N * H * W * Cin * Cout * k * k / us / 1e6, nby / us / 1e6, which.replace("igemm<bf16,", "<")))
    print((N, H, W, Cin, Cout, k, "res%d" % res, "relu%d" % relu, "mask%d" % mask), " | ".join(row), flush=True)
L.reset_tuning()
