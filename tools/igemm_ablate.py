"""where a short-K 1x1 layer's time goes: igemm_dbg ablation bits (4 = no epilogue, 8 = one K slab only, 16 = every DMA source inside one 4-KB window)
on the direct-epilogue kernels, back-to-back launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
g = torch.Generator(device="cuda").manual_seed(0)
SH = [(4, 50, 84, 256, 1024, "f3"), (4, 50, 84, 1024, 256, "f1"), (4, 100, 168, 128, 512, "f3"), (4, 25, 42, 512, 2048, "f3"), (2, 50, 84, 256, 1024, "f3")]
for (N, H, W, Cin, Cout, kind) in SH:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) / Cin ** 0.5).bfloat16()
    r = torch.randn(N, H, W, Cout, device="cuda", generator=g).bfloat16()
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    bits = torch.empty(N * H * W * Cout // 8, dtype=torch.uint8, device="cuda")
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    kw = dict(scale=sc, shift=sh, relu=True, bits_out=bits, out=y)
    if kind == "f3":
        kw.update(res=r, res_mode=1)
    row = []
    for dbg in (0, 4, 8, 12, 16, 20):
        ts = []
        for rd in range(5):
            L.reset_tuning(); L.set_tuning("igemm_dbg", dbg)
            ops.conv2d(x, w, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d(x, w, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 50)
        row.append("dbg%-2d %5.1f us [%s]" % (dbg, sorted(ts)[2], L.last_dispatch().replace("igemm<bf16,", "<")[:34]))
    print((N, H, W, Cin, Cout, kind), " | ".join(row), flush=True)
L.reset_tuning()
