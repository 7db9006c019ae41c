"""fp32 igemm tiles at the Deformable-DETR step's shapes: default rule vs forced 128x128 / 128x64 / 64x64 (igemm_force 1 / 2 / 3)"""
import torch
from aldi_amd import _lib as L, ops
CASES = [(2, 200, 336, 256, 256, 3), (4, 200, 336, 256, 256, 3), (4, 50, 84, 256, 256, 3), (4, 100, 168, 128, 512, 1), (2, 50, 84, 256, 256, 3), (2, 100, 168, 128, 128, 3), (2, 25, 42, 512, 512, 3), (2, 200, 336, 64, 64, 3), (2, 25, 42, 2048, 256, 3),
         (2, 50, 84, 1024, 256, 1), (2, 50, 84, 256, 1024, 1), (2, 100, 168, 512, 128, 1), (2, 100, 168, 128, 512, 1), (2, 25, 42, 512, 2048, 1), (2, 25, 42, 2048, 512, 1),
         (2, 200, 336, 64, 256, 1), (44646, 1, 1, 256, 256, 1), (44646, 1, 1, 1024, 256, 1), (44646, 1, 1, 256, 1024, 1), (44646, 1, 1, 256, 384, 1), (600, 1, 1, 256, 256, 1)]
for (N, H, W, Cin, Cout, k) in CASES:
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k, k, Cin, device="cuda") * (k * k * Cin) ** -0.5
    b = torch.randn(Cout, device="cuda")
    row = []
    for force in (0, 1, 2, 3, 4):
        L.reset_tuning()
        if force:
            L.set_tuning("igemm_force", force)
        y = ops.conv2d(x, w, pad=k // 2, shift=b, relu=True)
        which = L.last_dispatch()
        for _ in range(2):
            ops.conv2d(x, w, pad=k // 2, shift=b, relu=True, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, w, pad=k // 2, shift=b, relu=True, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("%s %.0f us" % (which.replace("igemm<f32,", "<").replace(",pipe,tap>", ">").replace(",flat,tap>", ">"), us))
    L.reset_tuning()
    print((N, H, W, Cin, Cout, k), " | ".join(row))
