import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aldi_amd import _lib as L, ops
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, C) in [(2, 25, 42, 512), (4, 25, 42, 512), (2, 50, 84, 256), (2, 100, 168, 128)]:
    x = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) / (C * 9) ** 0.5).bfloat16()
    y = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    sc = torch.rand(C, device="cuda") + 0.5
    ref = None
    for halo, force in [(1, 0), (0, 0), (0, 2), (0, 3), (0, 1)]:
        L.reset_tuning(); L.set_tuning("igemm_halo", halo); L.set_tuning("igemm_force", force)
        run = lambda: ops.conv2d(x, w, pad=1, out=y, relu=True, scale=sc, shift=sc)
        run(); which = L.last_dispatch(); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = bool(torch.equal(ref, y)); md = float((ref.float() - y.float()).abs().max())
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        us = min(ts)
        print((N, H, W, C), "halo", halo, "force", force, "%.1f us %.0f TF" % (us, 2.0 * N * H * W * C * C * 9 / us / 1e6), which, "identical" if same else "maxdiff %.3g" % md, flush=True)
L.reset_tuning()
