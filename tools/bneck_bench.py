"""res2 bottleneck: the fused kernel (aldi_bottleneck_fused) against the three igemm launches it replaces, isolated, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for N in (4, 2):
    for Cin in (256, 64):
        H, W = 200, 336
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
        res = x if Cin == 256 else torch.randn(N, H, W, 256, device="cuda", generator=g).bfloat16()
        w = [(torch.randn(s, device="cuda", generator=g) / (s[1] * s[2] * s[3]) ** 0.5) for s in ((64, 1, 1, Cin), (64, 3, 3, 64), (256, 1, 1, 64))]
        sc = [torch.rand(c, device="cuda") + 0.5 for c in (64, 64, 256)]
        sh = [torch.randn(c, device="cuda") * 0.1 for c in (64, 64, 256)]
        plan = ops.FoldWeightsPlan(list(zip(w, sc))); plan.run()
        wb = [t.bfloat16() for t in w]
        y = torch.empty(N, H, W, 256, device="cuda", dtype=torch.bfloat16)

        def fused():
            ops.bottleneck_fused(x, res, *plan.out, *sh, out=y)

        def layers():
            a1 = ops.conv2d(x, wb[0], scale=sc[0], shift=sh[0], relu=True)
            a2 = ops.conv2d(a1, wb[1], pad=1, scale=sc[1], shift=sh[1], relu=True)
            ops.conv2d(a2, wb[2], scale=sc[2], shift=sh[2], relu=True, res=res, res_mode=1, out=y)
        for name, fn in (("fused", fused), ("3 launches", layers)):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 100)
            us = min(ts)
            fl = 2.0 * N * H * W * (Cin * 64 + 576 * 64 + 64 * 256)
            by = N * H * W * (Cin + 256 + 256) * 2
            print("N=%d Cin=%3d %-10s %7.1f us  %6.0f TFLOP/s  %5.2f TB/s (x + res + y once)" % (N, Cin, name, us, fl / us / 1e6, by / us / 1e6), flush=True)
