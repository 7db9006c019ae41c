"""Tile-count quantisation of the mid-size 3x3 layers: the same conv at neighbouring pixel counts (W varied), us per launch and us per 1000 pixels.
res4 conv2 (4 x 50 x W x 256 -> 256) and res5 conv2 (4 x 25 x W x 512 -> 512); 128 x 64 halo tiles: workgroups = ceil(pixels / 128) x Cout / 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import ops, _lib as L
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best
g = torch.Generator(device="cuda").manual_seed(0)
SH = ((4, 50, 256, (72, 76, 78, 80, 81, 82, 84, 88, 96)), (4, 25, 512, (32, 36, 40, 41, 42, 44, 48)), (2, 50, 256, (76, 80, 81, 82, 84, 88)), (2, 25, 512, (36, 40, 41, 42, 48)))
if os.environ.get("STEP_SHAPES"):
    SH = ((4, 50, 256, (84,)), (2, 50, 256, (84,)), (4, 25, 512, (42,)), (2, 25, 512, (42,)), (4, 100, 128, (168,)), (2, 100, 128, (168,)))
for (N, H, C, Ws) in SH:
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) / (9 * C) ** 0.5).bfloat16()
    sh = torch.randn(C, device="cuda") * 0.1
    for W in Ws:
        x = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
        y = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
        L.reset_tuning()
        us = t(lambda: ops.conv2d(x, w, pad=1, out=y, relu=True, shift=sh))
        name = L.last_dispatch()
        px = N * H * W
        wgs = -(-px // 128) * (C // 64)
        extra = ""
        for f in [int(v) for v in os.environ.get("ALT_FORCE", "16").split(",") if v]:
            L.set_tuning("igemm_force", f)
            y2 = torch.empty_like(y)
            us2 = t(lambda: ops.conv2d(x, w, pad=1, out=y2, relu=True, shift=sh))
            extra += "  | force %d: %6.1f us (max diff %.3g) %s" % (f, us2, float((y2.float() - y.float()).abs().max()), L.last_dispatch())
            L.reset_tuning()
        print("N=%d %dx%d C=%d: %6d px %4d workgroups (%.2f per CU)  %6.1f us  %6.2f us / 1000 px  %s%s" % (N, H, W, C, px, wgs, wgs / 256, us, us * 1000 / px, name, extra), flush=True)
