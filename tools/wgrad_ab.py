"""weight-gradient shapes of the step, single launches, variants of tuning knobs interleaved in one process (median of rounds).
usage: VARIANTS="wgrad_dma64=0;wgrad_dma64=3" python tools/wgrad_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [(4, 200, 336, 256, 256, 3), (4, 100, 168, 256, 256, 3), (4, 50, 84, 256, 256, 3), (4, 50, 84, 1024, 256, 1), (4, 50, 84, 256, 1024, 1), (4, 25, 42, 512, 512, 3),
          (4, 100, 168, 128, 128, 3), (4, 100, 168, 512, 128, 1), (2048, 1, 1, 12544, 1024, 1)]
variants = [[kv.split("=") for kv in v.split(",") if kv] for v in os.environ.get("VARIANTS", "wgrad_dma64=0;wgrad_dma64=3").split(";")]
rounds, reps = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("REPS", "5"))
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    gy = (torch.randn(N, H, W, Cout, device="cuda", generator=g) * 0.1).bfloat16()
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    times, names = [[] for _ in variants], []
    def setup(v):
        L.reset_tuning()
        for kk, vv in v:
            L.set_tuning(kk, int(vv))
    for v in variants:
        setup(v)
        ops.conv_wgrad(x, gy, dw, KH=k, KW=k, stride=1, pad=k // 2)
        names.append(L.last_dispatch())
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, v in enumerate(variants):
            setup(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv_wgrad(x, gy, dw, KH=k, KW=k, stride=1, pad=k // 2)
            e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) * 1e3 / reps)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    print((N, H, W, Cin, Cout, k), " | ".join("%s: %.1fus %.0fTF [%s]" % (",".join("=".join(kv) for kv in v), statistics.median(t), fl / statistics.median(t) / 1e6, n) for v, t, n in zip(variants, times, names)), flush=True)
L.reset_tuning()
