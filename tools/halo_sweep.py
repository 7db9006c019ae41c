"""3x3 'same' conv shapes of the R50 step under the halo tile templates, INTERLEAVED in one process (variant x round, median and min per
variant: single runs on this pool differ by up to 10 % between boxes and by 3 % between launches), each checked against the first listed
variant's output.  A variant is igemm_force[:igemm_direct[:igemm_dbg]], e.g. SWEEP_FORCE=1,4,11,11:7 python tools/halo_sweep.py
(1 = 128x128, 4 = 256x128, 9 = 240x128, 10 = role-split 256x128, 11 = 256x256 with 128-byte K slabs; 11:7 = the same with the staged epilogue)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [(4, 200, 336, 256, 256), (2, 200, 336, 256, 256), (4, 100, 168, 256, 256), (2, 100, 168, 256, 256), (4, 50, 84, 256, 256), (4, 100, 168, 128, 128),
          (4, 25, 42, 512, 512), (4, 200, 336, 64, 64)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["SHAPES"].split(";")]
variants = [v.split(":") for v in os.environ.get("SWEEP_FORCE", "4,9").split(",")]
rounds, reps = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("REPS", "5"))
relu_data = os.environ.get("RELU_DATA", "0") == "1"       # inputs like the step's: a ReLU output (half zeros) instead of randn
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Cin, Cout) in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    if relu_data:
        x = x.clamp_min(0)
    x = x.bfloat16()
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (Cin * 9) ** 0.5).bfloat16()
    sc = torch.rand(Cout, device="cuda") + 0.5
    ys, names, times = [], [], [[] for _ in variants]
    def setup(v):
        L.reset_tuning(); L.set_tuning("igemm_force", int(v[0]))
        if len(v) > 1:
            L.set_tuning("igemm_direct", int(v[1]))
        if len(v) > 2:
            L.set_tuning("igemm_dbg", int(v[2]))
    for v in variants:
        setup(v)
        y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
        ops.conv2d(x, w, pad=1, out=y, relu=True, scale=sc, shift=sc)
        names.append(L.last_dispatch().replace("igemm<bf16,", "<"))
        ys.append(y)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, v in enumerate(variants):
            setup(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d(x, w, pad=1, out=ys[k], relu=True, scale=sc, shift=sc)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / reps)
    row = []
    fl = 2.0 * N * H * W * Cin * Cout * 9
    for k, v in enumerate(variants):
        d = float((ys[0].float() - ys[k].float()).abs().max())
        rel = d / float(ys[0].float().abs().max())
        med, mn = statistics.median(times[k]), min(times[k])
        row.append("%s %.1f/%.1fus %.0fTF %s [%s]" % (":".join(v), med, mn, fl / med / 1e6, "same" if d == 0 else "maxdiff %.3g (%.2g of max)" % (d, rel), names[k]))
    print((N, H, W, Cin, Cout), " | ".join(row), flush=True)
L.reset_tuning()
