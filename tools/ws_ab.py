"""The short-K 1x1 layers of the R50 step on the weight-stationary persistent kernel (csrc/igemm_ws.h) against the tile kernels: INTERLEAVED in one
process (variant x round; median per variant), outputs compared (values and ReLU bits).  A variant is a list of knob=value pairs:
  VARIANTS="igemm_ws=0;igemm_ws=1;igemm_ws=1,igemm_ws_wgs=768" python tools/ws_ab.py
A shape is N,H,W,Cin,Cout,res,relu,maskbits,bitsout,scale (flags 0/1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [  # forward: conv1 (reduce), conv3 (expand + residual), shortcut-like expand; backward: conv1^T (expand + residual + mask bits), conv3^T (reduce + mask bits)
    (4, 100, 168, 512, 128, 0, 1, 0, 1, 1), (4, 100, 168, 128, 512, 1, 1, 0, 1, 1), (4, 100, 168, 128, 512, 1, 0, 1, 0, 0), (4, 100, 168, 512, 128, 0, 0, 1, 0, 0),
    (4, 50, 84, 256, 1024, 1, 1, 0, 1, 1), (4, 50, 84, 256, 1024, 1, 0, 1, 0, 0), (2, 100, 168, 128, 512, 1, 1, 0, 0, 1), (2, 100, 168, 512, 128, 0, 1, 0, 0, 1),
    (2, 50, 84, 256, 1024, 1, 1, 0, 0, 1), (4, 200, 336, 64, 256, 0, 0, 0, 0, 1), (4, 100, 168, 512, 256, 0, 0, 0, 0, 0), (4, 25, 42, 512, 2048, 1, 1, 0, 1, 1)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv) for v in os.environ.get("VARIANTS", "igemm_ws=0;igemm_ws=1").split(";")]
rounds, reps = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("REPS", "5"))
g = torch.Generator(device="cuda").manual_seed(0)


def setup(v):
    L.reset_tuning()
    for k, val in v.items():
        L.set_tuning(k, val)


for (N, H, W, Cin, Cout, res, relu, mbits, bout, scale) in SHAPES:
    M = N * H * W
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) / Cin ** 0.5).bfloat16()
    r = torch.randn(N, H // (2 if res == 2 else 1), W // (2 if res == 2 else 1), Cout, device="cuda", generator=g).bfloat16() if res else None      # (res 2: the coarser level's map, upsampled)
    mb = torch.randint(0, 256, (M * Cout // 8,), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8) if mbits else None
    sc = (torch.rand(Cout, device="cuda", generator=g) + 0.5) if scale else None
    sh = torch.randn(Cout, device="cuda", generator=g) * 0.1 if scale else None
    ys, bs, names, times = [], [], [], [[] for _ in variants]
    for v in variants:
        setup(v)
        y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
        b = torch.zeros(M * Cout // 8, device="cuda", dtype=torch.uint8) if bout else None
        ops.conv2d(x, w, out=y, relu=bool(relu), res=r, res_mode=res, scale=sc, shift=sh, mask_bits=mb, bits_out=b)
        names.append(L.last_dispatch().replace("igemm<bf16,", "<"))
        ys.append(y); bs.append(b)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, v in enumerate(variants):
            setup(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d(x, w, out=ys[k], relu=bool(relu), res=r, res_mode=res, scale=sc, shift=sh, mask_bits=mb, bits_out=bs[k])
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / reps)
    nby = 2 * (x.numel() + w.numel() + ys[0].numel() + (r.numel() if res else 0)) + (mb.numel() if mbits else 0) + (bs[0].numel() if bout else 0)
    row = []
    for k, v in enumerate(variants):
        d = float((ys[0].float() - ys[k].float()).abs().max())
        nb = int((bs[0] != bs[k]).sum()) if bout else 0
        med = statistics.median(times[k])
        row.append("%.1fus %.0fTF %.2fTB/s %s%s [%s]" % (med, 2.0 * M * Cin * Cout / med / 1e6, nby / med / 1e6, "same" if d == 0 else "maxdiff %.3g" % d,
                                                       ", %d bit bytes differ" % nb if nb else "", names[k]))
    print((N, H, W, Cin, Cout, "res%d relu%d mbits%d bout%d scale%d" % (res, relu, mbits, bout, scale)), " | ".join(row), flush=True)
L.reset_tuning()
