"""Where a 3x3 halo conv's time goes: igemm_dbg ablation bits on the role-split kernel (igemm_force 10) and the lockstep one (4).
bits: 4 = no epilogue, 16 = all DMA sources in one 4-KB window (no memory traffic), 32 = no DMA in the K loop, 64 = no MFMAs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
N, H, W, Cin, Cout = [int(v) for v in os.environ.get("SHAPE", "4,200,336,256,256").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (Cin * 9) ** 0.5).bfloat16()
sc = torch.rand(Cout, device="cuda") + 0.5
y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
for force in [int(v) for v in os.environ.get("SWEEP_FORCE", "10,4").split(",")]:
    for dbg in [int(v) for v in os.environ.get("DBG", "0,4,16,20,32,36,64,68,96,100").split(",")]:
        L.reset_tuning(); L.set_tuning("igemm_force", force); L.set_tuning("igemm_dbg", dbg)
        run = lambda: ops.conv2d(x, w, pad=1, out=y, relu=True, scale=sc, shift=sc)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        print("force %2d dbg %3d: %7.1f us  (%s)" % (force, dbg, min(ts), L.last_dispatch()), flush=True)
L.reset_tuning()
