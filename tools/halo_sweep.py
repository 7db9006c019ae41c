"""3x3 'same' conv shapes of the R50 step under the halo tile templates (igemm_force values given in SWEEP_FORCE), each checked
against the first listed template's output.  usage: SWEEP_FORCE=4,9,1,2 python tools/halo_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [(4, 200, 336, 256, 256), (2, 200, 336, 256, 256), (4, 100, 168, 256, 256), (2, 100, 168, 256, 256), (4, 50, 84, 256, 256), (4, 100, 168, 128, 128),
          (4, 25, 42, 512, 512), (4, 200, 336, 64, 64)]
forces = [int(v) for v in os.environ.get("SWEEP_FORCE", "4,9").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Cin, Cout) in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (Cin * 9) ** 0.5).bfloat16()
    sc = torch.rand(Cout, device="cuda") + 0.5
    ref, row = None, []
    for force in forces:
        L.reset_tuning(); L.set_tuning("igemm_force", force)
        y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.conv2d(x, w, pad=1, out=y, relu=True, scale=sc, shift=sc)
        run(); which = L.last_dispatch()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        us = min(ts)
        if ref is None:
            ref = y.clone()
        ok = torch.equal(ref, y)
        row.append("f%d %.1fus %.0fTF %s [%s]" % (force, us, 2.0 * N * H * W * Cin * Cout * 9 / us / 1e6, "same" if ok else "DIFF %.3g" % float((ref.float() - y.float()).abs().max()),
                                                  which.replace("igemm<bf16,", "<")))
    print((N, H, W, Cin, Cout), " | ".join(row), flush=True)
L.reset_tuning()
