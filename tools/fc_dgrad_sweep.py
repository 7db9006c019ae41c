"""forced tile templates on the GEMM shapes of the box head (tuning aid): FC1 forward / dgrad, FC2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
SHAPES = [(2048, 1024, 12544), (1024, 1024, 12544), (2048, 12544, 1024), (2000, 12544, 1024), (2048, 1024, 1024), (16800, 256, 1024), (16800, 1024, 256), (67200, 128, 512), (67200, 512, 128)]
for (M, K, Cout) in SHAPES:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16()
    w = (torch.randn(Cout, 1, 1, K, device="cuda") * 0.02).bfloat16()
    row = []
    for force in (0, 1, 2, 3, 4, 6, 7, 8):
        L.reset_tuning(); L.set_tuning("igemm_force", force)
        os.environ["ALDI_SPLITK"] = "0"
        run = lambda: ops.conv2d(x, w)
        try:
            run()
        except Exception as e:
            row.append("f%d: -" % force); continue
        which = L.last_dispatch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("f%d %.0fus %.0fTF %s" % (force, us, 2.0 * M * K * Cout / us / 1e6, which.replace("igemm<bf16,", "<")))
    print((M, K, Cout), " | ".join(row), flush=True)
L.reset_tuning()
