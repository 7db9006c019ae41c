import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from aldi_amd import _lib as L, ops
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
SH = [("fc1 dgrad", 2048, 1024, 12544), ("fc1 dgrad teacher?", 1024, 1024, 12544), ("vit fc2", 16800, 3072, 768), ("vit fc1", 16800, 768, 3072), ("vit qkv", 16800, 768, 2304)]
for name, M, K, Cout in SH:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16()
    w = (torch.randn(Cout, 1, 1, K, device="cuda") * 0.01).bfloat16()
    row = []
    for tile in (0, 8):
        L.reset_tuning(); L.set_tuning("igemm_tile", tile)
        y = torch.empty(M, 1, 1, Cout, device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.conv2d(x, w, out=y, ksplit=0)
        run(); which = L.last_dispatch()
        ts = {}
        for cold in (False, True):
            tt = []
            for _ in range(8):
                if cold: flush.fill_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1) * 1e3)
            ts[cold] = statistics.median(tt)
        fl = 2.0 * M * K * Cout
        row.append("tile %d: warm %.0fus %.0fTF cold %.0fus %.0fTF [%s]" % (tile, ts[False], fl / ts[False] / 1e6, ts[True], fl / ts[True] / 1e6, which.replace("igemm<bf16,", "<")))
    print(name, (M, K, Cout), " | ".join(row), flush=True)
