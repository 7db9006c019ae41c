"""box head FC1 forward / dgrad shapes, WARM (back to back) and COLD (a 1-GB fill between launches evicts L2 and the Infinity Cache, as the rest of
the step does between two uses of the 26-MB weight): per-launch HIP-event times.  usage: python tools/fc1_cold.py [force ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def timed(run, cold, n=8):
    ts = []
    for _ in range(n):
        if cold:
            flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
variants = [v.split(":") for v in (sys.argv[1:] or ["0"])]          # igemm_force[:ksplit[:igemm_splitk_tile]]
SH = [("fc1 fwd", 2048, 12544, 1024), ("fc1 fwd teacher", 1024, 12544, 1024), ("fc1 dgrad", 2048, 1024, 12544), ("fc2", 2048, 1024, 1024)]
for name, M, K, Cout in SH:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16()
    w = (torch.randn(Cout, 1, 1, K, device="cuda") * 0.01).bfloat16()
    b = torch.randn(Cout, device="cuda")
    ref, row = None, []
    for v in variants:
        L.reset_tuning(); L.set_tuning("igemm_force", int(v[0]))
        ks = int(v[1]) if len(v) > 1 and v[1] != "" else None
        if len(v) > 2:
            L.set_tuning("igemm_splitk_tile", int(v[2]))
        y = torch.empty(M, 1, 1, Cout, device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.conv2d(x, w, shift=b, relu=True, out=y, ksplit=ks)
        try:
            run()
        except Exception as e:
            row.append("%s: %s" % (":".join(v), str(e)[:40])); continue
        which = L.last_dispatch()
        tw, tc = timed(run, False), timed(run, True)
        if ref is None:
            ref = y.clone()
        d = float((ref.float() - y.float()).abs().max())
        fl = 2.0 * M * K * Cout
        row.append("%s warm %.0fus %.0fTF cold %.0fus %.0fTF d=%.3g [%s]" % (":".join(v), tw, fl / tw / 1e6, tc, fl / tc / 1e6, d, which.replace("igemm<bf16,", "<")))
    print(name, (M, K, Cout), " | ".join(row), flush=True)
L.reset_tuning()
