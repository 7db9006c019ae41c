import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
def run(case, kind, force):
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    kw = dict(stride=stride, pad=pad)
    if kind == "f1": kw.update(scale=sc, shift=sh, relu=True)
    if kind == "sc": kw.update(scale=sc, shift=sh)
    outs = {}
    for d in (0, 7):
        L.reset_tuning(); L.set_tuning("igemm_direct", d); L.set_tuning("igemm_force", force)
        outs[d] = ops.conv2d(x, w, **kw).float(); nm = L.last_dispatch()
        torch.cuda.synchronize()
    a, b = outs[0].view(-1, Cout), outs[7].view(-1, Cout)
    bad = (a != b)
    print(case, kind, force, nm, "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().view(-1); cols = bad.any(0).nonzero().view(-1)
        print("  bad rows", rows[:40].tolist(), "... n", len(rows)); print("  bad cols", cols[:80].tolist(), "n", len(cols))
        r0, c0 = int(rows[0]), int(cols[0])
        print("  a", a[r0, c0:c0 + 8].tolist()); print("  b", b[r0, c0:c0 + 8].tolist())
for K in (32, 64, 96, 128):
    for kind in ("none", "sc"):
        run((3, 7, 9, K, 200, 1, 1, 0), kind, 2)
run((3, 7, 9, 64, 64, 1, 1, 0), "sc", 2)
run((1, 8, 16, 64, 64, 1, 1, 0), "sc", 2)
run((1, 8, 16, 64, 64, 1, 1, 0), "none", 2)
