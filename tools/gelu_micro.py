import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aldi_amd import vit_ops as V
import torch.nn.functional as F
def timed(fn, reps=30):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, C in ((16800, 3072), (268800, 768), (4200, 6144)):
    x = (torch.randn(rows, C, device="cuda") * 2).bfloat16(); g = torch.randn_like(x)
    tf = timed(lambda: V.gelu(x)); tb = timed(lambda: V.gelu_backward(x, g))
    mb = x.numel() * 2 / 1e6
    print("rows %d C %d: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (rows, C, tf, 2 * mb / tf, tb, 3 * mb / tb))
x = torch.linspace(-12, 12, 1 << 20, device="cuda"); xr = x.clone().requires_grad_(True)
y = F.gelu(xr); y.backward(torch.ones_like(y))
print("fp32 max abs err fwd %.2e bwd %.2e" % ((V.gelu(x.view(1024, -1)).view(-1) - y).abs().max().item(), (V.gelu_backward(x.view(1024, -1), torch.ones_like(x).view(1024, -1)).view(-1) - xr.grad).abs().max().item()))
