"""HBM bandwidth yardsticks with torch elementwise kernels (fill / copy / add) at conv-output sizes."""
import torch
def t(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for mb in (34, 137, 550):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device="cuda", dtype=torch.bfloat16); b = torch.ones_like(a); c = torch.ones_like(a)
    us = t(lambda: a.fill_(1.0)); print("fill  %4d MB: %7.1f us  %.2f TB/s (write)" % (mb, us, mb / us))
    us = t(lambda: a.copy_(b)); print("copy  %4d MB: %7.1f us  %.2f TB/s (r+w total)" % (mb, us, 2 * mb / us))
    us = t(lambda: torch.add(b, c, out=a)); print("add   %4d MB: %7.1f us  %.2f TB/s (2r+w total)" % (mb, us, 3 * mb / us))
    us = t(lambda: b.sum()); print("sum   %4d MB: %7.1f us  %.2f TB/s (read)" % (mb, us, mb / us))
