"""Layer-scale residual (ConvNeXt block tail) micro-benchmark: us per call and effective HBM rate, forward and backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import vit_ops as V
def timed(fn, reps=30):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, C in ((268800, 192), (67200, 384), (16800, 768), (4200, 1536)):
    x, y, g = (torch.randn(rows, C, device="cuda").bfloat16() for _ in range(3))
    gamma = torch.randn(C, device="cuda"); dg = torch.zeros(C, device="cuda")
    tf = timed(lambda: V.scale_add(x, y, gamma, None, rows)); tb = timed(lambda: V.scale_add_backward(g, y, gamma, None, dg, rows))
    mb = rows * C * 2 / 1e6
    print("rows %6d C %4d: fwd %6.1f us (%.2f TB/s)  bwd %6.1f us (%.2f TB/s)" % (rows, C, tf, 3 * mb / tf, tb, 3 * mb / tb))
