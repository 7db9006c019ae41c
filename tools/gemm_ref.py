"""Yardstick: library GEMM (hipBLASLt via torch.matmul) at the GEMM-equivalent shapes of the conv layers."""
import torch
def t(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, N, K in ((268800, 256, 2304), (67200, 256, 2304), (16800, 256, 2304), (268800, 256, 64), (268800, 64, 256), (16800, 1024, 256), (8192, 8192, 8192), (2304, 256, 268800)):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    us = t(lambda: torch.matmul(a, b.t()))
    print("gemm M=%d N=%d K=%d: %.1f us %.1f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
