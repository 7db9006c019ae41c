"""fp32 weight gradients at the Deformable-DETR step's shapes: 64x64 tile kernel vs the 128x128 one (knob wgrad_f32_tile128)."""
import torch
from aldi_amd import _lib as L, ops

CASES = [(44646, 1, 1, 256, 1024, 1, 1, 0), (44646, 1, 1, 1024, 256, 1, 1, 0), (44646, 1, 1, 256, 256, 1, 1, 0), (44646, 1, 1, 256, 384, 1, 1, 0),
         (2, 50, 84, 256, 256, 3, 1, 1), (2, 100, 168, 128, 128, 3, 1, 1), (2, 25, 42, 512, 512, 3, 1, 1), (2, 50, 84, 1024, 256, 1, 1, 0), (2, 50, 84, 256, 1024, 1, 1, 0)]
for case in CASES:
    N, H, W, Cin, Cout, k, s, p = case
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Cin, device="cuda")
    g = torch.randn(N, Ho, Wo, Cout, device="cuda")
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    row = []
    for knob in (0, 1):
        L.reset_tuning()
        L.set_tuning("wgrad_f32_tile128", knob)
        for _ in range(3):
            ops.conv_wgrad(x, g, dw, KH=k, KW=k, stride=s, pad=p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_wgrad(x, g, dw, KH=k, KW=k, stride=s, pad=p)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append("%s %7.1f us %6.1f TF/s" % (L.last_dispatch(), us, 2.0 * N * Ho * Wo * Cout * k * k * Cin / us / 1e6))
    print(case, " | ".join(row))
