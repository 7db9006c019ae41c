"""fp32 3x3 convolutions (parity mode / the Deformable-DETR step's trunk): tap form vs halo form (knob igemm_halo_f32)"""
import torch
from aldi_amd import _lib as L, ops
CASES = [(2, 200, 336, 256, 256), (2, 100, 168, 256, 256), (2, 50, 84, 256, 256), (4, 50, 84, 256, 256), (2, 100, 168, 128, 128), (4, 100, 168, 128, 128), (2, 25, 42, 512, 512), (4, 25, 42, 512, 512),
         (2, 200, 336, 64, 64), (2, 25, 42, 2048, 256)]
for (N, H, W, Cin, Cout) in CASES:
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5
    b = torch.randn(Cout, device="cuda")
    row, ys = [], []
    for knob in (0, 1):
        L.reset_tuning()
        L.set_tuning("igemm_halo_f32", 1 if knob else 0)
        y = ops.conv2d(x, w, pad=1, shift=b, relu=True)
        which = L.last_dispatch()
        for _ in range(2):
            ops.conv2d(x, w, pad=1, shift=b, relu=True, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, w, pad=1, shift=b, relu=True, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        ys.append(y.clone())
        row.append("%s %.1f us %.1f TF/s" % (which, us, 2.0 * N * H * W * Cout * 9 * Cin / us / 1e6))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).relu().permute(0, 2, 3, 1)
    errs = [((y.double() - ref).abs().max() / ref.abs().max()).item() for y in ys]
    print((N, H, W, Cin, Cout), " | ".join(row), "| err vs fp64 %.1e %.1e" % tuple(errs))
