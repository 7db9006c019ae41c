"""does the ORDER in which RoIAlign's workgroups visit the ROIs matter (L2 locality)?  the same ROI set in sampler order (random positions) and
sorted by (image, 64-pixel row band, x) -- forward and backward"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops
dev = "cuda"
Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for N, per in ((4, 512), (2, 1000)):
    g = torch.Generator().manual_seed(1)
    R = N * per
    w = (torch.rand(R, generator=g) * 100 + 28); h = (torch.rand(R, generator=g) * 50 + 15)
    cx = torch.rand(R, generator=g) * 1333; cy = torch.rand(R, generator=g) * 800
    img = torch.arange(R) // per
    rois = torch.stack([img.float(), (cx - w / 2).clamp(0, 1332), (cy - h / 2).clamp(0, 799), (cx + w / 2).clamp(1, 1333), (cy + h / 2).clamp(1, 800)], 1)
    key = img * 10**7 + (cy // 64).long() * 10**4 + cx.long()
    rois_sorted = rois[torch.argsort(key)]
    feats = [torch.randn((N, Hs[l], Ws[l], 256), device=dev).to(torch.bfloat16) for l in range(4)]
    rf = ops.make_roi_feats(feats, None, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    pooled = torch.empty((R, 7, 7, 256), dtype=torch.bfloat16, device=dev)
    for name, rr in (("sampler order", rois), ("sorted by position", rois_sorted)):
        rd = rr.to(dev).contiguous()
        t = [timeit(lambda: ops.roialign(rf, rd, R, 7, pooled, backward=False)) for _ in range(3)]
        print(f"fwd N={N} R={R} {name}: {min(t):.1f} us", flush=True)
    if N == 4:
        gp = torch.randn((R, 7, 7, 256), device=dev).to(torch.bfloat16)
        grads = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=torch.bfloat16, device=dev) for l in range(4)]
        rfg = ops.make_roi_feats(feats, grads, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
        for name, rr in (("sampler order", rois), ("sorted by position", rois_sorted)):
            rd = rr.to(dev).contiguous()
            t = [timeit(lambda: ops.roialign_backward(rfg, rd, R, 7, gp, N, rois_sorted=True, grad_dtype=torch.bfloat16)) for _ in range(3)]
            print(f"bwd N={N} R={R} {name}: {min(t):.1f} us", flush=True)
