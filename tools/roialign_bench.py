"""RoIAlign forward (separable form, with / without the row look-ahead) and backward (gather form) on the benchmark step's ROI distribution
(tools/roi_stats.py): 512 sampled ROIs per image (student, 4 images) or 1000 proposals per image (teacher, 2 images), nearly all on P2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops, _lib as L
dev = "cuda"
Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
def mk(N, per, seed):
    g = torch.Generator().manual_seed(seed)
    R = N * per
    w = (torch.rand(R, generator=g) * 100 + 28); h = (torch.rand(R, generator=g) * 50 + 15)
    cx = torch.rand(R, generator=g) * 1333; cy = torch.rand(R, generator=g) * 800
    rois = torch.stack([torch.arange(R) // per, (cx - w / 2).clamp(0, 1332), (cy - h / 2).clamp(0, 799), (cx + w / 2).clamp(1, 1333), (cy + h / 2).clamp(1, 800)], 1).float().to(dev)
    feats = [torch.randn((N, Hs[l], Ws[l], 256), device=dev).to(torch.bfloat16) for l in range(4)]
    return R, rois, feats
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for N, per in ((4, 512), (2, 1000), (2, 512)):
    R, rois, feats = mk(N, per, 1)
    rf = ops.make_roi_feats(feats, None, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    pooled = torch.empty((R, 7, 7, 256), dtype=torch.bfloat16, device=dev)
    outs = {}
    for knob in (2, 1):
        L.reset_tuning(); L.set_tuning("roialign_sep", knob)
        ops.roialign(rf, rois, R, 7, pooled, backward=False); torch.cuda.synchronize()
        outs[knob] = pooled.clone()
        t = [timeit(lambda: ops.roialign(rf, rois, R, 7, pooled, backward=False)) for _ in range(3)]
        print(f"fwd N={N} R={R} roialign_sep={knob}: {min(t):.1f} us (median {sorted(t)[1]:.1f})", flush=True)
    print("   identical:", bool(torch.equal(outs[1], outs[2])))
L.reset_tuning()
N, per = 4, 512
R, rois, feats = mk(N, per, 1)
gp = torch.randn((R, 7, 7, 256), device=dev).to(torch.bfloat16)
for gd in (torch.bfloat16, torch.float32):
    grads = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=gd, device=dev) for l in range(4)]
    rf = ops.make_roi_feats(feats, grads, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    t = [timeit(lambda: ops.roialign_backward(rf, rois, R, 7, gp, N, rois_sorted=True, grad_dtype=gd)) for _ in range(3)]
    print(f"bwd N={N} R={R} grads {gd}: {min(t):.1f} us (median {sorted(t)[1]:.1f})", flush=True)
