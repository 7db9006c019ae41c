"""micro-benchmark of the RoIAlign backward (gather form) on the benchmark step's ROI distribution (tools/roi_stats.py):
512 ROIs per image, nearly all on P2, ~19 x 10 pixels at the level.  `flags`: 1 = ROI rows sorted by image (the engine's call), 0 = unsorted.
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops, _lib as L
import ctypes as C
torch.manual_seed(0)
dev = "cuda"
N, R = 4, 2048
Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
g = torch.Generator().manual_seed(1)
w = (torch.rand(R, generator=g) * 100 + 28); h = (torch.rand(R, generator=g) * 50 + 15)
cx = torch.rand(R, generator=g) * 1333; cy = torch.rand(R, generator=g) * 800
rois = torch.stack([torch.arange(R) // 512, (cx - w / 2).clamp(0, 1332), (cy - h / 2).clamp(0, 799), (cx + w / 2).clamp(1, 1333), (cy + h / 2).clamp(1, 800)], 1).float().to(dev)
feats = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=torch.bfloat16, device=dev) for l in range(4)]
grads = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=torch.float32, device=dev) for l in range(4)]
gp = torch.randn((R, 7, 7, 256), device=dev).to(torch.bfloat16)
rf = ops.make_roi_feats(feats, grads, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
def run(flags):
    L.call("aldi_roialign_backward", C.byref(rf), ops._p(rois), R, 7, ops._p(gp), N, flags, ops.dtype_code(gp.dtype), ops.dtype_code(grads[0].dtype), ops.stream_ptr())
for flags in (1, 0):
    for _ in range(3): run(flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(flags)
    e1.record(); torch.cuda.synchronize()
    print(f"flags {flags:2d}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")
