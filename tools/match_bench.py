"""micro-benchmark of the anchor matcher + label compaction at the benchmark size (268k anchors x 4 images), by GT count"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aldi_amd import ops
from aldi_amd.engine import GMAX, make_anchors
DEV = "cuda"
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
anchors = make_anchors(shapes, DEV)
N, sumA = 4, anchors.shape[0]
g = torch.Generator().manual_seed(3)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1000
for G in (0, 8, 100):
    gb = torch.zeros(N, GMAX, 4)
    for i in range(N):
        w = torch.rand(G, generator=g) * 200 + 20; h = torch.rand(G, generator=g) * 150 + 20
        x = torch.rand(G, generator=g) * 1100; y = torch.rand(G, generator=g) * 600
        gb[i, :G] = torch.stack([x, y, x + w, y + h], 1)
    cnt = torch.full((N,), G, dtype=torch.int32)
    gbd, cntd = gb.to(DEV), cnt.to(DEV)
    best_iou = torch.empty(N, sumA, device=DEV); best_idx = torch.empty(N, sumA, dtype=torch.int32, device=DEV)
    labels = torch.empty(N, sumA, dtype=torch.int32, device=DEV); scratch = ops.box_match_scratch(N, GMAX, DEV) if os.environ.get("PACKED") != "1" else torch.empty(N, GMAX, dtype=torch.int32, device=DEV)
    lists = torch.empty(N, 2, sumA, dtype=torch.int32, device=DEV); counts = torch.empty(N, 2, dtype=torch.int32, device=DEV)
    t1 = timeit(lambda: ops.box_match(anchors, 0, None, sumA, gbd, cntd, GMAX, N, 0.3, 0.7, True, best_iou, best_idx, scratch, labels))
    t2 = timeit(lambda: ops.compact_labels(labels, sumA, N, 0, lists, counts))
    print(f"G = {G:3d}: box_match (memset + iou + label) {t1:.1f} us, compact {t2:.1f} us, positives {counts[:, 0].tolist()}")
