"""aldi_rpn_proposals at the step's size (N images of 800 x 1344: pyramid 200x336 .. 13x21, 3 anchors): whole chain and per-kernel times, alone on the
chip, variants of tuning knobs interleaved.  usage: VARIANTS="rpn_topk_fused=0;rpn_topk_fused=1" [N=4] python tools/proposals_bench.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import _lib as L, ops
from aldi_amd.engine import make_anchors
N = int(os.environ.get("N", "4"))
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
A, C = 3, 16
g = torch.Generator(device="cuda").manual_seed(0)
heads = []
for (h, w) in shapes:
    t = torch.randn(N, h, w, C, device="cuda", generator=g)
    t[..., A:] *= 0.3
    heads.append(t)
anchors = make_anchors(shapes, "cuda")
geom = ops.make_geom(shapes, A, C)
hw = torch.tensor([(800, 1333)] * N, dtype=torch.int32, device="cuda")
ws = torch.empty(ops.rpn_proposals_workspace(N, 5), dtype=torch.uint8, device="cuda")
boxes, scores = torch.empty(N, 1000, 4, device="cuda"), torch.empty(N, 1000, device="cuda")
count, err = torch.empty(N, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
variants = [[kv.split("=") for kv in v.split(",") if kv] for v in os.environ.get("VARIANTS", "rpn_topk_fused=0;rpn_topk_fused=1").split(";")]
def setup(v):
    L.reset_tuning()
    for k, x in v:
        L.set_tuning(k, int(x))
run = lambda: ops.rpn_proposals(geom, heads, anchors, hw, N, 2000, 1000, 0.7, ws, boxes, scores, count, err)
times, outs = [[] for _ in variants], []
for v in variants:
    setup(v); run(); torch.cuda.synchronize()
    outs.append((boxes.clone(), scores.clone(), count.clone(), int(err)))
for _ in range(9):
    for i, v in enumerate(variants):
        setup(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3 / 5)
for v, t, o in zip(variants, times, outs):
    same = all(torch.equal(a, b) for a, b in zip(o[:3], outs[0][:3]))
    print("%-40s %7.1f us (min %.1f)  err %d  %s  kept %s" % (",".join("=".join(kv) for kv in v), statistics.median(t), min(t), o[3], "same as first" if same else "DIFFERENT", o[2].tolist()))
L.reset_tuning()
