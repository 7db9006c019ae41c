"""Host cost of ENQUEUEING the static launch sequences (no GPU wait inside the timed region)."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aldi_amd import synthetic as syn
from aldi_amd.trainer import ALDITrainer
cfg = bench.make_cfg(1, 800, 1333, False); cfg.SOLVER.FUSED_STEP = True
random.seed(1); torch.manual_seed(1)
tr = ALDITrainer(cfg)
eng = tr.model.engine
data = syn.make_batch(2, 2, 800, 1333, 8, seed=100)
imgs = [d["image"].cuda() for d in data[1]] + [d["image"].cuda() for d in data[3]]
st, sizes, hw = eng.stage_images(imgs)
shapes, geom, anchors = eng.geometry(st.shape[2], st.shape[3])
for name, fn in (("trunk N=4", lambda: eng.trunk(st, sizes, save=True)),):
    for _ in range(3): c = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t = time.perf_counter(); c = fn(); ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    t = time.perf_counter(); c = fn(); torch.cuda.synchronize(); tg = time.perf_counter() - t
    print("%-12s host enqueue %.2f ms (min %.2f) | enqueue+GPU %.2f ms | launches ~%d" % (name, 1e3 * sum(ts) / len(ts), 1e3 * min(ts), 1e3 * tg, 53 * 2 + 10))
c.N, c.sizes, c.hw, c.geom, c.anchors, c.shapes = 4, sizes, hw, geom, anchors, shapes
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t = time.perf_counter(); eng.rpn_head(c, save=True); ts.append(time.perf_counter() - t)
print("rpn_head     host enqueue %.2f ms" % (1e3 * sum(ts) / len(ts)))
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t = time.perf_counter(); eng.proposals(c, geom, anchors, hw, 4, training=True); ts.append(time.perf_counter() - t)
print("proposals    host enqueue %.2f ms" % (1e3 * sum(ts) / len(ts)))
