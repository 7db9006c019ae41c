"""ROI level / footprint statistics of one benchmark step (what the RoIAlign backward's gather kernel sees per workgroup)."""
import math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aldi_amd.config import add_aldi_config, get_cfg
from aldi_amd.trainer import ALDITrainer
from aldi_amd import synthetic as syn
cfg = get_cfg(); add_aldi_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "SOLVER.BASE_LR", 1e-4])
cfg.SOLVER.FUSED_STEP = True
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
for it in range(3):
    tr.iter = it; tr.before_step(); tr.run_step(); tr.after_step()
torch.cuda.synchronize()
c = tr.model._last_fused
rois = c.rois[: c.R].float().cpu()
print("R", c.R, "rows", c.rows)
w = rois[:, 3] - rois[:, 1]; h = rois[:, 4] - rois[:, 2]
lv = torch.floor(4 + torch.log2(torch.sqrt(w * h) / 224 + 1e-8)).clamp(2, 5).int()
for l in range(2, 6):
    m = lv == l
    s = 4 * 2 ** (l - 2)
    if m.any():
        print(f"level {l}: {int(m.sum())} rois, mean w {float(w[m].mean()) / s:.1f} h {float(h[m].mean()) / s:.1f} px at the level; max h {float(h[m].max()) / s:.1f}")
# candidates per workgroup (row, 32-px segment)
Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
tot = 0
for l in range(2, 6):
    s = 4 * 2 ** (l - 2); H, W = Hs[l - 2], Ws[l - 2]
    segs = (W + 31) // 32
    cnt = torch.zeros(4, H, segs, dtype=torch.int32)
    for r in range(rois.shape[0]):
        if lv[r] != l: continue
        b = int(rois[r, 0]); x1, y1, x2, y2 = [float(v) / s - 0.5 for v in rois[r, 1:]]
        r0 = min(max(math.floor(y1) - 1, 0), H - 1); r1 = min(max(math.floor(y2) + 2, 0), H - 1)
        c0 = min(max(math.floor(x1) - 1, 0), W - 1); c1 = min(max(math.floor(x2) + 2, 0), W - 1)
        cnt[b, r0:r1 + 1, c0 // 32: c1 // 32 + 1] += 1
    tot += int(cnt.sum())
    print(f"level {l}: workgroups {cnt.numel()}, (wg, candidate) pairs {int(cnt.sum())}, per wg mean {float(cnt.float().mean()):.1f} max {int(cnt.max())}")
print("total pairs", tot)
