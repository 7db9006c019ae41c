"""RoIAlign backward (gather form) on the ROIs of a real benchmark step and on the synthetic distribution of tools/roialign_bench.py;
ROIALIGN_DBG bits (temporary ablation arms) select what is skipped."""
import math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aldi_amd import ops, _lib as L
dev = "cuda"
Hs, Ws = [200, 100, 50, 25], [336, 168, 84, 42]
def real_rois():
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer
    cfg = get_cfg(); add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "SOLVER.BASE_LR", 1e-4])
    random.seed(1234); torch.manual_seed(100)
    tr = ALDITrainer(cfg)
    for it in range(3):
        tr.iter = it; tr.before_step(); tr.run_step(); tr.after_step()
    torch.cuda.synchronize()
    c = tr.model._last_fused
    r = c.rois[: c.R].float().clone()
    del tr
    return r
def synth(N, per, seed):
    g = torch.Generator().manual_seed(seed)
    R = N * per
    w = (torch.rand(R, generator=g) * 100 + 28); h = (torch.rand(R, generator=g) * 50 + 15)
    cx = torch.rand(R, generator=g) * 1333; cy = torch.rand(R, generator=g) * 800
    return torch.stack([torch.arange(R) // per, (cx - w / 2).clamp(0, 1332), (cy - h / 2).clamp(0, 799), (cx + w / 2).clamp(1, 1333), (cy + h / 2).clamp(1, 800)], 1).float().to(dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
sets = {"synthetic": synth(4, 512, 1)}
if os.environ.get("REAL", "1") == "1":
    sets["real step"] = real_rois()
    torch.save(sets["real step"].cpu(), os.path.join(ROOT, "gpurun_out", "rois_step.pt"))
N = 4
feats = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=torch.bfloat16, device=dev) for l in range(4)]
for name, rois in sets.items():
    R = rois.shape[0]
    gp = torch.randn((R, 7, 7, 256), device=dev).to(torch.bfloat16)
    grads = [torch.zeros((N, Hs[l], Ws[l], 256), dtype=torch.bfloat16, device=dev) for l in range(4)]
    rf = ops.make_roi_feats(feats, grads, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    ref = None
    for arm in [int(a) for a in os.environ.get("ARMS", "0,4,1,2").split(",")]:
        L.reset_tuning(); L.set_tuning(os.environ.get("KNOB", "wgrad_dbg"), arm)
        t = [timeit(lambda: ops.roialign_backward(rf, rois, R, 7, gp, N, rois_sorted=True, grad_dtype=torch.bfloat16)) for _ in range(3)]
        torch.cuda.synchronize()
        same = ""
        if ref is None: ref = [g.clone() for g in grads]
        else: same = " identical" if all(torch.equal(a, b) for a, b in zip(ref, grads)) else " differs"
        print(f"{name} R={R} arm {arm}: {min(t):.1f} us (median {sorted(t)[1]:.1f}){same}", flush=True)
L.reset_tuning()
