"""The weight-gradient launches of ONE benchmark step, each re-run in isolation right where the step issues it:
dispatch string (workgroups, pixels per workgroup), microseconds alone on the chip, TFLOP/s (tuning aid)."""
import os, random, sys
os.environ["ALDI_STEP_GRAPH"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aldi_amd import _lib as L, ops
from aldi_amd.config import add_aldi_config, get_cfg
from aldi_amd.trainer import ALDITrainer
cfg = get_cfg(); add_aldi_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SEED", 1, "SYNTHETIC.HEIGHT", 800, "SYNTHETIC.WIDTH", 1333, "SOLVER.BASE_LR", 1e-4])
cfg.SOLVER.FUSED_STEP = True
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
REC = {"on": False}
def timed(fn, name):
    def wrap(*a, **kw):
        fn(*a, **kw)
        if not REC["on"]:
            return
        which = L.last_dispatch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn(*a, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 3 * 1000
        if name == "group":
            probs = a[0]
            fl = sum(2.0 * x.shape[0] * g.shape[1] * g.shape[2] * x.shape[3] * g.shape[3] * kw_["KH"] * kw_["KW"] for x, g, dw, kw_ in probs)
            desc = "%d problems" % len(probs)
        else:
            x, g = a[0], a[1]
            fl = 2.0 * x.shape[0] * g.shape[1] * g.shape[2] * x.shape[3] * g.shape[3] * kw.get("KH", 1) * kw.get("KW", 1)
            desc = "x %s g %s k%d" % (tuple(x.shape), tuple(g.shape), kw.get("KH", 1))
        print("%-9s %8.1f us %7.1f TFLOP/s  %-34s %s" % (name, us, fl / us / 1e6, desc, which), flush=True)
    return wrap
ops.conv_wgrad_group = timed(ops.conv_wgrad_group, "group")
ops.conv_wgrad = timed(ops.conv_wgrad, "single")
for it in range(4):
    REC["on"] = it == 3
    tr.iter = it; tr.before_step(); tr.run_step(); tr.after_step()
torch.cuda.synchronize()
