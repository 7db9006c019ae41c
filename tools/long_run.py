import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, random
import bench
from aldi_amd import synthetic as syn
from aldi_amd.trainer import ALDITrainer
wl = sys.argv[1] if len(sys.argv) > 1 else "r50_fpn"          # or vitdet_b
cfg = bench.make_cfg(1, 800, 1333, False, wl); cfg.SOLVER.FUSED_STEP = True
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
per = 1 if wl == "vitdet_b" else 2
data = syn.make_batch(per, per, 800, 1333, 8, seed=100)
tr._trainer.data_loader = bench.FixedGpuLoader(data, torch.device("cuda"))
tr.iter = 0
def one():
    tr.before_step(); tr.run_step(); tr.after_step(); tr.iter += 1
for k in range(120):
    one()
    if k % 20 == 19:
        torch.cuda.synchronize()
        l = tr._trainer.last_loss_dict
        tot = sum(float(v) for v in l.values())
        print(k + 1, "alloc MB %.0f reserved MB %.0f max MB %.0f" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6, torch.cuda.max_memory_allocated() / 1e6), "loss %.4f" % tot, "err", int(tr.model.engine.err))
