"""cProfile of the host side of a few bench steps (where does the Python/ctypes time go?)."""
import cProfile, pstats, sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aldi_amd import synthetic as syn
from aldi_amd.trainer import ALDITrainer
cfg = bench.make_cfg(1, 800, 1333, False)
cfg.SOLVER.BASE_LR = 1e-4
cfg.SOLVER.FUSED_STEP = True
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
data = syn.make_batch(2, 2, 800, 1333, 8, seed=100)
tr._trainer.data_loader = bench.FixedGpuLoader(data, torch.device("cuda"))
def one():
    tr.before_step(); tr.run_step(); tr.after_step(); tr.iter += 1
tr.iter = 0
for _ in range(3): one()
torch.cuda.synchronize()
import time
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): one()
torch.cuda.synchronize()
pr.disable()
print("ms/step under cProfile:", (time.perf_counter() - t) / 3 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
t = time.perf_counter()
for _ in range(5): one()
torch.cuda.synchronize()
print("ms/step plain:", (time.perf_counter() - t) / 5 * 1e3)
