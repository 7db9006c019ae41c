"""Is the step host-bound?  Per step: wall time, time the host spends BLOCKED in the one device->host sync of the fused
forward (the GPU is behind the host there), and the time the host needs to issue everything else.  If the blocked time is
~0 the Python sequencer, not the GPU, sets the pace of the first half of the step."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aldi_amd import synthetic as syn
from aldi_amd.trainer import ALDITrainer
cfg = bench.make_cfg(1, 800, 1333, False)
cfg.SOLVER.FUSED_STEP = True
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
data = syn.make_batch(2, 2, 800, 1333, 8, seed=100)
tr._trainer.data_loader = bench.FixedGpuLoader(data, torch.device("cuda"))
blocked = [0.0]
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    t = time.perf_counter(); r = orig_cpu(self, *a, **k); blocked[0] += time.perf_counter() - t; return r
torch.Tensor.cpu = cpu
def one():
    tr.before_step(); tr.run_step(); tr.after_step(); tr.iter += 1
tr.iter = 0
for _ in range(5): one()
torch.cuda.synchronize()
K = 20
blocked[0] = 0.0
t0 = time.perf_counter()
for _ in range(K): one()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("free-running: %.2f ms/step wall | host issue %.2f ms/step of which blocked in the sync %.2f ms | final drain %.2f ms total"
      % (t_all / K * 1e3, t_issue / K * 1e3, blocked[0] / K * 1e3, (t_all - t_issue) * 1e3))
# host alone: sync before every step so the GPU is idle when issuing starts; the blocked time then = GPU time of the first half
blocked[0] = 0.0; tot = 0.0; drain = 0.0
for _ in range(K):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); one(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tot += t1 - t0; drain += t2 - t1
print("synchronised: host issue %.2f ms/step (blocked in the sync %.2f) + drain %.2f ms => pure host work %.2f ms/step"
      % (tot / K * 1e3, blocked[0] / K * 1e3, drain / K * 1e3, (tot - blocked[0]) / K * 1e3))
