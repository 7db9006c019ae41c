"""The data-parallel code path of the headline step on ONE GPU, for a kernel trace: a world-size-1 process group on the real backend (torch "nccl"
= RCCL), ALDI_DP_FORCE=1 (bucketed exchange on its launch stream behind the producers' events, recorded into the phase-B hipGraph),
ALDI_DP_TRACE=1 (an empty marker kernel where each bucket's collective is issued).

    rocprofv3 --kernel-trace -d gpurun_out/prof_dp/kt -o kt -- python tools/dp_trace.py [all_reduce|rs_ag] [steps]
    python tools/rocprof_summary.py dp gpurun_out/prof_dp/kt > gpurun_out/r04_dp_streams.txt
"""
import os
import random
import sys

os.environ.setdefault("ALDI_DP_FORCE", "1")
os.environ.setdefault("ALDI_DP_TRACE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29617")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench
from aldi_amd import synthetic as syn
from aldi_amd.trainer import ALDITrainer

exchange = sys.argv[1] if len(sys.argv) > 1 else "all_reduce"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = bench.make_cfg(1, 800, 1333, False)
cfg.SOLVER.GRAD_EXCHANGE = exchange
random.seed(1234)
torch.manual_seed(100)
tr = ALDITrainer(cfg)
data = syn.make_batch(2, 2, 800, 1333, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=100)
tr._trainer.data_loader = bench.FixedGpuLoader(data, torch.device("cuda", 0))
tr._trainer._data_loader_iter_obj = None
for it in range(steps):
    tr.iter = it
    tr.before_step()
    tr.run_step()
    tr.after_step()
torch.cuda.synchronize()
print("fused step stats:", dict(tr._trainer._fused_step.stats), "exchange:", exchange)
dist.destroy_process_group()
