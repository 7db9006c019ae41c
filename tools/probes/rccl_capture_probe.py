"""Can this stack (torch + RCCL) record a collective into a hipGraph?  world size 1 on cuda:0.  usage: rccl_capture_probe.py <variant>
variants: same = all_reduce on the capturing stream; side = all_reduce on a forked stream behind an event, joined before capture ends; async = side + async_op"""
import os, sys, faulthandler
faulthandler.enable()
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PORT", "29533"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
variant = sys.argv[1]
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()           # communicator set up eagerly
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y = x * 2
    if variant == "same":
        dist.all_reduce(y)
    else:
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            if variant == "async":
                w = dist.all_reduce(y, async_op=True); w.wait()
            else:
                dist.all_reduce(y)
        torch.cuda.current_stream().wait_stream(side)
    z = y + 1
print(variant, "captured"); sys.stdout.flush()
g.replay(); torch.cuda.synchronize()
print(variant, "replayed, z[0] =", float(z[0]))
dist.destroy_process_group()
