import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import ops
import aldi_amd._lib as L
dev = torch.device("cuda")
def bench(name, fn, n=2000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t)/n:8.1f} us/call   +drain {1e3*(t2-t1):.2f} ms")
x = torch.zeros(1024, device=dev)
bench("torch.empty small", lambda: torch.empty(1024, device=dev))
bench("torch.empty 64MB", lambda: torch.empty(16 << 20, device=dev))
bench("x.add_(1) (torch kernel)", lambda: x.add_(1))
src = torch.zeros(4096, device=dev); dst = torch.empty(4096, dtype=torch.bfloat16, device=dev)
bench("ops.cast_from_f32 (tiny HIP kernel)", lambda: ops.cast_from_f32(src, torch.bfloat16, out=dst))
xa = torch.zeros(2, 8, 8, 64, dtype=torch.bfloat16, device=dev); w = torch.zeros(64, 1, 1, 64, dtype=torch.bfloat16, device=dev); y = torch.empty(2, 8, 8, 64, dtype=torch.bfloat16, device=dev)
bench("ops.conv2d tiny (out given)", lambda: ops.conv2d(xa, w, out=y))
bench("ops.conv2d tiny (alloc out)", lambda: ops.conv2d(xa, w))
bench("stream_ptr()", lambda: ops.stream_ptr())
import ctypes
bench("L.lib.aldi_version()", lambda: L.lib.aldi_version())
c = torch.zeros(4, dtype=torch.int32, device=dev)
bench("c.cpu().tolist()", lambda: c.cpu().tolist(), 500)
bench("torch.tensor([..]).to(dev)", lambda: torch.tensor([1, 2, 3, 4], dtype=torch.int32).to(dev), 500)
bench("torch.randperm(268000)", lambda: torch.randperm(268000), 50)
bench("torch.randperm(1000)", lambda: torch.randperm(1000), 500)
