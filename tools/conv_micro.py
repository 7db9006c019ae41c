"""Run one conv shape repeatedly (for rocprofv3 --pmc).  usage: conv_micro.py N H W Cin Cout k [reps] [wgrad | f3]
f3 = the real forward conv3 launch of a bottleneck: FrozenBN scale / shift, residual, ReLU, mask bits written"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import ops
N, H, W, Cin, Cout, k = (int(a) for a in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
wg = len(sys.argv) > 8 and sys.argv[8] == "wgrad"
f3 = len(sys.argv) > 8 and sys.argv[8] == "f3"
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, H, W, Cin, device="cuda", generator=g).bfloat16()
w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / (Cin * k * k) ** 0.5).bfloat16()
y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
gy = torch.randn(N, H, W, Cout, device="cuda", generator=g).bfloat16()
dw = torch.zeros(Cout, k, k, Cin, device="cuda")
res = torch.randn(N, H, W, Cout, device="cuda", generator=g).bfloat16()
bits = torch.empty(N * H * W * Cout // 8, dtype=torch.uint8, device="cuda")
sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
def run():
    if f3:
        ops.conv2d(x, w, pad=k // 2, out=y, relu=True, scale=sc, shift=sh, res=res, res_mode=1, bits_out=bits)
    elif wg:
        ops.conv_wgrad(x, gy, dw, KH=k, KW=k, stride=1, pad=k // 2)
    else:
        ops.conv2d(x, w, pad=k // 2, out=y, relu=True)
run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%s %s: %.1f us  %.1f TFLOP/s" % ("wgrad" if wg else "igemm", sys.argv[1:7], us, 2.0 * N * H * W * Cin * Cout * k * k / us / 1e6))
