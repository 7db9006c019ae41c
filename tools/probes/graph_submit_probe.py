"""How fast does a replayed hipGraph feed the GPU?  (a) one chain of N empty kernels: microseconds per node; (b) two branches of N
short kernels each (fork at the start, join at the end), captured alternately: do the branches run side by side from the start,
or does the second one begin when the first has been submitted?  (probe, not product code)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aldi_amd import _lib as L, ops

def noop():
    L.call("aldi_noop", ops.stream_ptr())

def timed(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    t_issue = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    return t_issue * 1e6, t_all * 1e6

x = torch.randn(1 << 22, device="cuda")
def work(n=1):           # a ~10 us kernel
    for _ in range(n):
        x.mul_(1.0001)

for N in (100, 400):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            noop()
    i, a = timed(g)
    print("chain of %d empty kernels: host issue %.0f us, wall %.0f us -> %.2f us per node" % (N, i, a, a / N))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work(N)
    i, a = timed(g)
    print("chain of %d elementwise kernels (4 M floats): host issue %.0f us, wall %.0f us -> %.2f us per node" % (N, i, a, a / N))

side = torch.cuda.Stream()
y = torch.randn(1 << 22, device="cuda")
for N in (100,):
    for mode in ("alternate", "a-then-b"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            if mode == "alternate":
                for _ in range(N):
                    x.mul_(1.0001)
                    with torch.cuda.stream(side):
                        y.mul_(1.0001)
            else:
                for _ in range(N):
                    x.mul_(1.0001)
                with torch.cuda.stream(side):
                    for _ in range(N):
                        y.mul_(1.0001)
            main.wait_stream(side)
        i, a = timed(g)
        print("two branches of %d elementwise kernels, captured %s: wall %.0f us (one branch alone: see above)" % (N, mode, a))

# (c) branches of low-occupancy compute kernels (a res5-sized 3x3 conv at N = 1: ~35 us, 17 x 4 workgroups on 256 CUs): side by side
# they should take as long as one branch.  One graph with a fork vs two graphs replayed on two streams.
xa = torch.randn(1, 25, 42, 512, device="cuda").bfloat16(); xb = xa.clone()
w = (torch.randn(512, 3, 3, 512, device="cuda") * 0.02).bfloat16()
N = 80
def chain(t):
    for _ in range(N):
        t = ops.conv2d(t, w, pad=1, relu=True)
    return t
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    ra = chain(xa)
_, a1 = timed(g1)
print("one branch of %d small convs: wall %.0f us" % (N, a1))
gf = torch.cuda.CUDAGraph()
with torch.cuda.graph(gf):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ra = chain(xa)
    with torch.cuda.stream(side):
        rb = chain(xb)
    main.wait_stream(side)
_, af = timed(gf)
print("ONE graph, two branches (fork / join): wall %.0f us" % af)
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga):
    ra = chain(xa)
with torch.cuda.graph(gb):
    rb = chain(xb)
def two():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ga.replay()
    with torch.cuda.stream(side):
        gb.replay()
    main.wait_stream(side)
class _G:
    replay = staticmethod(two)
_, a2 = timed(_G)
print("TWO graphs replayed on two streams: wall %.0f us" % a2)
def eager():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    t, u = xa, xb
    for _ in range(N):
        t = ops.conv2d(t, w, pad=1, relu=True)
        with torch.cuda.stream(side):
            u = ops.conv2d(u, w, pad=1, relu=True)
    main.wait_stream(side)
class _E:
    replay = staticmethod(eager)
_, a3 = timed(_E, reps=5)
print("eager, two streams, alternating issue: wall %.0f us" % a3)
