import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aldi_amd import vit_ops as V
def timed(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for N, H, W, C in ((4, 200, 336, 192), (4, 100, 168, 384), (4, 50, 84, 768), (4, 25, 42, 1536)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); g = torch.randn_like(x)
    wt = torch.randn(7, 7, C, device="cuda").bfloat16(); b = torch.zeros(C, device="cuda"); dw = torch.zeros(7, 7, C, device="cuda")
    tf = timed(lambda: V.dwconv7(x, wt, b)); tw = timed(lambda: V.dwconv7_wgrad(x, g, dw))
    mb = x.numel() * 2 / 1e6
    print("N%d %dx%d C%d (%.0f MB): fwd %.1f us (%.2f TB/s of 2x tensor)  wgrad %.1f us" % (N, H, W, C, mb, tf, 2 * mb / tf, tw))
