"""deformable attention backward at the encoder's full size (1333 x 800, 2 images): general scatter vs the gather form"""
import numpy as np
import torch
from aldi_amd import _lib as L
from aldi_amd.ops import _p, stream_ptr
g = torch.Generator().manual_seed(0)
shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
N, M, D, Lv, P = 2, 8, 32, 4, 4
S = sum(h * w for h, w in shapes)
sh = torch.tensor(shapes, dtype=torch.int32)
ls = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1]), dtype=torch.int32)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in shapes])
for px in (0.0, 2.0, 4.0, 8.0):
    if px == 0.0:      # the authors' initial offsets: whole pixels along eight directions
        th = torch.arange(M, dtype=torch.float32) * (2.0 * np.pi / M)
        gi = torch.stack([th.cos(), th.sin()], -1)
        gi = (gi / gi.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, Lv, P, 1) * torch.arange(1, P + 1).view(1, 1, P, 1)
        off = gi.view(1, 1, M, Lv, P, 2).expand(N, S, M, Lv, P, 2) / torch.tensor([[w, h] for h, w in shapes]).view(1, 1, 1, Lv, 1, 2)
    else:
        off = torch.randn(N, S, M, Lv, P, 2, generator=g) * px / torch.tensor([[w, h] for h, w in shapes]).view(1, 1, 1, Lv, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + off).contiguous().cuda()
    aw = torch.softmax(torch.randn(N, S, M, Lv * P, generator=g), -1).view(N, S, M, Lv, P).contiguous().cuda()
    value, gout = torch.randn(N, S, M, D, generator=g).cuda(), torch.randn(N, S, M * D, generator=g).cuda()
    shd, lsd = sh.cuda(), ls.cuda()
    gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(aw)
    host = np.ascontiguousarray(sh.numpy())
    res, keep = [], []
    for form in (0, "walk", 128, 512, 1024, 2048):
        L.reset_tuning()
        ws = None
        if form not in (0, "walk"):
            L.set_tuning("msda_bin_list", form)
            ws = torch.empty(int(L.lib.aldi_ms_deform_attn_backward_self_workspace(host.ctypes.data, N, S, M, Lv, P)), dtype=torch.uint8, device="cuda")
        def run():
            if form:
                L.call("aldi_ms_deform_attn_backward_self", _p(value), _p(shd), _p(lsd), host.ctypes.data, _p(loc), _p(aw), _p(gout), _p(gv), _p(gl), _p(ga),
                       _p(ws) if ws is not None else None, ws.numel() if ws is not None else 0, N, S, M, D, Lv, P, stream_ptr())
            else:
                L.call("aldi_ms_deform_attn_backward", _p(value), _p(shd), _p(lsd), _p(loc), _p(aw), _p(gout), _p(gv), _p(gl), _p(ga), N, S, M, D, S, Lv, P, stream_ptr())
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 200)
        keep.append(gv.clone())
    err = max((keep[0] - k).abs().max().item() for k in keep[1:]) / keep[0].abs().max().item()
    print("offsets %s: scatter %.0f us; walk form %.0f; binned form with lists >= 128 %.0f, 512 %.0f, 1024 %.0f, 2048 %.0f us (whole backward); max rel diff %.1e" % (("initial" if px == 0 else "sigma %.0f px" % px,) + tuple(res) + (err,)))
