"""Multi-scale deformable attention sampling at Deformable-DETR encoder size (800x1344 input: levels 100x168 .. 13x21, every
pixel a query, 8 heads x 4 points x 4 levels, head dim 32): time and achieved HBM rate of the forward / backward kernels."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aldi_amd.detr import MSDeformAttnFunction
    dev = "cuda"
    N, M, D, P = 2, 8, 32, 4
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    Lq, Lv = S, len(shapes)
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(N, S, M, D, device=dev, generator=g, requires_grad=True)
    # encoder queries look around their own position: reference points + small offsets
    ref = torch.rand(N, Lq, 1, 1, 1, 2, device=dev, generator=g)
    loc = (ref + 0.02 * torch.randn(N, Lq, M, Lv, P, 2, device=dev, generator=g)).clamp(0, 1).requires_grad_(True)
    w = torch.softmax(torch.randn(N, Lq, M, Lv * P, device=dev, generator=g), -1).view(N, Lq, M, Lv, P).requires_grad_(True)
    shp = torch.tensor(shapes, device=dev)
    lstart = torch.cat([shp.new_zeros(1), (shp[:, 0] * shp[:, 1]).cumsum(0)[:-1]])
    gout = torch.randn(N, Lq, M * D, device=dev, generator=g)

    def timed(fn, reps=20):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    out = MSDeformAttnFunction.apply(value, shp, lstart, loc, w, 64)
    t_f = timed(lambda: MSDeformAttnFunction.apply(value.detach(), shp, lstart, loc.detach(), w.detach(), 64))

    def bwd():
        o = MSDeformAttnFunction.apply(value, shp, lstart, loc, w, 64)
        o.backward(gout)
    t_fb = timed(bwd)
    samples = N * Lq * M * Lv * P
    # algorithmic bytes: every sample reads 4 corners x D fp32 (cache hits included: this is the gather volume), plus loc / weights / output
    gather = samples * 4 * D * 4
    io = value.numel() * 4 + loc.numel() * 4 + w.numel() * 4 + out.numel() * 4
    print(json.dumps({"workload": f"ms_deform_attn encoder N={N} S=Lq={S} heads={M} D={D} levels={Lv} points={P}", "forward_us": round(t_f, 1),
                      "forward_plus_backward_us": round(t_fb, 1), "samples": samples, "gather_GB": round(gather / 1e9, 3), "compulsory_io_GB": round(io / 1e9, 3),
                      "forward_gather_TBps": round(gather / t_f / 1e6, 2), "forward_compulsory_TBps": round(io / t_f / 1e6, 3)}))


if __name__ == "__main__":
    main()
