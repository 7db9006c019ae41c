"""ViTDet-B backbone (ViT-B/16 + SimpleFeaturePyramid) forward+backward on synthetic 800x1344 images: wall time per pass and,
with --profile, a per-kernel table (rocprofv3 --kernel-trace --stats wraps this script; see tools/profile_step.sh).

python tools/bench_vit.py [--n 2] [--steps 10] [--warmup 3] [--h 800 --w 1344] [--no-bwd]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--h", type=int, default=800)
    ap.add_argument("--w", type=int, default=1344)
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--no-sfp", action="store_true")
    a = ap.parse_args()
    from aldi_amd.vit import SimpleFeaturePyramid, ViT, VitConfig, VitParams
    dev = "cuda"
    cfg = VitConfig(sfp=not a.no_sfp, drop_path_rate=0.1)
    params = VitParams(cfg, dev)
    params.init_random(0)
    vit = ViT(params)
    sfp = SimpleFeaturePyramid(params) if cfg.sfp else None
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (a.n, 3, a.h, a.w), dtype=torch.uint8, generator=g).to(dev)
    sizes = [(a.h, a.w)] * a.n
    gh, gw = a.h // cfg.patch, a.w // cfg.patch

    def step():
        ds = vit.drop_path_scales(a.n, g) if not a.no_bwd else None
        ctx = vit.forward(img, sizes, save=not a.no_bwd, drop_scales=ds)
        x = ctx.out.view(a.n, gh, gw, cfg.embed)
        if sfp is not None:
            c = sfp.forward(x, save=not a.no_bwd)
        if a.no_bwd:
            return
        params.zero_grad()
        if sfp is not None:
            gP = [torch.ones_like(p_) for p_ in c.P[:4]]
            gx = sfp.backward(c, gP)
        else:
            gx = torch.ones_like(ctx.out)
        vit.backward(ctx, gx.view(-1, cfg.embed))
        params.adamw_step(1e-4)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    T = a.n * gh * gw
    E = cfg.embed
    lin = 2 * T * (3 * E * E + E * E + 2 * cfg.mlp_ratio * E * E) * cfg.depth              # qkv, proj, fc1, fc2
    nw = (gh + 13) // 14 * ((gw + 13) // 14) * a.n
    att_w = 4 * nw * 196 * 196 * 64 * cfg.heads * (cfg.depth - len(cfg.global_blocks))
    att_g = 4 * a.n * (gh * gw) ** 2 * 64 * cfg.heads * len(cfg.global_blocks)
    fwd = lin + att_w + att_g
    total = fwd if a.no_bwd else 3 * fwd
    print(json.dumps({"workload": f"vitdet_b backbone {'fwd' if a.no_bwd else 'fwd+bwd+adamw'} N={a.n} {a.h}x{a.w}", "ms_per_pass": round(ms, 3),
                      "images_per_s": round(a.n / ms * 1e3, 2), "vit_algorithmic_tflop": round(total / 1e12, 3),
                      "vit_tflops": round(total / ms / 1e9, 1), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
