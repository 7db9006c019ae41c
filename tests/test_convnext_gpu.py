"""ConvNeXt trunk (HIP) vs the REFERENCE's own ConvNeXt class (aldi/backbone.py:189-352): golden g10 was produced by importing that
class in the build container (tests/golden/make_golden.py g10) -- state_dict, image, the four normalised stage outputs and every
parameter gradient.  bf16 activations: 4e-2 of range on the maps, 6e-2 (L2 3e-2) on gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def l2err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 13, 17, 64), (1, 5, 3, 8), (1, 9, 41, 200), (4, 50, 84, 768)],
                         ids=["small", "tiny", "ragged", "stage2"])
def test_dwconv7_fwd_bwd_vs_torch(dtype, shape):
    """tiny: H < 7, W < one pixel quad; ragged: W % 4 != 0, channel groups not a power of two; stage2: a ConvNeXt-L stage-2 map
    (threads walk several quads: the cross-quad operand prefetch and its tail)"""
    from aldi_amd import vit_ops as V
    torch.manual_seed(0)
    N, H, W, C = shape
    x = torch.randn(N, H, W, C, device=DEV).to(dtype)
    w = (torch.randn(C, 1, 7, 7, device=DEV) * 0.2).to(dtype)
    b = torch.randn(C, device=DEV) * 0.1
    g = torch.randn(N, H, W, C, device=DEV).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=3, groups=C)
    yr.backward(g.float().permute(0, 3, 1, 2))
    wt = w[:, 0].permute(1, 2, 0).contiguous()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(V.dwconv7(x, wt, b), yr.permute(0, 2, 3, 1)) < tol
    assert relerr(V.dwconv7(g, wt, None, flip=True), xr.grad.permute(0, 2, 3, 1)) < tol
    dw = torch.zeros(7, 7, C, device=DEV)
    V.dwconv7_wgrad(x, g, dw)
    assert relerr(dw, wr.grad[:, 0].permute(1, 2, 0)) < max(tol, 1e-4)


def test_scale_add_fwd_bwd_vs_torch():
    from aldi_amd import vit_ops as V
    torch.manual_seed(1)
    rows, C, rps = 60, 32, 30
    x, y, g = (torch.randn(rows, C, device=DEV) for _ in range(3))
    gamma = torch.randn(C, device=DEV)
    s = torch.tensor([0.0, 1.25], device=DEV)
    yr, gr = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    out_r = x + s.repeat_interleave(rps)[:, None] * gr * yr
    out_r.backward(g)
    assert torch.allclose(V.scale_add(x, y, gamma, s, rps), out_r, atol=1e-6)
    dg = torch.zeros(C, device=DEV)
    dy = V.scale_add_backward(g, y, gamma, s, dg, rps)
    assert torch.allclose(dy, yr.grad, atol=1e-6) and torch.allclose(dg, gr.grad, atol=1e-4)


def test_convnext_trunk_vs_reference_class_golden(golden_dir):
    from aldi_amd.convnext import ConvNeXt, ConvNeXtConfig
    from aldi_amd.vit import VitParams
    G = np.load(os.path.join(golden_dir, "g10_convnext.npz"))
    cfg = ConvNeXtConfig(depths=(1, 1, 2, 1), dims=(32, 64, 96, 128), drop_path_rate=0.0)
    params = VitParams(cfg, DEV)
    keys = [str(k) for k in G["keys"]]
    assert [cfg.prefix + k for k in keys] == list(params.spec.keys())          # same state_dict keys, same order as the reference module
    sd = {cfg.prefix + k: torch.from_numpy(G["sd." + k]) for k in keys}
    params.load_state_dict(sd)
    rt = params.state_dict()
    assert all(torch.equal(rt[k], sd[k]) for k in sd)
    net = ConvNeXt(params)
    img = torch.from_numpy(G["img"]).to(DEV)
    params.zero_grad()
    ctx = net.forward(img, [(64, 96), (64, 96)], save=True)
    for i in range(4):
        ref = torch.from_numpy(G[f"out{i}"]).to(DEV).permute(0, 2, 3, 1)
        assert relerr(ctx.outs[i], ref) < 4e-2, i
    net.backward(ctx, [torch.from_numpy(G[f"gout{i}"]).to(DEV).permute(0, 2, 3, 1).contiguous().bfloat16() for i in range(4)])
    grads = params.state_dict_like(params.grad)
    bad = {}
    for k in keys:
        ref = torch.from_numpy(G["grad." + k])
        mine = grads[cfg.prefix + k]
        e_inf, e_2 = relerr(mine, ref), l2err(mine, ref)
        if e_inf > 6e-2 or e_2 > 3e-2:
            bad[k] = (round(e_inf, 4), round(e_2, 4))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ the ConvNeXt-FPN detector
K = 8
CC = dict(depths=(1, 1, 2, 1), dims=(32, 64, 96, 128))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detector(seed=0, drop=0.0):
    from aldi_amd.convnext import ConvNeXtConfig, ConvNeXtRCNN
    from aldi_amd.vit import VitParams
    cfg = ConvNeXtConfig(depths=CC["depths"], dims=CC["dims"], drop_path_rate=drop, num_classes=K, fc_dim=256)
    params = VitParams(cfg, DEV)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (shape, _) in params.spec.items():
        if name.endswith("gamma"):
            t = 0.5 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and name.endswith(".weight"):
            t = 1 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("dwconv.weight"):
            t = 0.15 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        sd[name] = t.bfloat16().float()
    params.load_state_dict(sd)
    return cfg, params, sd, ConvNeXtRCNN(params, K)


def _batch(seed=0):
    from aldi_amd import synthetic as syn
    _, data, _, _ = syn.make_batch(2, 0, 128, 160, K, seed=seed, boxes_per_image=(3, 6))
    return data


def test_convnext_fpn_train_step_close_to_oracle():
    """one training step of the ConvNeXt-FPN detector (anchor sizes 64..1024) vs the fp32 CPU oracle on identical inputs and RNG draws:
    losses within 8 % (bf16 trunk), every parameter group receives gradient"""
    from oracle import d2_convnext as oc
    from oracle import d2_rcnn as d2
    cfg, params, sd, m = _detector(5)
    data = _batch(0)
    ocfg = d2.make_cfg(num_classes=K, pixel_mean=cfg.pixel_mean, pixel_std=cfg.pixel_std, anchor_sizes=cfg.anchor_sizes, fc_dim=256)
    torch.manual_seed(5)
    ol = d2.forward_train(ocfg, sd, data, roi_seed=9, arch=oc.arch(CC))
    torch.manual_seed(5)
    c = m.forward_train([d["image"] for d in data], [d["instances"] for d in data], roi_seed=9)
    params.zero_grad()
    m.backward(c, {k: 1.0 for k in ol})
    torch.cuda.synchronize()
    assert int(m.err) == 0
    hl = {k: float(v) for k, v in m.loss_dict(c).items()}
    for k in ol:
        assert abs(hl[k] - float(ol[k])) < 0.08 * max(1.0, abs(float(ol[k]))), (k, hl[k], float(ol[k]))
    g = params.state_dict_like(params.grad)
    dead = [k for k, v in g.items() if v.abs().max() == 0]
    assert not dead and torch.isfinite(params.grad).all(), dead[:8]


def test_convnext_fpn_adamw_overfits_one_batch():
    cfg, params, sd, m = _detector(7, drop=0.1)
    data = _batch(1)
    imgs, insts = [d["image"] for d in data], [d["instances"] for d in data]
    totals = []
    for it in range(10):
        torch.manual_seed(11)
        c = m.forward_train(imgs, insts, roi_seed=13)
        ld = m.loss_dict(c)
        totals.append(sum(float(v) for v in ld.values()))
        params.zero_grad()
        m.backward(c, {k: 1.0 for k in ld})
        params.adamw_step(2e-4)
    assert all(t == t for t in totals) and totals[-1] < 0.8 * totals[0], totals


@pytest.mark.parametrize("fused", [True, False])
def test_convnext_fpn_aldi_trainer_runs(fused, tmp_path):
    """configs/cityscapes/ALDI-Best-ConvNeXt-Cityscapes.yaml through the reference-shaped trainer (EMA teacher, pseudo labels, soft
    distillation, AdamW) on a small ConvNeXt: finite losses with the reference's key set, teacher trails the student, checkpoint keys"""
    import random
    from aldi_amd.config import add_aldi_config, get_cfg
    from aldi_amd.trainer import ALDITrainer, EngineAdamW
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-ConvNeXt-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "SOLVER.IMS_PER_GPU", 2, "SOLVER.WARMUP_ITERS", 0, "SEED", 1, "EMA.ALPHA", 0.9,
                         "SYNTHETIC.HEIGHT", 128, "SYNTHETIC.WIDTH", 160, "SOLVER.BASE_LR", 2e-4])
    cfg.MODEL.CONVNEXT.DEPTHS, cfg.MODEL.CONVNEXT.DIMS = [1, 1, 2, 1], [32, 64, 96, 128]
    cfg.SOLVER.FUSED_STEP = fused
    cfg.OUTPUT_DIR = str(tmp_path)
    random.seed(0)
    torch.manual_seed(3)
    tr = ALDITrainer(cfg)
    assert tr.model.convnext and isinstance(tr._trainer.optimizer, EngineAdamW)
    w0 = tr.model.weights.master.clone()
    for _ in range(3):
        tr.before_step()
        tr.run_step()
        tr.after_step()
        tr.iter += 1
    torch.cuda.synchronize()
    ld = tr._trainer.last_loss_dict
    assert {"loss_cls_source_strong", "loss_obj_bce_distill", "loss_cls_ce_distill", "loss_rpn_l1_distill", "loss_roih_l1_distill"} <= set(ld)
    assert all(float(v) == float(v) and abs(float(v)) < 1e4 for v in ld.values()), ld
    assert int(tr.model.engine.err) == 0 and int(tr.ema.model.engine.err) == 0
    s, t = tr.model.weights.master, tr.ema.model.weights.master
    assert (s - w0).abs().max() > 0 and (t - s).abs().max() > 0
    sd = tr.model.state_dict()
    assert sd["backbone.bottom_up.stages.2.1.dwconv.weight"].shape == (96, 1, 7, 7) and "backbone.fpn_lateral3.weight" in sd


def test_convnext_trunk_ragged_batch_vs_oracle():
    """images of different sizes in one batch: the padding of the staging buffer must enter the 4x4 stem as zeros of the NORMALISED
    image (detectron2 pads after normalisation), here with stochastic depth on (host-drawn multipliers fed to both sides)"""
    from aldi_amd.convnext import ConvNeXt, ConvNeXtConfig
    from aldi_amd.vit import VitParams
    from oracle import d2_convnext as oc
    cfg, params, sd, _ = _detector(9)
    net = ConvNeXt(params)
    torch.manual_seed(10)
    img = torch.randint(0, 256, (2, 3, 96, 128), dtype=torch.uint8, device=DEV)
    sizes = [(96, 128), (70, 90)]
    img[1, :, 70:, :] = 0
    img[1, :, :, 90:] = 0
    ds = torch.tensor([[1.0, 0.0], [1.25, 1.25], [0.0, 1.0 / 0.7], [1.0, 1.0], [2.0, 0.0]])        # one row per block (5 blocks)
    ctx = net.forward(img, sizes, save=False, drop_scales=ds)
    x = img.float() - torch.tensor(cfg.pixel_mean, device=DEV).view(1, 3, 1, 1)
    x[1, :, 70:, :] = 0
    x[1, :, :, 90:] = 0
    sdd = {k: v.to(DEV) for k, v in sd.items()}
    ref = oc.convnext_forward(CC, sdd, x, drop_scales=ds.to(DEV))
    for i in range(4):
        assert relerr(ctx.outs[i], ref[i].permute(0, 2, 3, 1)) < 4e-2, i
