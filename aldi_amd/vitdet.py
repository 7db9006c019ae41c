"""ViTDet-B Faster R-CNN on the HIP engine (BASELINE cfg 4; reference configs/Base-RCNN-VitDetB.yaml + aldi/backbone.py:36-43).

Only the architecture-specific parts differ from engine.RCNN: trunk (ViT + SimpleFeaturePyramid), the two-conv RPN head
(RPN.CONV_DIMS [-1, -1]), the 4 x (conv3x3 + LN + ReLU) + FC box head, and the backward pass through them.  Anchors, proposal
generation, matching / sampling, ROIAlign, every loss and the distillation targets are the engine's, unchanged: the ALDI step
(burn-in, EMA teacher, pseudo labels, hard / soft distillation) drives this model through the same calls as the R50-FPN one.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from types import SimpleNamespace

from . import ops
from . import vit_ops as V
from .engine import RCNN, Ctx
from .vit import SimpleFeaturePyramid, ViT, VitConfig, VitParams


class FlatParamRCNN(RCNN):
    """engine.RCNN over a `VitParams` flat container (detectron2-named parameters, packed predictor rows) instead of the R50
    `Weights` / `ParamLayout`: shared by the ViTDet and ConvNeXt-FPN detectors, which override the architecture-specific methods"""

    def _init_flat(self, params: VitParams, num_classes: int, seed: int):
        cfg = params.cfg
        assert cfg.num_classes == num_classes
        self.vp = params
        self.wts = params                       # what engine.RCNN calls `weights` (only touched by the overrides)
        self.K = num_classes
        self.device, self.dtype = params.device, torch.bfloat16
        self.Cp = (5 * num_classes + 1 + 15) // 16 * 16
        self.Ch = (5 * cfg.num_anchors + 15) // 16 * 16
        self._anchor_cache, self._ws = {}, {}
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.set_num_threads(1)
        self.has_img_da = self.has_ins_da = False
        self.drop_gen = torch.Generator().manual_seed(seed)     # stochastic-depth masks (host-drawn, see ViT.drop_path_scales)

    # ------------------------------------------------------------------ small helpers over the flat parameter container
    def _w(self, name):
        return self.vp.w(name + ".weight")

    def _b(self, name):
        return self.vp.m(name + ".bias")

    def _pack_w(self, name, cin):
        rows = self.Ch if name == "rpn_head_out" else self.Cp
        return self.vp.pack(self.vp.compute, name + ".weight", (rows, 1, 1, cin)), self.vp.pack(self.vp.master, name + ".bias", (rows,))

    def _pack_wt(self, name, cin):
        """data-gradient weights of a packed output layer (tiny: re-derived per use)"""
        rows = self.Ch if name == "rpn_head_out" else self.Cp
        return ops.dgrad_weights(self.vp.pack(self.vp.master, name + ".weight", (rows, 1, 1, cin)), None, torch.bfloat16)

    def _wg(self, name, x, g, k):
        ops.conv_wgrad(x, g, self.vp.g(name + ".weight"), KH=k, KW=k, stride=1, pad=k // 2)

    def _wg_pack(self, name, x, g, cin):
        rows = self.Ch if name == "rpn_head_out" else self.Cp
        ops.conv_wgrad(x, g, self.vp.pack(self.vp.grad, name + ".weight", (rows * cin,)), KH=1, KW=1)
        ops.bias_grad(g.view(-1, rows), self.vp.pack(self.vp.grad, name + ".bias", (rows,)))

    def _grads_final(self, names):
        return

    def _wgrad(self, name, x, g):
        raise RuntimeError("engine.RCNN._wgrad is tied to the R50-FPN layout; the flat-container engines do not use it")


def _refresh(eng, ds):
    if ds is None:
        eng._ds_dev = None
    else:
        # one (pinned ring, device buffer) pair PER SHAPE, never freed: a hipGraph recorded for another batch size keeps reading the buffer it was
        # recorded with, so a change of N must not reallocate it
        bufs = eng.__dict__.setdefault("_ds_bufs", {})
        ent = bufs.get(tuple(ds.shape))
        if ent is None:
            # two pinned slots, each guarded by the event of its last upload: the host may be a step ahead of the GPU, and the copy out of a slot
            # must have run before the next draw overwrites it
            cuda = torch.cuda.is_available()
            ent = bufs[tuple(ds.shape)] = SimpleNamespace(
                ring=[[torch.empty_like(ds).pin_memory() if cuda else torch.empty_like(ds), None] for _ in range(2)],
                dev=torch.empty(ds.shape, dtype=ds.dtype, device=eng.device), slot=0)
        eng._ds_dev = ent.dev
        slot = ent.ring[ent.slot]
        ent.slot ^= 1
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].copy_(ds)
        eng._ds_dev.copy_(slot[0], non_blocking=True)
        if eng._ds_dev.is_cuda:
            slot[1] = slot[1] or torch.cuda.Event()
            slot[1].record()
        eng._ds_host = slot[0]                      # (the draw the device buffer holds once the stream reaches this point: tests compare the two)
    eng._ds_fresh = N_of(ds)


def N_of(ds):
    return True if ds is None else int(ds.shape[-1])


def _staged_drop_scales(eng, draw, N):
    """stochastic-depth multipliers of one training pass in a PERSISTENT device buffer: the host draws them (same generator stream as before) into
    pinned memory and one stream-ordered copy refreshes the buffer -- the kernels read the multipliers from device memory, so the launches of the
    pass do not depend on what was drawn and a captured step can be replayed (FusedStep calls `refresh_drop_scales` before every pass and clears the
    mark behind it; a pass that finds no fresh draw FOR ITS N -- the eager paths -- draws itself and takes a PRIVATE copy: its saved context keeps
    views of the multipliers for the backward, and the sequential driver runs several forwards before their backwards)."""
    fresh = eng.__dict__.pop("_ds_fresh", None)
    dev = eng.__dict__.get("_ds_dev")
    if fresh is not None and (fresh is True or fresh == N) and (dev is None or dev.shape[-1] == N):
        return dev
    eng.refresh_drop_scales(N)
    eng.__dict__.pop("_ds_fresh", None)
    dev = eng.__dict__.get("_ds_dev")
    return None if dev is None else dev.clone()


class VitDetRCNN(FlatParamRCNN):
    def __init__(self, params: VitParams, num_classes: int, seed: int = 0):
        assert params.cfg.sfp
        self._init_flat(params, num_classes, seed)
        self.vit = ViT(params)
        self.sfp = SimpleFeaturePyramid(params)

    graph_safe = True            # (fused_step: the launches of a pass are the same every step)

    def refresh_drop_scales(self, N: int):
        _refresh(self, self.vit.drop_path_scales(N, self.drop_gen))

    # ------------------------------------------------------------------ forward
    def trunk(self, st_u8: torch.Tensor, sizes, save: bool) -> Ctx:
        cfg = self.vp.cfg
        N = st_u8.shape[0]
        ds = _staged_drop_scales(self, None, N) if save else None               # stochastic depth: training (student) passes only
        cv = self.vit.forward(st_u8, sizes, save=save, drop_scales=ds, hw_dev=self.__dict__.get("_hw_dev", {}).get(st_u8.data_ptr()))
        gh, gw = st_u8.shape[2] // cfg.patch, st_u8.shape[3] // cfg.patch
        cs = self.sfp.forward(cv.out.view(N, gh, gw, cfg.embed), save=save)
        c = Ctx()
        c.P = cs.P
        if save:
            c.vit_ctx, c.sfp_ctx = cv, cs
        return c

    def rpn_head(self, c: Ctx, save: bool):
        rp = "proposal_generator.rpn_head."
        C = self.vp.cfg.fpn_channels
        w_out, b_out = self._pack_w("rpn_head_out", C)
        heads, ts = [], []
        for f in c.P:
            t0 = ops.conv2d(f, self._w(rp + "conv.conv0"), pad=1, shift=self._b(rp + "conv.conv0"), relu=True)
            t1 = ops.conv2d(t0, self._w(rp + "conv.conv1"), pad=1, shift=self._b(rp + "conv.conv1"), relu=True)
            heads.append(ops.conv2d(t1, w_out, shift=b_out, want_f32=True))
            if save:
                ts.append((t0, t1))
        c.head = heads
        if save:
            c.rpn_t = ts

    def box_head(self, pooled: torch.Tensor, c: Optional[Ctx] = None):
        cfg = self.vp.cfg
        R, C = pooled.shape[0], cfg.fpn_channels
        bh = "roi_heads.box_head."
        x = pooled
        recs = []
        for i in range(1, cfg.box_convs + 1):
            y = ops.conv2d(x, self._w(f"{bh}conv{i}"), pad=1)
            a, mean, rstd = V.layernorm_forward(y.view(-1, C), self.vp.m(f"{bh}conv{i}.norm.weight"), self.vp.m(f"{bh}conv{i}.norm.bias"),
                                                eps=cfg.ln_eps, relu=True)
            a = a.view(y.shape)
            recs.append((x, y, mean, rstd, a))
            x = a
        fc1 = ops.conv2d(x.view(R, 1, 1, -1), self.vp.w(bh + "fc1.weight", (cfg.fc_dim, 1, 1, cfg.pool * cfg.pool * C)),
                         shift=self._b(bh + "fc1"), relu=True)
        w_out, b_out = self._pack_w("box_pred", cfg.fc_dim)
        pred = ops.conv2d(fc1, w_out, shift=b_out, want_f32=True).view(R, self.Cp)
        if c is not None:
            c.pooled, c.bh_recs, c.fc1, c.fc2 = pooled, recs, fc1, fc1       # `fc2` = the box-head output the hooks / distiller read
        return pred, fc1

    # ------------------------------------------------------------------ backward
    def _box_head_backward(self, c: Ctx) -> torch.Tensor:
        """c.gpred (fp32 [R, Cp]) -> gradient wrt the pooled ROI features [R, 7, 7, C]; parameter gradients accumulate"""
        cfg, vp, T = self.vp.cfg, self.vp, torch.bfloat16
        C, bh = cfg.fpn_channels, "roi_heads.box_head."
        gpred = ops.cast_from_f32(c.gpred[:c.R], T).view(c.R, 1, 1, self.Cp)
        self._wg_pack("box_pred", c.fc1, gpred, cfg.fc_dim)
        g_fc1 = ops.conv2d(gpred, self._pack_wt("box_pred", cfg.fc_dim), mask=c.fc1)
        x_last = c.bh_recs[-1][4].view(c.R, 1, 1, -1)
        ops.conv_wgrad(x_last, g_fc1, vp.g(bh + "fc1.weight"), KH=1, KW=1)
        ops.bias_grad(g_fc1.view(c.R, -1), vp.g(bh + "fc1.bias"))
        g = ops.conv2d(g_fc1, vp.wt(bh + "fc1.weight")).view(c.R, cfg.pool, cfg.pool, C)
        for i in range(cfg.box_convs, 0, -1):
            x, y, mean, rstd, a = c.bh_recs[i - 1]
            dy = V.layernorm_backward(g.reshape(-1, C), y.view(-1, C), vp.m(f"{bh}conv{i}.norm.weight"), mean, rstd,
                                      vp.g(f"{bh}conv{i}.norm.weight"), vp.g(f"{bh}conv{i}.norm.bias"), mask=a.view(-1, C)).view(y.shape)
            self._wg(f"{bh}conv{i}", x, dy, 3)
            g = ops.conv2d(dy, vp.wt(f"{bh}conv{i}.weight"), pad=1)
        return g

    def _rpn_head_backward(self, c: Ctx) -> List[torch.Tensor]:
        """c.ghead (fp32 per level) -> gradients wrt p2..p6 (bf16); the head is shared over the 5 levels"""
        vp, T = self.vp, torch.bfloat16
        C, rp = vp.cfg.fpn_channels, "proposal_generator.rpn_head."
        gP = []
        wt_out = self._pack_wt("rpn_head_out", C)
        for l in range(len(c.P)):
            t0, t1 = c.rpn_t[l]
            gh = ops.cast_from_f32(c.ghead[l], T)
            self._wg_pack("rpn_head_out", t1, gh, C)
            g1 = ops.conv2d(gh, wt_out, mask=t1)
            self._wg(rp + "conv.conv1", t0, g1, 3)
            ops.bias_grad(g1.view(-1, C), vp.g(rp + "conv.conv1.bias"))
            g0 = ops.conv2d(g1, vp.wt(rp + "conv.conv1.weight"), pad=1, mask=t0)
            self._wg(rp + "conv.conv0", c.P[l], g0, 3)
            ops.bias_grad(g0.view(-1, C), vp.g(rp + "conv.conv0.bias"))
            gP.append(ops.conv2d(g0, vp.wt(rp + "conv.conv0.weight"), pad=1))
        return gP

    def _backward_trunk(self, c: Ctx, align_list: List[dict]):
        """heads -> SimpleFeaturePyramid -> ViT given d(loss)/d(head outputs) in c.ghead / c.gpred (fp32)"""
        assert not align_list, "adversarial alignment is not wired for the ViTDet trunk"
        cfg, dev = self.vp.cfg, self.device
        gP_roi = [(torch.empty if c.R > 0 else torch.zeros)(f.shape, dtype=torch.float32, device=dev) for f in c.P[:4]]
        if c.R > 0:
            g_pooled = self._box_head_backward(c)
            ops.roialign_backward(self.roi_feats(c, gP_roi), c.rois, c.R, cfg.pool, g_pooled, c.N, rois_sorted=True)
        gP = self._rpn_head_backward(c)
        ops.subsample2_bwd(gP[4], gP[3])                               # p6 = p5[:, ::2, ::2]
        for l in range(4):
            ops.add_f32(gP[l], gP_roi[l], gP[l])
        cb = getattr(self, "grad_ready", None)
        if cb is not None:                                             # heads are final: start their exchange under the trunk backward
            cb(self.vp.ranges([n for n in self.vp.spec if n.startswith(("proposal_generator.", "roi_heads."))]))
        gx = self.sfp.backward(c.sfp_ctx, gP[:4])
        if cb is not None:
            cb(self.vp.ranges([n for n in self.vp.spec if n.startswith("backbone.simfp_")]))
        self.vit.grad_ready = cb
        try:
            self.vit.backward(c.vit_ctx, gx.view(-1, cfg.embed))
        finally:
            self.vit.grad_ready = None

