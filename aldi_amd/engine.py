"""Host orchestration of the GeneralizedRCNN (R50-FPN) forward / backward / inference on the HIP
kernels.  This layer owns NO arithmetic: it allocates device memory through torch, sequences
C-ABI calls (aldi_amd.ops) on the current HIP stream, and draws the sampling permutations from
the host torch RNG in exactly the order Detectron2 does (SURVEY.md section 7 "RNG parity").

What it replaces in the reference: the detectron2 ``GeneralizedRCNN.forward`` / ``.inference``
Python + ATen/cuDNN/torchvision graph behind ``model(...)`` at aldi/trainer.py:87,
aldi/distill.py:157,162 and aldi/pseudolabeler.py:21, and autograd's backward at aldi/trainer.py:79.
"""
from __future__ import annotations

import contextlib
import dataclasses
import inspect
import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .arch import FC_DIM, FPN_C, NUM_ANCHORS, POOL, STAGE_BLOCKS, Packed, ParamLayout, pad_to

PIXEL_MEAN = (103.530, 116.280, 123.675)
PIXEL_STD = (1.0, 1.0, 1.0)
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
STRIDES = (4, 8, 16, 32, 64)
GMAX = 128            # max GT / pseudo-GT boxes per image held on device
RPN_BATCH, RPN_POS_FRAC = 256, 0.5
ROI_BATCH, ROI_POS_FRAC = 512, 0.25
RPN_PRE = (2000, 1000)
RPN_POST = (1000, 1000)
RPN_NMS = 0.7
ROI_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
SCORE_THRESH, NMS_TEST, DETS = 0.05, 0.5, 100


_FAST_RANDPERM = None


def randperm_prefix(n: int, k: int) -> torch.Tensor:
    """torch.randperm(n)[:k] on the global CPU generator, bit-exact (incl. the generator state afterwards), through
    aldi_torch_randperm_prefix (O(k + n/624) instead of O(n) divisions).  Verified once against torch itself; if the
    installed torch ever changed its CPU randperm algorithm the plain call is used instead."""
    global _FAST_RANDPERM
    from . import _lib as L

    def fast(n_, k_):
        st = torch.get_rng_state()
        out = torch.empty(min(k_, n_), dtype=torch.int64)
        L.call("aldi_torch_randperm_prefix", st.data_ptr(), n_, k_, out.data_ptr())
        torch.set_rng_state(st)
        return out
    if _FAST_RANDPERM is None:
        saved = torch.get_rng_state()
        ok = True
        for n_, k_ in ((1000, 40), (70001, 300), (3, 3)):
            torch.manual_seed(12345)
            a, a2 = torch.randperm(n_)[:k_], torch.randperm(17)
            torch.manual_seed(12345)
            b, b2 = fast(n_, k_), torch.randperm(17)
            ok = ok and torch.equal(a, b) and torch.equal(a2, b2)
        torch.set_rng_state(saved)
        _FAST_RANDPERM = ok
    if _FAST_RANDPERM:
        return fast(n, k)
    return torch.randperm(n)[:k]


class Weights:
    """One model's state on device: flat fp32 master (whole state_dict incl. FrozenBN buffers),
    compute-dtype copy of the weights, folded BN scale/shift, and (student) grads + momentum."""

    def __init__(self, layout: ParamLayout, device, dtype: torch.dtype, trainable: bool):
        self.layout, self.device, self.dtype, self.trainable = layout, device, dtype, trainable
        L = layout
        self.master = torch.zeros(L.n_total, dtype=torch.float32, device=device)
        self.compute = self.master if dtype == torch.float32 else torch.zeros(L.n_weights, dtype=dtype, device=device)
        self.bn_scale = torch.zeros(max(L.bn_channels, 1), dtype=torch.float32, device=device)
        self.bn_shift = torch.zeros(max(L.bn_channels, 1), dtype=torch.float32, device=device)
        self._grad = None      # allocated on first use (an EMA teacher never needs them)
        self._mom = None
        self.first_step = True
        self._wt: Dict[str, torch.Tensor] = {}
        self._wt_keys: List[str] = []          # dgrad weights requested so far (re-derived in one launch per refresh)
        self._wt_plan = None
        self.wt_epoch = 0
        self._neg1: Dict[int, torch.Tensor] = {}
        self._stem_pk, self._stem_name = None, None
        self.lazy_wt = False             # True: sgd_step leaves the dgrad weights stale until someone asks / refreshes
        self._wt_dirty = False
        self._fold_names: List[str] = []    # layers whose scale-folded bf16 weights the fused bottleneck kernel reads
        self._fold_plan = None

    @property
    def grad(self) -> torch.Tensor:
        if self._grad is None:
            self._grad = torch.zeros(self.layout.n_train, dtype=torch.float32, device=self.device)
        return self._grad

    @property
    def mom(self) -> torch.Tensor:
        if self._mom is None:
            self._mom = torch.zeros(self.layout.n_train, dtype=torch.float32, device=self.device)
        return self._mom

    # ---- views -----------------------------------------------------------------------------
    def w(self, name: str) -> torch.Tensor:
        p = self.layout.t[name]
        n = p.rows * p.kk * p.kk * p.cin
        return self.compute[p.w_off: p.w_off + n].view(p.wshape)

    def w_master(self, name: str) -> torch.Tensor:
        p = self.layout.t[name]
        n = p.rows * p.kk * p.kk * p.cin
        return self.master[p.w_off: p.w_off + n].view(p.wshape)

    def b(self, name: str) -> Optional[torch.Tensor]:
        p = self.layout.t[name]
        return self.master[p.b_off: p.b_off + p.rows] if p.bias else None

    def scale(self, name: str) -> Optional[torch.Tensor]:
        p = self.layout.t[name]
        return self.bn_scale[p.bn_off: p.bn_off + p.rows] if p.bn else None

    def shift(self, name: str) -> Optional[torch.Tensor]:
        p = self.layout.t[name]
        return self.bn_shift[p.bn_off: p.bn_off + p.rows] if p.bn else self.b(name)

    def gw(self, name: str) -> torch.Tensor:
        p = self.layout.t[name]
        n = p.rows * p.kk * p.kk * p.cin
        return self.grad[p.w_off: p.w_off + n]

    def gb(self, name: str) -> torch.Tensor:
        p = self.layout.t[name]
        return self.grad[p.b_off: p.b_off + p.rows]

    def wt(self, name: str, negate: bool = False) -> torch.Tensor:
        """dgrad weights (rotated/transposed, BN scale folded; negated for the gradient-reversal layer)."""
        key = name + ("-" if negate else "")
        if self._wt_dirty:
            self._refresh_wt()
        if key not in self._wt:
            # first request of this layer: single-layer kernel now, and from the next refresh() on it is part of the one
            # batched launch that re-derives every requested layer right after the weights change
            self._wt[key] = ops.dgrad_weights(*self._wt_source(key), self.dtype)
            if key not in self._wt_keys:
                self._wt_keys.append(key)
                if self._wt_plan is not None:
                    self.wt_epoch += 1      # recorded graphs hold the old plan's buffers (fused_step drops them)
                self._wt_plan = None
        return self._wt[key]

    def wt_flat(self, name: str) -> torch.Tensor:
        """[KH*KW*Cin][1][1][Cout] = the layer's weight matrix transposed, rows in (tap, ci) order: one 1x1 GEMM with it gives the
        data-gradient contributions of a pixel to its KHxKW neighbourhood (sparse RPN backward)"""
        return self.wt(name + "@flat")

    def folded(self, names: Sequence[str]) -> List[torch.Tensor]:
        """bf16 weights of `names` with their FrozenBN scale folded in (the fused bottleneck's operands), re-derived by refresh().
        The set of names is fixed by the first call (one launch re-derives them all)."""
        if self._fold_plan is None:
            self._fold_names = list(names) if not self._fold_names else self._fold_names
            self._fold_plan = ops.FoldWeightsPlan([(self.w_master(n), self.scale(n)) for n in self._fold_names])
            self._fold_plan.run()
        idx = [self._fold_names.index(n) for n in names]
        return [self._fold_plan.out[i] for i in idx]

    def stem_packed(self, name: str) -> torch.Tensor:
        """the stem kernel in the fused stem+pool kernel's bf16 [64][200] layout; re-derived by refresh()"""
        if self._stem_pk is None:
            self._stem_pk = ops.stem_pack_weights(self.w_master(name))
            self._stem_name = name
        return self._stem_pk

    def _wt_source(self, key: str):
        """(fp32 master view, per-Cout scale) a dgrad-weight key is derived from"""
        neg = key.endswith("-")
        key = key[:-1] if neg else key
        flat = key.endswith("@flat")
        name = key[:-len("@flat")] if flat else key
        w = self.w_master(name)
        if flat:
            w = w.view(w.shape[0], 1, 1, -1)
        return w, self._wt_scale(name, neg)

    def _wt_scale(self, name: str, negate: bool):
        if not negate:
            return self.scale(name)
        rows = self.layout.t[name].rows
        if rows not in self._neg1:
            self._neg1[rows] = torch.full((rows,), -1.0, dtype=torch.float32, device=self.device)
        return self._neg1[rows]

    def _refresh_wt(self):
        self._wt_dirty = False
        self._wt.clear()
        if not self._wt_keys:
            return
        if self._wt_plan is None:
            self._wt_plan = ops.DgradWeightsPlan([self._wt_source(key) for key in self._wt_keys], self.dtype)
        self._wt_plan.run()
        for key, o in zip(self._wt_keys, self._wt_plan.out):
            self._wt[key] = o

    # ---- state -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        missing = [k for k in self.layout.state_dict_keys() if k not in sd]
        if missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.master.copy_(self.layout.pack(sd).to(self.device))
        self.refresh()

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return self.layout.unpack(self.master)

    def refresh(self, cast: bool = True):
        """re-derive everything computed from the master state (after load / optimizer step / EMA)."""
        L = self.layout
        if L.bn_channels:
            b = L.bn_base
            c = L.bn_channels
            ops.bn_fold(self.master[b:b + c], self.master[b + c:b + 2 * c], self.master[b + 2 * c:b + 3 * c],
                        self.master[b + 3 * c:b + 4 * c], self.bn_scale, self.bn_shift, c)
        if self.dtype != torch.float32 and cast:
            ops.cast_from_f32(self.master[:L.n_weights], self.dtype, out=self.compute)
        if self._stem_pk is not None:
            ops.stem_pack_weights(self.w_master(self._stem_name), out=self._stem_pk)
        if self._fold_plan is not None:
            self._fold_plan.run()
        self._refresh_wt()

    def zero_grad(self):
        self.grad.zero_()
        self._gscale = 1.0

    def scale_grad(self, f: float):
        """deferred scalar on the gradient (applied inside the fused optimizer kernel), e.g. 1/world after all-reduce(SUM)"""
        self._gscale = getattr(self, "_gscale", 1.0) * f

    def sgd_step(self, lr: float, momentum: float = 0.9, weight_decay: float = 1e-4, grad_scale: float = 1.0):
        n = self.layout.n_train
        ops.sgd_step(self.master, self.grad, self.mom, self.compute if self.dtype != torch.float32 else None, n, lr, momentum,
                     weight_decay, grad_scale * getattr(self, "_gscale", 1.0), self.first_step, self.dtype)
        self._after_sgd()

    def _after_sgd(self):
        self.first_step = False
        if self.lazy_wt:
            self._wt_dirty = True        # re-derived where the next backward needs them (the fused step does it beside its forward)
        else:
            self._refresh_wt()

    def sgd_range_dev(self, lo: int, hi: int, hyper: torch.Tensor):
        """the optimizer step of elements [lo, hi) with device-resident scalars (ops.sgd_step_dev): issued per layer group from
        inside the backward by the fused step"""
        hi = min(hi, self.layout.n_train)
        if hi > lo:
            ops.sgd_step_dev(self.master, self.grad, self.mom, self.compute if self.dtype != torch.float32 else None, lo, hi, hyper, self.dtype)

    def ema_from(self, student: "Weights", alpha: float, copy_only: bool):
        """reference aldi/ema.py:29-57 over the whole state (params AND buffers)."""
        n = self.layout.n_total
        if self.dtype == torch.bfloat16:                     # the bf16 compute copy of the weights in the same pass (no cast pass)
            ops.ema_update(self.master, student.master, self.compute, n, alpha, copy_only, self.dtype, n_compute=self.layout.n_weights)
            self.refresh(cast=False)
        else:
            ops.ema_update(self.master, student.master, None, n, alpha, copy_only, torch.float32)
            self.refresh()


def make_anchors(shapes: Sequence[Tuple[int, int]], device, sizes: Sequence[float] = ANCHOR_SIZES) -> torch.Tensor:
    """DefaultAnchorGenerator (offset 0): (sumA, 4), level-major, then (h, w, a). Host float32 math as Detectron2."""
    out = []
    for (h, w), size, stride in zip(shapes, sizes, STRIDES):
        cell = []
        for r in ANCHOR_RATIOS:
            area = size ** 2.0
            aw = math.sqrt(area / r)
            ah = r * aw
            cell.append([-aw / 2.0, -ah / 2.0, aw / 2.0, ah / 2.0])
        cell = torch.tensor(cell, dtype=torch.float32)
        sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return torch.cat(out).contiguous().to(device)


@dataclasses.dataclass(frozen=True)
class D2Params:
    """The Detectron2 config keys the engine's arithmetic depends on (defaults = detectron2's own; the module
    constants above).  `from_cfg` reads them from the config node the reference hands to `build_model` (configs/detectron2/
    Base-RCNN-FPN.yaml + overrides) and REJECTS values the kernels do not implement instead of silently ignoring them."""
    pixel_mean: Tuple[float, ...] = PIXEL_MEAN
    pixel_std: Tuple[float, ...] = PIXEL_STD
    anchor_sizes: Tuple[float, ...] = ANCHOR_SIZES
    rpn_batch: int = RPN_BATCH
    rpn_pos_frac: float = RPN_POS_FRAC
    rpn_iou: Tuple[float, float] = (0.3, 0.7)
    rpn_pre: Tuple[int, int] = RPN_PRE          # (train, test)
    rpn_post: Tuple[int, int] = RPN_POST
    rpn_nms: float = RPN_NMS
    roi_batch: int = ROI_BATCH
    roi_pos_frac: float = ROI_POS_FRAC
    roi_iou: float = 0.5
    roi_weights: Tuple[float, float, float, float] = ROI_WEIGHTS
    score_thresh: float = SCORE_THRESH
    nms_test: float = NMS_TEST
    dets: int = DETS

    @classmethod
    def from_cfg(cls, cfg) -> "D2Params":
        M = cfg.MODEL

        def get(node, key, default):
            return node.get(key, default) if hasattr(node, "get") else getattr(node, key, default)

        def need(cond, what):
            if not cond:
                raise ValueError(f"aldi_amd R50-FPN engine: unsupported config value: {what}")
        rpn, roi, box, ag = M.RPN, M.ROI_HEADS, get(M, "ROI_BOX_HEAD", {}), M.ANCHOR_GENERATOR
        sizes = [list(s_) for s_ in ag.SIZES]
        need(len(sizes) == 5 and all(len(s_) == 1 for s_ in sizes), f"ANCHOR_GENERATOR.SIZES {sizes} (one size per level p2..p6)")
        ratios = [list(r_) for r_ in ag.ASPECT_RATIOS]
        need(all(tuple(float(v) for v in r_) == ANCHOR_RATIOS for r_ in ratios), f"ANCHOR_GENERATOR.ASPECT_RATIOS {ratios} (kernels: {ANCHOR_RATIOS})")
        need(len(M.PIXEL_MEAN) == 3 and len(M.PIXEL_STD) == 3, "PIXEL_MEAN / PIXEL_STD must have 3 entries")
        need(list(get(rpn, "IN_FEATURES", ["p2", "p3", "p4", "p5", "p6"])) == ["p2", "p3", "p4", "p5", "p6"], "RPN.IN_FEATURES")
        need(list(get(roi, "IN_FEATURES", ["p2", "p3", "p4", "p5"])) == ["p2", "p3", "p4", "p5"], "ROI_HEADS.IN_FEATURES")
        need(tuple(get(rpn, "BBOX_REG_WEIGHTS", (1.0, 1.0, 1.0, 1.0))) == (1.0, 1.0, 1.0, 1.0), "RPN.BBOX_REG_WEIGHTS != (1, 1, 1, 1)")
        need(float(get(rpn, "SMOOTH_L1_BETA", 0.0)) == 0.0 and float(get(box, "SMOOTH_L1_BETA", 0.0)) == 0.0, "SMOOTH_L1_BETA != 0 (the losses are pure L1)")
        need(get(rpn, "BBOX_REG_LOSS_TYPE", "smooth_l1") == "smooth_l1" and get(box, "BBOX_REG_LOSS_TYPE", "smooth_l1") == "smooth_l1", "BBOX_REG_LOSS_TYPE")
        need(float(get(rpn, "LOSS_WEIGHT", 1.0)) == 1.0 and float(get(rpn, "BBOX_REG_LOSS_WEIGHT", 1.0)) == 1.0, "RPN loss weights != 1")
        need(int(get(box, "POOLER_RESOLUTION", POOL)) == POOL and int(get(box, "POOLER_SAMPLING_RATIO", 0)) == 0 and
             get(box, "POOLER_TYPE", "ROIAlignV2") == "ROIAlignV2", "ROI_BOX_HEAD pooler (kernels: ROIAlignV2, 7x7, adaptive sampling)")
        need(not get(box, "CLS_AGNOSTIC_BBOX_REG", False), "CLS_AGNOSTIC_BBOX_REG")
        need(get(roi, "PROPOSAL_APPEND_GT", True), "ROI_HEADS.PROPOSAL_APPEND_GT False")
        need(len(rpn.IOU_THRESHOLDS) == 2 and len(roi.IOU_THRESHOLDS) == 1, "IOU_THRESHOLDS (RPN: two, ROI heads: one)")
        pre = (int(rpn.PRE_NMS_TOPK_TRAIN), int(rpn.PRE_NMS_TOPK_TEST))
        post = (int(rpn.POST_NMS_TOPK_TRAIN), int(rpn.POST_NMS_TOPK_TEST))
        need(max(pre) <= 2048 and min(pre + post) >= 1, f"RPN.PRE_NMS_TOPK {pre} (the NMS workspace holds 2048 boxes per level and image)")
        dets = int(get(get(cfg, "TEST", {}), "DETECTIONS_PER_IMAGE", DETS))
        need(1 <= dets <= GMAX, f"TEST.DETECTIONS_PER_IMAGE {dets} (<= {GMAX}: pseudo-label rows)")
        need(int(rpn.BATCH_SIZE_PER_IMAGE) >= 1 and int(roi.BATCH_SIZE_PER_IMAGE) >= 1, "BATCH_SIZE_PER_IMAGE")
        return cls(pixel_mean=tuple(float(v) for v in M.PIXEL_MEAN), pixel_std=tuple(float(v) for v in M.PIXEL_STD),
                   anchor_sizes=tuple(float(s_[0]) for s_ in sizes), rpn_batch=int(rpn.BATCH_SIZE_PER_IMAGE), rpn_pos_frac=float(rpn.POSITIVE_FRACTION),
                   rpn_iou=(float(rpn.IOU_THRESHOLDS[0]), float(rpn.IOU_THRESHOLDS[1])), rpn_pre=pre, rpn_post=post, rpn_nms=float(rpn.NMS_THRESH),
                   roi_batch=int(roi.BATCH_SIZE_PER_IMAGE), roi_pos_frac=float(roi.POSITIVE_FRACTION), roi_iou=float(roi.IOU_THRESHOLDS[0]),
                   roi_weights=tuple(float(v) for v in get(box, "BBOX_REG_WEIGHTS", ROI_WEIGHTS)), score_thresh=float(roi.SCORE_THRESH_TEST),
                   nms_test=float(roi.NMS_THRESH_TEST), dets=dets)


def raise_on_error(code: int, who: str = "engine"):
    """decode an engine's device error word (RCNN.err; the kernels OR bits into it, nothing on the device stops)"""
    if code & 1:       # detectron2's find_top_rpn_proposals raises the same way
        raise FloatingPointError(f"{who}: predicted boxes or scores contain Inf/NaN. Training has diverged.")
    if code & 2:
        raise RuntimeError(f"{who}: sparse RPN-head backward: more active pixels than (2 * BATCH_SIZE_PER_IMAGE + 4 * positives) per image; "
                           "the excess was dropped, gradients of this step are wrong")
    if code & 8:
        raise RuntimeError(f"{who}: RPN top-k: the group barrier of `topk_fused_kernel` timed out (its workgroups were not co-resident); the candidates "
                           "of this step are wrong.  Set ALDI_RPN_TOPK_FUSED=0 for the five-launch path")
    if code:
        raise RuntimeError(f"{who}: device error word {code}")


class Ctx(dict):
    """Saved tensors / intermediates of one forward (what the reference reads through forward hooks)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class RCNN:
    p = D2Params()          # (subclasses that do not go through this constructor run on detectron2's defaults)

    def __init__(self, weights: Weights, num_classes: int, params: Optional[D2Params] = None):
        self.wts = weights
        if params is not None:
            self.p = params
        self.K = num_classes
        self.device, self.dtype = weights.device, weights.dtype
        self.Cp = weights.layout.t["box_pred"].rows
        self.Ch = weights.layout.t["rpn_head_out"].rows
        self._anchor_cache: Dict[tuple, tuple] = {}
        self._ws: Dict[str, torch.Tensor] = {}
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the host only runs tiny torch-CPU ops (RNG draws, index packing); on a many-core host the default
        # intra-op thread pool makes torch.randperm(268k) ~20x slower than one thread
        torch.set_num_threads(1)
        # discriminator layers in nn.Sequential order (hidden convs / linears, the last entry is the 1-logit Linear) and the FPN
        # level the image-level one reads (aldi/align.py:22-52)
        idx = lambda n: int(n.rsplit(".", 1)[1])
        names = list(getattr(weights.layout, "t", {}))
        self.img_da_layers = sorted((n for n in names if n.startswith("img_align.model.")), key=idx)
        self.ins_da_layers = sorted((n for n in names if n.startswith("ins_align.model.")), key=idx)
        self.has_img_da, self.has_ins_da = bool(self.img_da_layers), bool(self.ins_da_layers)
        self.fused_stem = os.environ.get("ALDI_FUSED_STEM", "1") == "1"                  # bf16: stem conv + max-pool in one kernel
        self.wgrad_tail_split = int(os.environ.get("ALDI_WGRAD_TAIL_SPLIT", "2"))       # res3's weight-gradient group launched every this many blocks (0: once)
        self.group_wgrad = os.environ.get("ALDI_WGRAD_GROUP", "1") == "1"                # bf16: a layer group's weight gradients in one launch
        self.mask_bits = os.environ.get("ALDI_MASK_BITS", "1") == "1"                    # ReLU masks of the block outputs as bits for the backward
        self.mask_bits_inner = os.environ.get("ALDI_MASK_BITS_INNER", "1") == "1"        # ... and of the two inner maps of every bottleneck
        self.level_groups = os.environ.get("ALDI_LEVEL_GROUPS", "1") == "1"              # one launch for a layer applied to several pyramid levels
        self.fused_res2 = os.environ.get("ALDI_FUSED_RES2", "1") == "1"                  # bf16: a res2 bottleneck (no saved activations) in one kernel
        self._wg_queue: list = []
        self.sparse_rpn_backward = os.environ.get("ALDI_RPN_SPARSE_BWD", "1") == "1"      # tests flip the attribute to compare with the dense form
        spec = getattr(weights.layout, "img_da", None)
        self.img_da_level = ("p2", "p3", "p4", "p5", "p6").index(spec["layer"]) if spec else 0

    # ------------------------------------------------------------------ inputs
    def stage_images(self, images: Sequence[torch.Tensor]):
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
        Hs = pad_to(max(s[0] for s in sizes), 32)
        Ws = pad_to(max(s[1] for s in sizes), 32)
        N = len(images)
        if all(im.is_cuda for im in images):
            st = torch.zeros((N, 3, Hs, Ws), dtype=torch.uint8, device=self.device)
            for i, im in enumerate(images):
                st[i, :, : sizes[i][0], : sizes[i][1]] = im
        else:
            host = torch.zeros((N, 3, Hs, Ws), dtype=torch.uint8).pin_memory()
            for i, im in enumerate(images):
                host[i, :, : sizes[i][0], : sizes[i][1]] = im
            st = host.to(self.device, non_blocking=True)
        hw = torch.tensor(sizes, dtype=torch.int32).to(self.device)
        return st, sizes, hw

    def stage_gt(self, instances: Sequence[dict]):
        N = len(instances)
        gb = torch.zeros((N, GMAX, 4), dtype=torch.float32)
        gc = torch.zeros((N, GMAX), dtype=torch.int32)
        cnt = torch.zeros((N,), dtype=torch.int32)
        on_dev = []
        for i, inst in enumerate(instances):
            b = inst["gt_boxes"]
            b = b.tensor if hasattr(b, "tensor") else b
            g = int(b.shape[0])
            if g > GMAX:
                raise ValueError(f"more than {GMAX} GT boxes in one image")
            if g:
                if b.is_cuda:                        # already resident: no host round trip
                    on_dev.append((i, g, b, inst["gt_classes"]))
                else:
                    gb[i, :g] = b.reshape(-1, 4).to(torch.float32)
                    gc[i, :g] = inst["gt_classes"].to(torch.int32)
            cnt[i] = g
        gbd, gcd, cntd = ops.upload_packed([gb, gc, cnt], self.device)
        for i, g, b, cl in on_dev:
            gbd[i, :g] = b.reshape(-1, 4).to(torch.float32)
            gcd[i, :g] = cl.to(torch.int32)
        return {"boxes": gbd, "classes": gcd, "count": cntd, "host_count": cnt.tolist()}

    def geometry(self, Hs: int, Ws: int):
        key = (Hs, Ws)
        if key not in self._anchor_cache:
            shapes = []
            h, w = Hs // 4, Ws // 4
            for l in range(4):
                shapes.append((h, w))
                if l < 3:
                    h, w = h // 2, w // 2
            shapes.append(((shapes[3][0] - 1) // 2 + 1, (shapes[3][1] - 1) // 2 + 1))
            geom = ops.make_geom(shapes, NUM_ANCHORS, self.Ch)
            anchors = make_anchors(shapes, self.device, getattr(self, "anchor_sizes", self.p.anchor_sizes))
            self._anchor_cache[key] = (shapes, geom, anchors)
            while len(self._anchor_cache) > 16:               # multi-scale training: keep the most recent padded sizes only (4 MB of anchors each)
                self._anchor_cache.pop(next(iter(self._anchor_cache)))
        else:
            self._anchor_cache[key] = self._anchor_cache.pop(key)
        return self._anchor_cache[key]

    def workspace(self, name: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[name] = t
        return t

    # ------------------------------------------------------------------ trunk
    def _conv_call(self, x, name, *, relu=False, res=None, res_mode=0, want_f32=False, out=None, bits_out=None):
        """(x, weight, conv2d kwargs) of layer `name` applied to x"""
        W = self.wts
        p = W.layout.t[name]
        return x, W.w(name), dict(stride=p.stride, pad=p.pad, scale=W.scale(name), shift=W.shift(name), res=res, res_mode=res_mode, relu=relu,
                                  want_f32=want_f32, out=out, bits_out=bits_out)

    def conv(self, x, name, **kw):
        x, w, kw = self._conv_call(x, name, **kw)
        return ops.conv2d(x, w, **kw)

    # The trunk and the RPN head are written as generators that YIELD their convolutions ((x, layer name, flags) -> output): run
    # alone (`_drive`) each request is one launch; two models of the same architecture run in lockstep (`drive_pair`: the
    # student's N = 4 batch and the teacher's N = 2 batch through their own weights) share ONE launch per layer.
    # A request may also be a LIST of such triples: independent convolutions of one layer shape (the output conv on the four
    # pyramid levels, the RPN conv on five) that share one launch (aldi_conv_igemm_group) -- with the other model's list too.
    def _serve(self, req):
        if isinstance(req, list):
            return ops.conv2d_group([self._conv_call(r[0], r[1], **r[2]) for r in req])
        return self.conv(req[0], req[1], **req[2])

    def _drive(self, gen):
        try:
            req = next(gen)
            while True:
                req = gen.send(self._serve(req))
        except StopIteration as e:
            return e.value

    @staticmethod
    def drive_pair(eng_a, gen_a, eng_b, gen_b, streams=None):
        """advance two engines' generators together, one grouped launch per pair of requests -> (result_a, result_b).
        streams = (stream_a, stream_b): no grouping -- each model's launches go to its own stream, ISSUED alternately (a hipGraph
        hands its nodes to the queues in capture order: a branch captured after the other one starts when the host has submitted
        everything before it, 2 ms into the phase for the teacher's pass behind the student's trunk)"""
        res = [None, None]
        ctx_a = torch.cuda.stream(streams[0]) if streams else contextlib.nullcontext()
        ctx_b = torch.cuda.stream(streams[1]) if streams else contextlib.nullcontext()
        try:
            with ctx_a:
                ra = next(gen_a)
        except StopIteration as e:
            ra, res[0] = None, e.value
        try:
            with ctx_b:
                rb = next(gen_b)
        except StopIteration as e:
            rb, res[1] = None, e.value
        while ra is not None or rb is not None:
            if streams:
                if ra is not None:
                    with ctx_a:
                        ya = eng_a._serve(ra)
                        try:
                            ra = gen_a.send(ya)
                        except StopIteration as e:
                            ra, res[0] = None, e.value
                if rb is not None:
                    with ctx_b:
                        yb = eng_b._serve(rb)
                        try:
                            rb = gen_b.send(yb)
                        except StopIteration as e:
                            rb, res[1] = None, e.value
                continue
            if ra is not None and rb is not None:
                la, lb = (ra if isinstance(ra, list) else [ra]), (rb if isinstance(rb, list) else [rb])
                ys = ops.conv2d_group([eng_a._conv_call(r[0], r[1], **r[2]) for r in la] + [eng_b._conv_call(r[0], r[1], **r[2]) for r in lb])
                ya = ys[:len(la)] if isinstance(ra, list) else ys[0]
                yb = ys[len(la):] if isinstance(rb, list) else ys[len(la)]
            elif ra is not None:
                ya = eng_a._serve(ra)
            else:
                yb = eng_b._serve(rb)
            if ra is not None:
                try:
                    ra = gen_a.send(ya)
                except StopIteration as e:
                    ra, res[0] = None, e.value
            if rb is not None:
                try:
                    rb = gen_b.send(yb)
                except StopIteration as e:
                    rb, res[1] = None, e.value
        return res[0], res[1]

    def trunk(self, st_u8: torch.Tensor, sizes, save: bool) -> Ctx:
        """preprocess + ResNet-50 + FPN -> P2..P6.  `save` keeps the activations backward needs."""
        return self._drive(self.trunk_steps(st_u8, sizes, save))

    def prefix_pipelinable(self) -> bool:
        """stem + res2 are frozen (FREEZE_AT = 2: nothing of them is saved for the backward and no update touches their weights) and run as the
        fused bf16 kernels: their forward for batch k + 1 may run while step k still computes (fused_step: cross-step pipelining)"""
        return type(self) is RCNN and self.dtype == torch.bfloat16 and self.fused_stem and self.fused_res2

    def trunk_steps(self, st_u8: torch.Tensor, sizes, save: bool, fpn: bool = True, pre: Optional[torch.Tensor] = None,
                    prefix_out: Optional[torch.Tensor] = None):
        """pre = the res2 output of THIS batch computed earlier (by a `prefix_out` run of the same images): the pass starts at res3.
        prefix_out = run stem + res2 only, the last res2 block writes into this buffer; returns it."""
        W = self.wts
        c = Ctx()
        bu = "backbone.bottom_up."
        if pre is not None:
            x = pre
        elif self.dtype == torch.bfloat16 and self.fused_stem:
            x = ops.stem_pool_forward(st_u8, sizes, W.stem_packed(bu + "stem.conv1"), W.scale(bu + "stem.conv1"), W.shift(bu + "stem.conv1"),
                                      self.p.pixel_mean, self.p.pixel_std)
        else:
            stem = ops.stem_forward(st_u8, sizes, W.w_master(bu + "stem.conv1"), W.scale(bu + "stem.conv1"), W.shift(bu + "stem.conv1"),
                                    self.p.pixel_mean, self.p.pixel_std, self.dtype)
            x = ops.maxpool3s2(stem)
            del stem
        blocks = []
        cs = []
        out_bits = {}                          # data_ptr of a saved block output -> its ReLU bit mask
        # res2 keeps nothing for the backward (FREEZE_AT = 2; `blocks` below starts at res3): each of its bottlenecks is ONE kernel
        # whose two 64-channel intermediate maps stay in the LDS (csrc/bneck.hip)
        fuse2 = self.dtype == torch.bfloat16 and self.fused_res2
        if fuse2:
            res2 = [f"{bu}res2.{b}.conv{k}" for b in range(STAGE_BLOCKS[0]) for k in (1, 2, 3)]
            fw = dict(zip(res2, W.folded(res2)))
        assert prefix_out is None or (fuse2 and pre is None)
        for si, nb in enumerate(STAGE_BLOCKS):
            if si == 0 and pre is not None:
                cs.append(x)
                continue
            for b in range(nb):
                p = f"{bu}res{si + 2}.{b}."
                sc = (yield x, p + "shortcut", {}) if b == 0 else x
                if si == 0 and fuse2:
                    x = ops.bottleneck_fused(x, sc, fw[p + "conv1"], fw[p + "conv2"], fw[p + "conv3"], W.shift(p + "conv1"), W.shift(p + "conv2"),
                                             W.shift(p + "conv3"), out=prefix_out if (prefix_out is not None and b == nb - 1) else None)
                    continue
                # every saved ReLU output's mask as BITS beside it (1/16 of the tensor): what the data-gradient launch that needs the mask -- the
                # next block's conv1 / the lateral conv for a block output, conv3's / conv2's for the two inner maps -- multiplies by instead of
                # reading the whole activation again for its sign; with bits those launches also take the direct epilogue (csrc/igemm.hip)
                want_bits = save and si > 0 and self.mask_bits and self.dtype == torch.bfloat16

                def bits_for(inp, name):
                    if not want_bits:
                        return None
                    lt = W.layout.t[name]
                    ho = (inp.shape[1] + 2 * lt.pad - lt.kk) // lt.stride + 1
                    wo = (inp.shape[2] + 2 * lt.pad - lt.kk) // lt.stride + 1
                    return torch.empty(inp.shape[0] * ho * wo * (lt.wshape[0] // 8), dtype=torch.uint8, device=self.device)
                b1 = bits_for(x, p + "conv1") if self.mask_bits_inner else None
                h1 = yield x, p + "conv1", dict(relu=True, bits_out=b1)
                b2 = bits_for(h1, p + "conv2") if self.mask_bits_inner else None
                h2 = yield h1, p + "conv2", dict(relu=True, bits_out=b2)
                bits = bits_for(h2, p + "conv3")
                out = yield h2, p + "conv3", dict(relu=True, res=sc, res_mode=1, bits_out=bits)
                if save and si > 0:
                    blocks.append((p, x, h1, h2, out, b == 0))
                    if bits is not None:
                        out_bits[out.data_ptr()] = bits
                        if b1 is not None:
                            out_bits[h1.data_ptr()], out_bits[h2.data_ptr()] = b1, b2
                x = out
            cs.append(x)
            if prefix_out is not None:
                return x
        if not fpn:                            # the bare trunk (the Deformable-DETR detector takes C3..C5 themselves)
            if save:
                c.blocks, c.cs, c.out_bits = blocks, cs, out_bits
            else:
                c.cs = cs
            return c
        prev = {}
        P = {}
        prev[5] = yield cs[3], "backbone.fpn_lateral5", {}
        if self.level_groups:
            # the top-down sums first, then the four output convs (256 -> 256, 3x3, on 268800 / 67200 / 16800 / 4200 pixels at N = 4)
            # as ONE launch: alone, the p4 / p5 ones are a fraction of a round of workgroups each
            for lvl in (4, 3, 2):
                prev[lvl] = yield cs[lvl - 2], f"backbone.fpn_lateral{lvl}", dict(res=prev[lvl + 1], res_mode=2)
            outs = yield [(prev[lvl], f"backbone.fpn_output{lvl}", {}) for lvl in (2, 3, 4, 5)]
            P.update(zip((2, 3, 4, 5), outs))
        else:
            P[5] = yield prev[5], "backbone.fpn_output5", {}
            for lvl in (4, 3, 2):
                prev[lvl] = yield cs[lvl - 2], f"backbone.fpn_lateral{lvl}", dict(res=prev[lvl + 1], res_mode=2)
                P[lvl] = yield prev[lvl], f"backbone.fpn_output{lvl}", {}
        P[6] = ops.subsample2(P[5])
        c.P = [P[2], P[3], P[4], P[5], P[6]]
        if save:
            c.blocks, c.cs, c.prev, c.out_bits = blocks, cs, prev, out_bits
        return c

    def rpn_head(self, c: Ctx, save: bool):
        self._drive(self.rpn_head_steps(c, save))

    def rpn_head_steps(self, c: Ctx, save: bool):
        # The hidden maps of the five levels live in ONE buffer, level after level: the 1x1 objectness / delta heads (shared
        # weights, pixel-wise) then run as a single launch over all 358 k pixel positions instead of five (the p2 one alone on
        # the chip, the others far below one round of workgroups) on the student's serial proposal chain.
        px = [f.shape[0] * f.shape[1] * f.shape[2] for f in c.P]
        flat = torch.empty((1, sum(px), 1, FPN_C), dtype=self.dtype, device=self.device)
        ts, o, reqs = [], 0, []
        for f, n in zip(c.P, px):
            view = flat[0, o:o + n, 0].view(f.shape[0], f.shape[1], f.shape[2], FPN_C)
            reqs.append((f, "proposal_generator.rpn_head.conv", dict(relu=True, out=view)))
            o += n
        if self.level_groups:
            ts = yield reqs             # the shared 3x3 conv on the five levels: one launch
        else:
            for r in reqs:
                ts.append((yield r))
        hf = yield flat, "rpn_head_out", dict(want_f32=True)
        heads, o = [], 0
        for f, n in zip(c.P, px):
            heads.append(hf[0, o:o + n, 0].view(f.shape[0], f.shape[1], f.shape[2], hf.shape[3]))
            o += n
        c.head, c.head_flat = heads, hf
        if save:
            c.rpn_t = ts

    def box_head(self, pooled: torch.Tensor, c: Optional[Ctx] = None):
        R = pooled.shape[0]
        x = pooled.view(R, 1, 1, POOL * POOL * FPN_C)
        fc1 = self.conv(x, "roi_heads.box_head.fc1", relu=True)
        # FC2's ReLU mask as bits for the predictors' data gradient (1/16 of the bytes, and that launch takes the direct epilogue); FC1 is a
        # split-K launch whose second pass does not write bits: its mask stays the activation itself
        bits2 = None
        if c is not None and self.mask_bits and self.dtype == torch.bfloat16 and R > 0 and os.environ.get("ALDI_MASK_BITS_FC", "1") == "1":
            bits2 = torch.empty(R * (FC_DIM // 8), dtype=torch.uint8, device=self.device)
        fc2 = self.conv(fc1, "roi_heads.box_head.fc2", relu=True, bits_out=bits2)
        pred = self.conv(fc2, "box_pred", want_f32=True).view(R, self.Cp)
        if c is not None:
            c.pooled, c.fc1, c.fc2 = pooled, fc1, fc2
            if bits2 is not None:
                if c.get("out_bits") is None:
                    c.out_bits = {}
                c.out_bits[fc2.data_ptr()] = bits2
        return pred, fc2

    def roi_feats(self, c: Ctx, grads=None):
        return ops.make_roi_feats(c.P[:4], grads, [1.0 / s for s in STRIDES[:4]])

    # ------------------------------------------------------------------ sampling (host RNG, D2 order)
    def _sample_host(self, counts: List[List[int]], batch: int, frac: float):
        """subsample_labels for every image: two torch.randperm draws per image (pos then neg). Host tensors."""
        N = len(counts)
        S = batch
        sel = torch.zeros((N, 2, S), dtype=torch.int32)
        nsel = torch.zeros((N, 2), dtype=torch.int32)
        for n, (npos, nneg) in enumerate(counts):
            num_pos = min(npos, int(batch * frac))
            num_neg = min(nneg, batch - num_pos)
            perm1 = randperm_prefix(npos, num_pos)
            perm2 = randperm_prefix(nneg, num_neg)
            sel[n, 0, :num_pos] = perm1.to(torch.int32)
            sel[n, 1, :num_neg] = perm2.to(torch.int32)
            nsel[n, 0], nsel[n, 1] = num_pos, num_neg
        return sel, nsel, nsel.tolist()

    def _sample(self, counts: List[List[int]], batch: int, frac: float):
        sel, nsel, h = self._sample_host(counts, batch, frac)
        sel_d, nsel_d = ops.upload_packed([sel, nsel], self.device)
        return sel_d, nsel_d, h

    def rpn_match(self, geom, anchors, gt, N):
        """Matcher(0.3/0.7, low-quality) on the anchors: labels before sampling, matched GT index, ordered pos/neg lists."""
        sumA = anchors.shape[0]
        dev = self.device
        best_iou = torch.empty((N, sumA), dtype=torch.float32, device=dev)
        best_idx = torch.empty((N, sumA), dtype=torch.int32, device=dev)
        labels = torch.empty((N, sumA), dtype=torch.int32, device=dev)
        scratch = ops.box_match_scratch(N, GMAX, dev)
        ops.box_match(anchors, 0, None, sumA, gt["boxes"], gt["count"], GMAX, N, self.p.rpn_iou[0], self.p.rpn_iou[1], True, best_iou, best_idx, scratch, labels)
        lists = torch.empty((N, 2, sumA), dtype=torch.int32, device=dev)
        counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
        ops.compact_labels(labels, sumA, N, 0, lists, counts)
        return labels, best_idx, lists, counts

    def rpn_sample(self, lists, counts, N, host_counts=None):
        """host draws (2 randperm per image) -> NEW label tensor in {-1,0,1}; returns (labels, n_valid, n_fg, host_counts)."""
        if host_counts is None:
            both = torch.cat([counts.view(-1), self.err.view(-1)]).cpu().tolist()   # device->host sync: the RNG needs the list lengths
            raise_on_error(both[-1])
            host_counts = [both[2 * i: 2 * i + 2] for i in range(N)]
        lists = lists.contiguous()
        sel, nsel, nsel_h = self._sample(host_counts, self.p.rpn_batch, self.p.rpn_pos_frac)
        L_ = lists.shape[2]
        labels = torch.empty((N, L_), dtype=torch.int32, device=self.device)
        ops.rpn_apply_sample(labels, L_, N, lists, sel, nsel, self.p.rpn_batch)
        n_fg = sum(a for a, _ in nsel_h)
        n_valid = sum(a + b for a, b in nsel_h)
        return labels, n_valid, n_fg, host_counts

    def proposals(self, c: Ctx, geom, anchors, hw, N, training: bool):
        nl = 5
        ws = self.workspace("rpn", ops.rpn_proposals_workspace(N, nl))
        post = self.p.rpn_post[0 if training else 1]
        boxes = torch.empty((N, post, 4), dtype=torch.float32, device=self.device)
        scores = torch.empty((N, post), dtype=torch.float32, device=self.device)
        count = torch.empty((N,), dtype=torch.int32, device=self.device)
        ops.rpn_proposals(geom, c.head, anchors, hw, N, self.p.rpn_pre[0 if training else 1], post, self.p.rpn_nms, ws, boxes, scores, count, self.err)
        return boxes, scores, count

    # ------------------------------------------------------------------ training forward
    def forward_train(self, images, instances, *, roi_seed: Optional[int] = None, pre_roi_hook=None,
                      gt_dev: Optional[dict] = None, do_align: bool = False, labeled: bool = True,
                      da_weights: Tuple[float, float] = (0.0, 0.0)) -> Ctx:
        """GeneralizedRCNN.forward (training) + AlignMixin.forward (aldi/align.py:71-101): activations
        and (unscaled) loss VALUES.  Gradients are produced later by ``backward(c, scales)``."""
        N = len(images)
        st, sizes, hw = self.stage_images(images)
        shapes, geom, anchors = self.geometry(st.shape[2], st.shape[3])
        gt = gt_dev if gt_dev is not None else self.stage_gt(instances)
        c = self.trunk(st, sizes, save=True)
        c.N, c.sizes, c.hw, c.geom, c.anchors, c.gt, c.shapes = N, sizes, hw, geom, anchors, gt, shapes
        self.rpn_head(c, save=True)
        dev = self.device
        # --- RPN labels + losses
        _, matched, lists, counts = self.rpn_match(geom, anchors, gt, N)
        labels, _, _, c.rpn_host_counts = self.rpn_sample(lists, counts, N)
        c.rpn_labels, c.rpn_matched, c.rpn_lists, c.rpn_counts = labels, matched, lists, counts
        c.loss_rpn = torch.zeros(2, dtype=torch.float32, device=dev)
        ops.rpn_loss(geom, c.head, None, anchors, labels, matched, gt["boxes"], gt["count"], GMAX, N, 1.0 / (self.p.rpn_batch * N), 0.0, 0.0, c.loss_rpn)
        # --- proposals (detached)
        c.props, c.prop_scores, c.prop_count = self.proposals(c, geom, anchors, hw, N, training=True)
        # --- ROI heads
        if pre_roi_hook is not None:
            pre_roi_hook()                                   # forward pre-hooks on roi_heads (ManualSeed, aldi/helpers.py:25-26)
        if roi_seed is not None:
            torch.manual_seed(roi_seed)
        self.roi_sample(c, c.props, c.prop_count, gt, N)
        self.roi_forward(c)
        c.loss_box = torch.zeros(2, dtype=torch.float32, device=dev)
        ops.box_loss(c.pred, self.Cp, self.K, c.R, c.rois, c.r_cls, c.r_gt, self.p.roi_weights, 0.0, 0.0, None, c.loss_box)
        c.align = {}
        c.distill = None
        c.labeled, c.da_weights = labeled, da_weights
        if do_align:
            self.align_forward(c, labeled, da_weights)
        return c

    # ------------------------------------------------------------------ fused multi-chunk training forward
    def forward_train_fused(self, specs: List[dict]) -> Ctx:
        """Several micro-batches ("chunks") of the reference schedule through ONE trunk / head launch sequence
        (FrozenBN => no cross-image coupling, so batching the source and target student
        passes is exact).  Each spec: dict(images, instances | gt_dev, labeled, do_align, pre_rpn, pre_roi) where
        pre_rpn / pre_roi are host callbacks fired right before that chunk's RPN / ROI sampling draws, which keeps
        the global torch RNG stream identical to the sequential schedule.  Losses are reduced per chunk."""
        dev = self.device
        images = [im for sp in specs for im in sp["images"]]
        N = len(images)
        st, sizes, hw = self.stage_images(images)
        shapes, geom, anchors = self.geometry(st.shape[2], st.shape[3])
        # everything that does not need ground truth first: a chunk's GT may be the pseudo-labels of a teacher inference
        # still running on another stream (spec["gt_wait"] joins it)
        c = self.trunk(st, sizes, save=True)
        c.N, c.sizes, c.hw, c.geom, c.anchors, c.shapes = N, sizes, hw, geom, anchors, shapes
        self.rpn_head(c, save=True)
        # proposal generation (top-k, NMS: latency-bound, a handful of workgroups) runs beside anchor matching / label
        # compaction on the second stream; ROI preparation needs both
        side = self._wgrad_stream()
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                c.props, c.prop_scores, c.prop_count = self.proposals(c, geom, anchors, hw, N, training=True)
        else:
            c.props, c.prop_scores, c.prop_count = self.proposals(c, geom, anchors, hw, N, training=True)
        for sp in specs:
            if sp.get("gt_lazy") is not None:              # e.g. launch the teacher now, behind the student's label-free work
                sp["gt_dev"] = sp["gt_lazy"]()
        for sp in specs:
            if sp.get("gt_wait") is not None:
                sp["gt_wait"]()
        gts = [sp["gt_dev"] if sp.get("gt_dev") is not None else self.stage_gt(sp["instances"]) for sp in specs]
        gt = {k: torch.cat([g[k] for g in gts]) for k in ("boxes", "classes", "count")}
        c.gt = gt
        _, matched, lists, counts = self.rpn_match(geom, anchors, gt, N)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        prep = self._roi_prepare(c.props, c.prop_count, gt, N)
        both = torch.cat([counts.view(-1), prep["counts"].view(-1), self.err.view(-1)]).cpu().tolist()      # the ONE device->host sync of the student pass
        raise_on_error(both[-1], "student")
        rpn_counts = [both[2 * i: 2 * i + 2] for i in range(N)]
        roi_counts = [both[2 * N + 2 * i: 2 * N + 2 * i + 2] for i in range(N)]
        # host draws, chunk by chunk, in the sequential schedule's order
        rsel, rnsel, rh, osel, onsel, oh = [], [], [], [], [], []
        n0 = 0
        chunks = []
        for sp in specs:
            n1 = n0 + len(sp["images"])
            if sp.get("pre_rpn"):
                sp["pre_rpn"]()
            a, b, h_ = self._sample_host(rpn_counts[n0:n1], self.p.rpn_batch, self.p.rpn_pos_frac)
            rsel.append(a); rnsel.append(b); rh += h_
            if sp.get("pre_roi"):
                sp["pre_roi"]()
            a, b, h2 = self._sample_host(roi_counts[n0:n1], self.p.roi_batch, self.p.roi_pos_frac)
            osel.append(a); onsel.append(b); oh += h2
            chunks.append(dict(n0=n0, n1=n1, labeled=sp.get("labeled", True), do_align=sp.get("do_align", False),
                               da_weights=sp.get("da_weights", (0.0, 0.0)), rpn_counts=rpn_counts[n0:n1], roi_counts=roi_counts[n0:n1]))
            n0 = n1
        labels = torch.empty((N, anchors.shape[0]), dtype=torch.int32, device=dev)
        rsel_d, rnsel_d, osel_d, onsel_d = ops.upload_packed([torch.cat(rsel), torch.cat(rnsel), torch.cat(osel), torch.cat(onsel)], dev)
        ops.rpn_apply_sample(labels, anchors.shape[0], N, lists, rsel_d, rnsel_d, self.p.rpn_batch)
        c.rpn_labels, c.rpn_matched, c.rpn_lists, c.rpn_counts = labels, matched, lists, counts
        self._roi_gather(c, prep, osel_d, onsel_d, oh, gt, N)
        r0 = 0
        for ch, sp in zip(chunks, specs):                   # e.g. the teacher's box head on this chunk's ROIs, on its own stream
            r1 = r0 + sum(c.rows[ch["n0"]:ch["n1"]])
            if sp.get("post_rois") is not None:
                sp["post_rois"](c, ch["n0"], r0, r1)
            r0 = r1
        self.roi_forward(c)
        # per-chunk loss values
        r0 = 0
        for ch in chunks:
            n0, n1 = ch["n0"], ch["n1"]
            r1 = r0 + sum(c.rows[n0:n1])
            ch["r0"], ch["r1"] = r0, r1
            ch["loss_rpn"] = torch.zeros(2, dtype=torch.float32, device=dev)
            ch["loss_box"] = torch.zeros(2, dtype=torch.float32, device=dev)
            nc = n1 - n0
            ops.rpn_loss(geom, [h[n0:n1] for h in c.head], None, anchors, labels[n0:n1], matched[n0:n1], gt["boxes"][n0:n1], gt["count"][n0:n1],
                         GMAX, nc, 1.0 / (self.p.rpn_batch * nc), 0.0, 0.0, ch["loss_rpn"])
            ops.box_loss(c.pred[r0:r1], self.Cp, self.K, r1 - r0, c.rois[r0:r1], c.r_cls[r0:r1], c.r_gt[r0:r1], self.p.roi_weights, 0.0, 0.0, None, ch["loss_box"])
            ch["align"] = {}
            ch["distill"] = None
            if ch["do_align"]:
                self._align_forward_chunk(c, ch)
            r0 = r1
        c.chunks = chunks
        c.align = {}
        c.distill = None
        return c

    def _align_forward_chunk(self, c: Ctx, ch: dict):
        dev = self.device
        label = 1.0 if ch["labeled"] else 0.0
        n0, n1, r0, r1 = ch["n0"], ch["n1"], ch["r0"], ch["r1"]
        if self.has_img_da:
            acts, pooled, logit = self._img_disc_forward(c.P[self.img_da_level][n0:n1])
            ch["loss_da_img"] = torch.zeros(1, dtype=torch.float32, device=dev)
            ops.domain_bce(logit, logit.shape[-1], n1 - n0, label, ch["da_weights"][0], 0.0, None, ch["loss_da_img"])
            ch["align"]["img"] = (acts, pooled, logit)
        if self.has_ins_da and r1 > r0:
            acts, logit = self._ins_disc_forward(c.fc2[r0:r1])
            ch["loss_da_ins"] = torch.zeros(1, dtype=torch.float32, device=dev)
            ops.domain_bce(logit, logit.shape[-1], r1 - r0, label, ch["da_weights"][1], 0.0, None, ch["loss_da_ins"])
            ch["align"]["ins"] = (acts, logit)

    def _img_disc_forward(self, feat: torch.Tensor):
        """ConvDiscriminator (aldi/align.py:103-119): [Conv2d(k=3, no padding) + ReLU] per hidden dim -> global average -> Linear"""
        acts = [feat]
        for name in self.img_da_layers[:-1]:
            acts.append(self.conv(acts[-1], name, relu=True))
        pooled = ops.avgpool(acts[-1])
        return acts, pooled, self.conv(pooled, self.img_da_layers[-1], want_f32=True)

    def _ins_disc_forward(self, feat: torch.Tensor):
        """FCDiscriminator (aldi/align.py:121-135): [Linear + ReLU] per hidden dim -> Linear"""
        acts = [feat]
        for name in self.ins_da_layers[:-1]:
            acts.append(self.conv(acts[-1], name, relu=True))
        return acts, self.conv(acts[-1], self.ins_da_layers[-1], want_f32=True)

    def distill_forward_chunk(self, c: Ctx, ch: dict, teacher_head, teacher_pred, labels, n_valid, n_fg, values=True, **kw):
        """distillation losses of one chunk of a fused forward (teacher tensors cover exactly that chunk).  values = False: only
        describe the losses; `backward_fused` then writes their values while it computes their gradients (ch["values_in_backward"])."""
        dev = self.device
        n0, n1, r0, r1 = ch["n0"], ch["n1"], ch["r0"], ch["r1"]
        ch["distill"] = dict(t_head=teacher_head, t_pred=teacher_pred, labels=labels, n_valid=n_valid, n_fg=n_fg, **kw)
        if not values:
            return
        ch["loss_dist_rpn"] = torch.zeros(2, dtype=torch.float32, device=dev)
        ch["loss_dist_roi"] = torch.zeros(2, dtype=torch.float32, device=dev)
        ops.rpn_distill_loss(c.geom, [h[n0:n1] for h in c.head], teacher_head, None, labels, n1 - n0, kw["obj_T"], n_valid, n_fg,
                             kw["do_obj"], kw["do_rpn_reg"], 0.0, ch["loss_dist_rpn"], counts_dev=kw.get("counts_dev"))
        ops.roih_distill_loss(c.pred[r0:r1], teacher_pred, self.Cp, self.K, r1 - r0, kw["cls_T"], kw["kl"], kw["do_cls"], kw["do_roih_reg"],
                              0.0, None, ch["loss_dist_roi"])

    def chunk_loss_dict(self, ch: dict) -> "OrderedDict[str, torch.Tensor]":
        d = OrderedDict()
        d["loss_cls"], d["loss_box_reg"] = ch["loss_box"][0], ch["loss_box"][1]
        d["loss_rpn_cls"], d["loss_rpn_loc"] = ch["loss_rpn"][0], ch["loss_rpn"][1]
        if "img" in ch["align"]:
            d["loss_da_img"] = ch["loss_da_img"][0]
        if "ins" in ch["align"]:
            d["loss_da_ins"] = ch["loss_da_ins"][0]
        return d

    def chunk_distill_loss_dict(self, ch: dict) -> "OrderedDict[str, torch.Tensor]":
        d = OrderedDict()
        k = ch["distill"]
        if k["do_obj"]:
            d["loss_obj_bce"] = ch["loss_dist_rpn"][0]
        if k["do_rpn_reg"]:
            d["loss_rpn_l1"] = ch["loss_dist_rpn"][1]
        if k["do_cls"]:
            d["loss_cls_ce"] = ch["loss_dist_roi"][0]
        if k["do_roih_reg"]:
            d["loss_roih_l1"] = ch["loss_dist_roi"][1]
        return d

    def backward_fused(self, c: Ctx, scales: List[Dict[str, float]], after_losses=None):
        """backward of forward_train_fused: per-chunk loss gradients (scales[i] for chunk i) into the shared head-gradient
        buffers, then ONE pass heads -> FPN -> res5..res3 over all images."""
        T, dev = self.dtype, self.device
        # The loss kernels produce value AND gradient in one pass.  Chunks flagged `values_in_backward` (the fused step) take their
        # loss values from THIS pass -- no separate value launches on the chain between the box head and the backward -- and all
        # their small zero-initialised outputs come out of one arena (one fill instead of one per tensor).
        early = c.get("rpn_early")
        if early is not None and early.get("arena_box") is not None and self._aux_stream() is not None:
            arena, c.gpred = early["arena_box"], early["gpred"]          # (zeroed on the auxiliary stream: ev_zero)
            torch.cuda.current_stream().wait_event(early["ev_zero"])
        else:
            arena = torch.zeros(2 + 8 * len(c.chunks), dtype=torch.float32, device=dev)
            c.gpred = torch.zeros((max(c.R, 1), self.Cp), dtype=torch.float32, device=dev)
        scratch = arena[:2]
        # The RPN-side loss kernels (RPN losses, RPN distillation) read the head outputs of phase A and the sampled labels only: given buffers
        # and an event from BEFORE the box head's forward (the fused step's `rpn_early`), they run on the auxiliary stream beside RoIAlign and
        # FC1 instead of behind the box head on the chain into the backward (~45 us of small launches).
        aux0 = self._aux_stream() if early is not None else None
        if aux0 is None:
            early = None
        rpn_ctx = (lambda: torch.cuda.stream(aux0)) if early is not None else contextlib.nullcontext
        if early is not None and early.get("ev") is not None:
            aux0.wait_event(early["ev"])
        hf = c.get("head_flat")
        with rpn_ctx():
            rpn_arena = arena
            if early is not None:
                rpn_arena = early["arena"]
                rpn_arena.zero_()
            if hf is not None:                                   # one fill for the five levels' head gradients (views like c.head)
                gf, o = (early["gf"].zero_() if early is not None else torch.zeros_like(hf)), 0
                c.ghead = []
                for h in c.head:
                    n = h.shape[0] * h.shape[1] * h.shape[2]
                    c.ghead.append(gf[0, o:o + n, 0].view(h.shape))
                    o += n
            else:
                c.ghead = [torch.zeros_like(h) for h in c.head]
        gt = c.gt
        align_list = []
        # the box head's losses of all chunks (+ RoI distillation + the compute-dtype copy of the gradient rows) as ONE launch: csrc/roi.hip
        box_fused = [] if (c.R > 0 and len(c.chunks) <= 8 and self.Cp % 2 == 0 and os.environ.get("ALDI_BOX_LOSS_FUSED", "1") == "1") else None
        c.gpred_lo = None
        for ci, (ch, sc_) in enumerate(zip(c.chunks, scales)):
            sc = lambda k: float(sc_.get(k, 0.0))
            n0, n1, r0, r1 = ch["n0"], ch["n1"], ch["r0"], ch["r1"]
            nc = n1 - n0
            heads = [h[n0:n1] for h in c.head]
            gheads = [g[n0:n1] for g in c.ghead]
            l_rpn = l_drpn = rpn_arena[:2]
            l_box = l_droi = scratch
            if ch.get("values_in_backward"):
                base = 2 + 8 * ci
                l_rpn, l_drpn = rpn_arena[base: base + 2], rpn_arena[base + 4: base + 6]
                l_box, l_droi = arena[base + 2: base + 4], arena[base + 6: base + 8]
                ch["loss_rpn"], ch["loss_box"] = l_rpn, l_box
                if ch["distill"] is not None:
                    ch["loss_dist_rpn"], ch["loss_dist_roi"] = l_drpn, l_droi
            with rpn_ctx():
                ops.rpn_loss(c.geom, heads, gheads, c.anchors, c.rpn_labels[n0:n1], c.rpn_matched[n0:n1], gt["boxes"][n0:n1], gt["count"][n0:n1],
                             GMAX, nc, 1.0 / (self.p.rpn_batch * nc), sc("loss_rpn_cls"), sc("loss_rpn_loc"), l_rpn)
            if box_fused is not None:
                box_fused.append(dict(r0=r0, r1=r1, gs_cls=sc("loss_cls"), gs_box=sc("loss_box_reg"), loss_box=l_box))
            else:
                ops.box_loss(c.pred[r0:r1], self.Cp, self.K, r1 - r0, c.rois[r0:r1], c.r_cls[r0:r1], c.r_gt[r0:r1], self.p.roi_weights,
                             sc("loss_cls"), sc("loss_box_reg"), c.gpred[r0:r1], l_box)
            d = ch["distill"]
            if d is not None:
                def rpn_d(do_obj, do_reg, s_):
                    ops.rpn_distill_loss(c.geom, heads, d["t_head"], gheads, d["labels"], nc, d["obj_T"], d["n_valid"], d["n_fg"], do_obj, do_reg, s_, l_drpn,
                                         counts_dev=d.get("counts_dev"))

                if d.get("t_ev") is not None:
                    torch.cuda.current_stream().wait_event(d["t_ev"])       # the teacher's box head (its own stream)
                if box_fused is not None:
                    box_fused[-1].update(t_pred=d["t_pred"], cls_T=d["cls_T"], kl=d["kl"], do_cls=d["do_cls"], do_reg=d["do_roih_reg"],
                                         gs_dcls=sc("loss_cls_ce"), gs_dreg=sc("loss_roih_l1"), loss_d=l_droi)

                def roi_d(do_cls, do_reg, s_):
                    if box_fused is not None:
                        return
                    ops.roih_distill_loss(c.pred[r0:r1], d["t_pred"], self.Cp, self.K, r1 - r0, d["cls_T"], d["kl"], do_cls, do_reg, s_, c.gpred[r0:r1], l_droi)
                with rpn_ctx():
                    if sc("loss_obj_bce") == sc("loss_rpn_l1"):
                        rpn_d(d["do_obj"], d["do_rpn_reg"], sc("loss_obj_bce"))
                    else:
                        rpn_d(d["do_obj"], False, sc("loss_obj_bce"))
                        rpn_d(False, d["do_rpn_reg"], sc("loss_rpn_l1"))
                if sc("loss_cls_ce") == sc("loss_roih_l1"):
                    roi_d(d["do_cls"], d["do_roih_reg"], sc("loss_cls_ce"))
                else:
                    roi_d(d["do_cls"], False, sc("loss_cls_ce"))
                    roi_d(False, d["do_roih_reg"], sc("loss_roih_l1"))
            label = 1.0 if ch["labeled"] else 0.0
            al = dict(n0=n0, n1=n1, r0=r0, r1=r1)
            if "img" in ch["align"]:
                acts, pooled, logit = ch["align"]["img"]
                glog = torch.empty(logit.shape, dtype=T, device=dev)
                ops.domain_bce(logit, logit.shape[-1], nc, label, ch["da_weights"][0], sc("loss_da_img"), glog, scratch)
                al["img"] = (acts, pooled, glog)
            if "ins" in ch["align"]:
                acts, logit = ch["align"]["ins"]
                glog = torch.empty(logit.shape, dtype=T, device=dev)
                ops.domain_bce(logit, logit.shape[-1], r1 - r0, label, ch["da_weights"][1], sc("loss_da_ins"), glog, scratch)
                al["ins"] = (acts, glog)
            if "img" in al or "ins" in al:
                align_list.append(al)
        if box_fused:
            if T == torch.bfloat16:
                c.gpred_lo = torch.empty((c.R, self.Cp), dtype=T, device=dev)
            ops.box_losses_fused(c.pred, self.Cp, self.K, c.rois, c.r_cls, c.r_gt, self.p.roi_weights, box_fused, c.gpred, c.gpred_lo)
        if early is not None:
            torch.cuda.current_stream().wait_stream(aux0)    # (the RPN-side values and head gradients: finished long ago)
        if after_losses is not None:
            after_losses()                                   # (the loss values are final here: the caller's logging branches off)
        self._backward_trunk(c, align_list)

    def align_forward(self, c: Ctx, labeled: bool, da_weights):
        """AlignMixin.forward (aldi/align.py:75-90): discriminators behind gradient reversal, BCE vs constant domain label."""
        dev = self.device
        label = 1.0 if labeled else 0.0
        if self.has_img_da:
            acts, pooled, logit = self._img_disc_forward(c.P[self.img_da_level])
            c.loss_da_img = torch.zeros(1, dtype=torch.float32, device=dev)
            ops.domain_bce(logit, logit.shape[-1], c.N, label, da_weights[0], 0.0, None, c.loss_da_img)
            c.align["img"] = (acts, pooled, logit)
        if self.has_ins_da and c.R > 0:
            acts, logit = self._ins_disc_forward(c.fc2)
            c.loss_da_ins = torch.zeros(1, dtype=torch.float32, device=dev)
            ops.domain_bce(logit, logit.shape[-1], c.R, label, da_weights[1], 0.0, None, c.loss_da_ins)
            c.align["ins"] = (acts, logit)

    def distill_forward(self, c: Ctx, teacher_head: List[torch.Tensor], teacher_pred: torch.Tensor, labels: torch.Tensor,
                        n_valid: int, n_fg: int, *, obj_T: float, cls_T: float, kl: bool,
                        do_obj: bool, do_rpn_reg: bool, do_cls: bool, do_roih_reg: bool):
        """ALDIDistiller.get_rpn_losses + get_roih_losses (aldi/distill.py:193-278): loss values now, gradients in backward."""
        dev = self.device
        c.distill = dict(t_head=teacher_head, t_pred=teacher_pred, labels=labels, n_valid=n_valid, n_fg=n_fg, obj_T=obj_T, cls_T=cls_T,
                         kl=kl, do_obj=do_obj, do_rpn_reg=do_rpn_reg, do_cls=do_cls, do_roih_reg=do_roih_reg)
        c.loss_dist_rpn = torch.zeros(2, dtype=torch.float32, device=dev)
        c.loss_dist_roi = torch.zeros(2, dtype=torch.float32, device=dev)
        ops.rpn_distill_loss(c.geom, c.head, teacher_head, None, labels, c.N, obj_T, n_valid, n_fg, do_obj, do_rpn_reg, 0.0, c.loss_dist_rpn)
        ops.roih_distill_loss(c.pred, teacher_pred, self.Cp, self.K, c.R, cls_T, kl, do_cls, do_roih_reg, 0.0, None, c.loss_dist_roi)

    def loss_dict(self, c: Ctx) -> "OrderedDict[str, torch.Tensor]":
        """0-d device tensors in the key order of GeneralizedRCNN.forward (+ AlignMixin)."""
        d = OrderedDict()
        d["loss_cls"], d["loss_box_reg"] = c.loss_box[0], c.loss_box[1]
        d["loss_rpn_cls"], d["loss_rpn_loc"] = c.loss_rpn[0], c.loss_rpn[1]
        if "img" in c.align:
            d["loss_da_img"] = c.loss_da_img[0]
        if "ins" in c.align:
            d["loss_da_ins"] = c.loss_da_ins[0]
        return d

    def distill_loss_dict(self, c: Ctx) -> "OrderedDict[str, torch.Tensor]":
        d = OrderedDict()
        k = c.distill
        if k["do_obj"]:
            d["loss_obj_bce"] = c.loss_dist_rpn[0]
        if k["do_rpn_reg"]:
            d["loss_rpn_l1"] = c.loss_dist_rpn[1]
        if k["do_cls"]:
            d["loss_cls_ce"] = c.loss_dist_roi[0]
        if k["do_roih_reg"]:
            d["loss_roih_l1"] = c.loss_dist_roi[1]
        return d

    def _roi_prepare(self, props, prop_count, gt, N, tail=None) -> dict:
        """append GT, match (IoU >= 0.5), classes, ordered fg/bg lists + their counts (device)."""
        dev = self.device
        P = props.shape[1]
        Lc = P + GMAX
        cand = torch.empty((N, Lc, 4), dtype=torch.float32, device=dev)
        ccount = torch.empty((N,), dtype=torch.int32, device=dev)
        best_iou = torch.empty((N, Lc), dtype=torch.float32, device=dev)
        best_idx = torch.empty((N, Lc), dtype=torch.int32, device=dev)
        labels = torch.empty((N, Lc), dtype=torch.int32, device=dev)
        cls = torch.empty((N, Lc), dtype=torch.int32, device=dev)
        lists = torch.empty((N, 2, Lc), dtype=torch.int32, device=dev)
        fused = os.environ.get("ALDI_ROI_PREPARE_FUSED", "1") == "1" and GMAX <= 256
        # tail = (word_a, word_b): two device words the fused launch appends to the counts (the fused step's error words: one copy to the host)
        cbuf = torch.empty((2 * N + 2,), dtype=torch.int32, device=dev) if (tail is not None and fused) else None
        counts = cbuf[: 2 * N].view(N, 2) if cbuf is not None else torch.empty((N, 2), dtype=torch.int32, device=dev)
        if fused:
            # one launch instead of eight on the chain that ends in the list lengths the host waits for (same results: tests/test_kernels_gpu.py)
            tk = self._ws.get("roi_tickets")
            if tk is None or tk.numel() < N:
                tk = self._ws["roi_tickets"] = torch.zeros(max(N, 64), dtype=torch.int32, device=dev)
            ops.roi_prepare_lists(props, prop_count, P, gt["boxes"], gt["classes"], gt["count"], GMAX, N, self.K, self.p.roi_iou, cand, ccount, best_iou,
                                  best_idx, labels, cls, lists, counts, tk, *(tail if cbuf is not None else ()))
        else:
            scratch = torch.empty((N, GMAX), dtype=torch.int32, device=dev)
            ops.roi_prepare(props, prop_count, P, gt["boxes"], gt["classes"], gt["count"], GMAX, N, self.K, self.p.roi_iou, cand, ccount, best_iou, best_idx,
                            scratch, labels, cls)
            ops.compact_labels(cls, Lc, N, self.K, lists, counts)
        return dict(cand=cand, cls=cls, best_idx=best_idx, lists=lists, counts=counts, Lc=Lc, counts_tail=cbuf)

    def _roi_gather(self, c: Ctx, prep: dict, sel, nsel, nsel_h, gt, N, row_off_dev=None):
        dev = self.device
        rows = [a + b for a, b in nsel_h]
        row_off = [0]
        for r in rows[:-1]:
            row_off.append(row_off[-1] + r)
        R = sum(rows)
        c.R, c.rows = R, rows
        c.rois = torch.empty((max(R, 1), 5), dtype=torch.float32, device=dev)
        c.r_cls = torch.empty((max(R, 1),), dtype=torch.int32, device=dev)
        c.r_gt = torch.empty((max(R, 1), 4), dtype=torch.float32, device=dev)
        c.r_idx = torch.empty((max(R, 1),), dtype=torch.int32, device=dev)
        if row_off_dev is None:
            row_off_dev = torch.tensor(row_off, dtype=torch.int32).to(dev)
        ops.roi_gather(prep["cand"], prep["cls"], prep["best_idx"], prep["Lc"], prep["lists"], sel, nsel, self.p.roi_batch,
                       row_off_dev, gt["boxes"], gt["count"], GMAX, N, c.rois, c.r_cls, c.r_gt, c.r_idx)

    def roi_sample(self, c: Ctx, props, prop_count, gt, N):
        prep = self._roi_prepare(props, prop_count, gt, N)
        host_counts = prep["counts"].cpu().tolist()          # device->host sync (RNG needs the list lengths)
        c.roi_host_counts = host_counts
        sel, nsel, nsel_h = self._sample(host_counts, self.p.roi_batch, self.p.roi_pos_frac)
        self._roi_gather(c, prep, sel, nsel, nsel_h, gt, N)

    def roi_forward(self, c: Ctx):
        R = c.R
        pooled = torch.empty((R, POOL, POOL, FPN_C), dtype=self.dtype, device=self.device)
        ops.roialign(self.roi_feats(c), c.rois, R, POOL, pooled, backward=False)
        c.pred, _ = self.box_head(pooled, c)

    def box_head_on(self, c_feats: Ctx, rois: torch.Tensor, R: int):
        """teacher-side: box head on given rois (the student's sampled proposals)."""
        pooled = torch.empty((R, POOL, POOL, FPN_C), dtype=self.dtype, device=self.device)
        ops.roialign(self.roi_feats(c_feats), rois, R, POOL, pooled, backward=False)
        pred, fc2 = self.box_head(pooled)
        return pred

    # ------------------------------------------------------------------ inference (teacher)
    def inference(self, images, pl_thresh: float, keep_ctx: bool = True, staged=None, pl_out=None) -> Ctx:
        """GeneralizedRCNN.inference(do_postprocess=False) + process_pseudo_label threshold
        (aldi/pseudolabeler.py:15-67).  Everything stays on device.  `staged` = (uint8 batch, sizes, hw) already in HBM."""
        st, sizes, hw = staged if staged is not None else self.stage_images(images)
        c = self.trunk(st, sizes, save=False)
        self.rpn_head(c, save=False)
        return self.inference_heads(c, st, sizes, hw, pl_thresh, pl_out=pl_out)

    def inference_heads(self, c: Ctx, st, sizes, hw, pl_thresh: float, pl_out=None) -> Ctx:
        """everything of the inference pass after the trunk and the RPN head (which may have run paired with another model's).
        pl_out = (boxes [N][GMAX][4], classes [N][GMAX], count [N]): write the pseudo-labels straight into the caller's
        ground-truth slots (the fused step's: no copies / concatenations between the detections and the anchor matcher)."""
        N = st.shape[0]
        shapes, geom, anchors = self.geometry(st.shape[2], st.shape[3])
        c.N, c.sizes, c.hw, c.geom, c.anchors, c.shapes = N, sizes, hw, geom, anchors, shapes
        props, pscores, pcount = self.proposals(c, geom, anchors, hw, N, training=False)
        P = props.shape[1]
        rois = torch.empty((N * P, 5), dtype=torch.float32, device=self.device)
        ops.rois_from_proposals(props, pcount, P, N, rois)
        pooled = torch.empty((N * P, POOL, POOL, FPN_C), dtype=self.dtype, device=self.device)
        ops.roialign(self.roi_feats(c), rois, N * P, POOL, pooled, backward=False)
        pred, _ = self.box_head(pooled)
        dev = self.device
        ws = self.workspace("det", ops.detections_workspace(N))
        d = Ctx()
        d.boxes = torch.empty((N, self.p.dets, 4), dtype=torch.float32, device=dev)
        d.scores = torch.empty((N, self.p.dets), dtype=torch.float32, device=dev)
        d.classes = torch.empty((N, self.p.dets), dtype=torch.int32, device=dev)
        d.count = torch.empty((N,), dtype=torch.int32, device=dev)
        if pl_out is not None:
            pl_boxes, pl_cls, pl_count = pl_out
        else:
            pl_boxes = torch.empty((N, GMAX, 4), dtype=torch.float32, device=dev)
            pl_cls = torch.empty((N, GMAX), dtype=torch.int32, device=dev)
            pl_count = torch.empty((N,), dtype=torch.int32, device=dev)
        pl_scores = torch.empty((N, GMAX), dtype=torch.float32, device=dev)
        # the kernel writes rows of GMAX entries and clears what it does not fill
        ops.detections(pred, self.Cp, self.K, props, pcount, P, N, hw, self.p.roi_weights, self.p.score_thresh, self.p.nms_test, self.p.dets, pl_thresh, ws,
                       d.boxes, d.scores, d.classes, d.count, pl_boxes, pl_cls, pl_scores, pl_count, self.err)
        c.det = d
        c.pseudo = {"boxes": pl_boxes, "classes": pl_cls, "count": pl_count, "scores": pl_scores}
        c.props, c.prop_count, c.pred_all = props, pcount, pred
        return c

    # ------------------------------------------------------------------ backward
    def backward(self, c: Ctx, scales: Dict[str, float]):
        """Accumulate d(sum_k scales[k] * loss_k)/d(params) into weights.grad: loss gradients w.r.t. the
        head outputs first, then heads -> FPN -> res5..res3.  A missing / zero scale contributes
        nothing (the reference's `v * 0`)."""
        W = self.wts
        T = self.dtype
        dev = self.device
        sc = lambda k: float(scales.get(k, 0.0))
        scratch = torch.zeros(2, dtype=torch.float32, device=dev)
        c.ghead = [torch.zeros_like(h) for h in c.head]
        c.gpred = torch.zeros((max(c.R, 1), self.Cp), dtype=torch.float32, device=dev)
        gt = c.gt
        ops.rpn_loss(c.geom, c.head, c.ghead, c.anchors, c.rpn_labels, c.rpn_matched, gt["boxes"], gt["count"], GMAX, c.N,
                     1.0 / (self.p.rpn_batch * c.N), sc("loss_rpn_cls"), sc("loss_rpn_loc"), scratch)
        ops.box_loss(c.pred, self.Cp, self.K, c.R, c.rois, c.r_cls, c.r_gt, self.p.roi_weights, sc("loss_cls"), sc("loss_box_reg"), c.gpred, scratch)
        if c.distill is not None:
            d = c.distill
            # each kernel handles a loss pair with ONE scale: split the call when the two scales differ
            def rpn_d(do_obj, do_reg, s_):
                ops.rpn_distill_loss(c.geom, c.head, d["t_head"], c.ghead, d["labels"], c.N, d["obj_T"], d["n_valid"], d["n_fg"],
                                     do_obj, do_reg, s_, scratch)

            def roi_d(do_cls, do_reg, s_):
                ops.roih_distill_loss(c.pred, d["t_pred"], self.Cp, self.K, c.R, d["cls_T"], d["kl"], do_cls, do_reg, s_, c.gpred, scratch)
            if sc("loss_obj_bce") == sc("loss_rpn_l1"):
                rpn_d(d["do_obj"], d["do_rpn_reg"], sc("loss_obj_bce"))
            else:
                rpn_d(d["do_obj"], False, sc("loss_obj_bce"))
                rpn_d(False, d["do_rpn_reg"], sc("loss_rpn_l1"))
            if sc("loss_cls_ce") == sc("loss_roih_l1"):
                roi_d(d["do_cls"], d["do_roih_reg"], sc("loss_cls_ce"))
            else:
                roi_d(d["do_cls"], False, sc("loss_cls_ce"))
                roi_d(False, d["do_roih_reg"], sc("loss_roih_l1"))
        label = 1.0 if c.labeled else 0.0
        if "img" in c.align:
            acts, pooled, logit = c.align["img"]
            glog = torch.empty(logit.shape, dtype=T, device=dev)
            ops.domain_bce(logit, logit.shape[-1], c.N, label, c.da_weights[0], sc("loss_da_img"), glog, scratch)
            c.align["img"] = (acts, pooled, glog)
        if "ins" in c.align:
            acts, logit = c.align["ins"]
            glog = torch.empty(logit.shape, dtype=T, device=dev)
            ops.domain_bce(logit, logit.shape[-1], c.R, label, c.da_weights[1], sc("loss_da_ins"), glog, scratch)
            c.align["ins"] = (acts, glog)
        al = dict(n0=0, n1=c.N, r0=0, r1=c.R, **c.align)
        self._backward_trunk(c, [al] if c.align else [])

    def _backward_trunk(self, c: Ctx, align_list: List[dict]):
        """heads -> FPN -> res5..res3 given d(loss)/d(head outputs) in c.ghead / c.gpred (fp32)."""
        W = self.wts
        T = self.dtype
        dev = self.device
        # The stride-2 stages scatter their input gradient into every other pixel of a zero map (res5 -> c4, res4 -> c3): clear
        # those maps NOW on the side stream, beside the box head's backward, instead of in the middle of the data-gradient chain
        gx_pre, gx_ev = {}, None
        side0 = self._wgrad_stream()
        if side0 is not None and "cs" in c:
            side0.wait_stream(torch.cuda.current_stream())     # (fork: inside a graph capture the side stream must descend from the captured one)
            with torch.cuda.stream(side0):
                for si_, ci_ in ((3, 2), (2, 1)):
                    gx_pre[si_] = torch.zeros_like(c.cs[ci_])
                gx_ev = torch.cuda.Event()
                gx_ev.record(side0)
        # The sparse RPN-head backward up to its scatter needs the head gradients only: on an auxiliary stream, beside the box head's
        # backward and ROIAlign's (it was ~0.12 ms of small launches between ROIAlign's backward and the FPN's)
        sp_pre, aux = None, None
        if self.sparse_rpn_backward:
            aux = self._aux_stream()
            if aux is not None:
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    sp_pre = self._rpn_sparse_prepare(c)
        # ---- box head (+ instance-level discriminator behind the gradient-reversal layer)
        # ROIAlign's share of d(loss)/d(P_l).  With the sparse RPN-head backward these maps ARE the FPN backward's input, so in bf16
        # mode they are written in bf16 directly (no fp32 map + cast pass): the ROI part rounded once, the RPN head's few thousand
        # active pixels added on top in bf16 -- the two bf16 gradients the reference's autocast sums (aldi/trainer.py:79)
        g_dt = T if (self.sparse_rpn_backward and T == torch.bfloat16) else torch.float32
        gP_roi = [(torch.empty if c.R > 0 else torch.zeros)(f.shape, dtype=g_dt, device=dev) for f in c.P[:4]]
        if c.R > 0:
            g_extra = None
            for al in align_list:
                if "ins" not in al:
                    continue
                acts, g_ = al["ins"]
                r0, r1 = al["r0"], al["r1"]
                if g_extra is None:
                    g_extra = torch.zeros((c.R, 1, 1, FC_DIM), dtype=T, device=dev)
                L_ = self.ins_da_layers
                for i in range(len(L_) - 1, -1, -1):             # last Linear first; the gradient-reversal layer sits below layer 0
                    self._wgrad(L_[i], acts[i], g_)
                    if i == 0:
                        ops.conv2d(g_, W.wt(L_[0], negate=True), out=g_extra[r0:r1])
                    else:
                        g_ = ops.conv2d(g_, W.wt(L_[i]), mask=acts[i])      # acts[i] is the ReLU output of layer i-1
            lo = c.get("gpred_lo")
            gpred = (lo if lo is not None else ops.cast_from_f32(c.gpred[:c.R], T)).view(c.R, 1, 1, self.Cp)
            self._wgrad("box_pred", c.fc2, gpred)
            g_fc2 = ops.conv2d(gpred, W.wt("box_pred"), res=g_extra, res_mode=1 if g_extra is not None else 0, **self._relu_mask(c, c.fc2))
            self._wgrad("roi_heads.box_head.fc2", c.fc1, g_fc2)
            g_fc1 = ops.conv2d(g_fc2, W.wt("roi_heads.box_head.fc2"), mask=c.fc1)
            x = c.pooled.view(c.R, 1, 1, POOL * POOL * FPN_C)
            self._wgrad("roi_heads.box_head.fc1", x, g_fc1)
            g_pooled = ops.conv2d(g_fc1, W.wt("roi_heads.box_head.fc1")).view(c.R, POOL, POOL, FPN_C)
            ops.roialign_backward(self.roi_feats(c, gP_roi), c.rois, c.R, POOL, g_pooled, c.N, rois_sorted=True, grad_dtype=g_dt)
        self._grads_final(["box_pred", "roi_heads.box_head.fc2", "roi_heads.box_head.fc1"])
        # ---- RPN head (shared weights over 5 levels)
        if self.sparse_rpn_backward:
            if sp_pre is not None:
                main_ = torch.cuda.current_stream()
                main_.wait_stream(aux)
                for t_ in sp_pre.values():
                    if isinstance(t_, torch.Tensor):
                        t_.record_stream(main_)
                gP = self._rpn_sparse_finish(c, gP_roi, sp_pre)
            else:
                gP = self._rpn_head_backward_sparse(c, gP_roi)
        else:
            gP = []
            for l in range(5):
                gh = ops.cast_from_f32(c.ghead[l], T)
                self._wgrad("rpn_head_out", c.rpn_t[l], gh)
                g_t = ops.conv2d(gh, W.wt("rpn_head_out"), mask=c.rpn_t[l])
                self._wgrad("proposal_generator.rpn_head.conv", c.P[l], g_t)
                gP.append(ops.conv2d(g_t, W.wt("proposal_generator.rpn_head.conv"), pad=1))
            for l in range(4):
                ops.add_f32(gP[l], gP_roi[l], gP[l])
        # ---- image-level discriminator behind the gradient-reversal layer (on any of p2..p6)
        for al in align_list:
            if "img" not in al:
                continue
            acts, pooled, glog = al["img"]
            n0, n1 = al["n0"], al["n1"]
            L_ = self.img_da_layers
            lvl = self.img_da_level
            self._wgrad(L_[-1], pooled, glog)
            g_pooled = ops.conv2d(glog, W.wt(L_[-1], negate=len(L_) == 1))
            if len(L_) == 1:                                     # no hidden layer: the reversed gradient of the average pool itself
                hw = acts[0].shape[1] * acts[0].shape[2]
                gP[lvl][n0:n1] += (g_pooled.float() / hw).to(T)
                continue
            g_ = ops.avgpool_bwd(g_pooled, acts[-1])             # through the average pool and the last hidden ReLU
            for i in range(len(L_) - 2, -1, -1):
                self._wgrad(L_[i], acts[i], g_)
                if i == 0:
                    ops.conv2d(g_, W.wt(L_[0], negate=True), pad=2, res=gP[lvl][n0:n1], res_mode=1, out=gP[lvl][n0:n1])
                else:
                    g_ = ops.conv2d(g_, W.wt(L_[i]), pad=2, mask=acts[i])
        ops.subsample2_bwd(gP[4], gP[3])                               # p6 = p5[:, ::2, ::2]
        # ---- FPN
        gprev = {}
        for i, lvl in enumerate((2, 3, 4, 5)):
            self._wgrad(f"backbone.fpn_output{lvl}", c.prev[lvl], gP[i])
        if self.level_groups:
            gs = ops.conv2d_group([(gP[i], W.wt(f"backbone.fpn_output{lvl}"), dict(pad=1)) for i, lvl in enumerate((2, 3, 4, 5))])
            gprev.update(zip((2, 3, 4, 5), gs))
        else:
            for i, lvl in enumerate((2, 3, 4, 5)):
                gprev[lvl] = ops.conv2d(gP[i], W.wt(f"backbone.fpn_output{lvl}"), pad=1)
        sh = [tuple(gprev[lvl].shape) for lvl in (2, 3, 4, 5)]
        if (all(sh[i][1] == 2 * sh[i + 1][1] and sh[i][2] == 2 * sh[i + 1][2] for i in range(3)) and all(gprev[lvl].is_contiguous() for lvl in (2, 3, 4, 5))
                and os.environ.get("ALDI_UPSAMPLE_CHAIN", "1") == "1"):
            ops.upsample2_bwd_chain(gprev[2], gprev[3], gprev[4], gprev[5])      # one launch instead of three dependent ones (same bits)
        else:
            for lvl in (3, 4, 5):
                ops.upsample2_bwd(gprev[lvl - 1], gprev[lvl], accumulate=True)
        for lvl in (2, 3, 4, 5):
            self._wgrad(f"backbone.fpn_lateral{lvl}", c.cs[lvl - 2], gprev[lvl])
        self._grads_final(["rpn_head_out", "proposal_generator.rpn_head.conv"] + [f"backbone.fpn_output{l}" for l in (2, 3, 4, 5)] +
                          [f"backbone.fpn_lateral{l}" for l in (2, 3, 4, 5)])
        # ---- res5 .. res3 (stem + res2 frozen: FREEZE_AT=2)
        g = ops.conv2d(gprev[5], W.wt("backbone.fpn_lateral5"), **self._relu_mask(c, c.cs[3]))
        blocks = c.blocks
        bi = len(blocks) - 1
        for si in (3, 2, 1):
            stage_names: List[str] = []
            for b in range(STAGE_BLOCKS[si] - 1, -1, -1):
                p, xin, h1, h2, out, first = blocks[bi]
                bi -= 1
                stage_names += [p + "conv3", p + "conv2", p + "conv1"] + ([p + "shortcut"] if first else [])
                self._wgrad(p + "conv3", h2, g)
                g2 = ops.conv2d(g, W.wt(p + "conv3"), **self._relu_mask(c, h2))
                self._wgrad(p + "conv2", h1, g2)
                g1 = ops.conv2d(g2, W.wt(p + "conv2"), pad=1, **self._relu_mask(c, h1))
                self._wgrad(p + "conv1", xin, g1)
                if si == 1 and not first and self.wgrad_tail_split > 0 and (STAGE_BLOCKS[si] - b) % self.wgrad_tail_split == 0:
                    # res3 is the LAST stage of the backward: launched as one group behind its last data gradient, its weight gradients
                    # (0.28 ms) run alone at the end of the step -- nothing is left to run beside them.  Launching the blocks done so
                    # far puts that part under the remaining blocks' data gradients (profiles/r05_timeline_spin.txt)
                    self._flush_wgrads()
                if first:
                    self._wgrad(p + "shortcut", xin, g)
                    self._grads_final(stage_names)
                    if si == 1:
                        break                                           # stage input (res2 output) needs no gradient
                    stride = W.layout.t[p + "conv1"].stride
                    Hin, Win = xin.shape[1], xin.shape[2]
                    gx = gx_pre.pop(si, None)
                    if gx is None or gx.shape != xin.shape:
                        gx = torch.zeros_like(xin)
                    else:
                        torch.cuda.current_stream().wait_event(gx_ev)
                        gx.record_stream(torch.cuda.current_stream())
                    ops.conv2d(g, W.wt(p + "shortcut"), out=gx, out_scale=stride, out_hw=(Hin, Win))
                    ops.conv2d(g1, W.wt(p + "conv1"), out=gx, out_scale=stride, out_hw=(Hin, Win), res=gx, res_mode=1)
                    lvl = si + 1                                        # this stage's input is C_{lvl}
                    g = ops.conv2d(gprev[lvl], W.wt(f"backbone.fpn_lateral{lvl}"), res=gx, res_mode=1, **self._relu_mask(c, xin))
                else:
                    g = ops.conv2d(g1, W.wt(p + "conv1"), res=g, res_mode=1, **self._relu_mask(c, xin))
        self._join_wgrads()

    def trunk_backward(self, c: Ctx, gC: Dict[int, Optional[torch.Tensor]]):
        """backward of the bare trunk (trunk_steps(..., fpn=False), fp32): gC[lvl] = d(loss)/d(C_lvl) for lvl in 3, 4, 5 (the stage outputs,
        post-ReLU; None = no gradient); weight gradients of res3..res5 accumulate as in `backward`, stem + res2 are frozen (FREEZE_AT = 2)"""
        W = self.wts
        assert self.dtype == torch.float32, "the bare-trunk backward runs in the fp32 mode (the DETR path's AMP is off)"
        def masked(a, b, act):                     # (a + b) where act > 0; a, b nullable (not both)
            if a is None and b is None:
                return None
            x, y = (a, b) if b is not None else (None, a)
            return ops.add_f32(x, y, torch.empty_like(act), relu_src=act)
        blocks = c.blocks
        bi = len(blocks) - 1
        g = masked(gC.get(5), None, c.cs[3])
        for si in (3, 2, 1):
            stage_names: List[str] = []
            for b in range(STAGE_BLOCKS[si] - 1, -1, -1):
                p, xin, h1, h2, out, first = blocks[bi]
                bi -= 1
                stage_names += [p + "conv3", p + "conv2", p + "conv1"] + ([p + "shortcut"] if first else [])
                self._wgrad(p + "conv3", h2, g)
                g2 = ops.conv2d(g, W.wt(p + "conv3"), mask=h2)
                self._wgrad(p + "conv2", h1, g2)
                g1 = ops.conv2d(g2, W.wt(p + "conv2"), pad=1, mask=h1)
                self._wgrad(p + "conv1", xin, g1)
                if first:
                    self._wgrad(p + "shortcut", xin, g)
                    self._grads_final(stage_names)
                    if si == 1:
                        break                                           # stage input (res2 output) needs no gradient
                    stride = W.layout.t[p + "conv1"].stride
                    Hin, Win = xin.shape[1], xin.shape[2]
                    gx = torch.zeros_like(xin)
                    ops.conv2d(g, W.wt(p + "shortcut"), out=gx, out_scale=stride, out_hw=(Hin, Win))
                    ops.conv2d(g1, W.wt(p + "conv1"), out=gx, out_scale=stride, out_hw=(Hin, Win), res=gx, res_mode=1)
                    g = masked(gC.get(si + 1), gx, xin)                # this stage's input is C_{si+1}: its own gradient joins here
                else:
                    g = ops.conv2d(g1, W.wt(p + "conv1"), res=g, res_mode=1, mask=xin)
        self._join_wgrads()

    @staticmethod
    def _relu_mask(c: Ctx, act: torch.Tensor) -> dict:
        """conv2d kwargs that mask a data gradient by (act > 0): the bit mask the forward wrote beside `act` when there is one"""
        bits = (c.get("out_bits") or {}).get(act.data_ptr())
        return dict(mask_bits=bits) if bits is not None else dict(mask=act)

    def _rpn_sparse_prepare(self, c: Ctx):
        """RPN head backward over the ACTIVE pixels only (csrc/rpn_sparse.hip), everything up to the scatter: d(loss)/d(head
        outputs) is non-zero at the sampled anchors' pixels (the RPN losses' sample + the positions the distillation losses' fresh
        sample selects), i.e. at <= 4 * 256 * N of the 358 k pixel positions.  Needs only c.ghead, so `_backward_trunk` runs it
        on an auxiliary stream beside the box head's backward."""
        W, T, dev = self.wts, self.dtype, self.device
        N, Cf, Ch = c.N, FPN_C, self.Ch
        # bound on the active pixels of one image: RPN_BATCH sampled anchors of the RPN losses + RPN_BATCH mask positions of the
        # distillation objectness loss + 4 per foreground position (<= RPN_BATCH * POSITIVE_FRACTION) of the distillation L1: the reference's
        # `repeat_interleave(fg, 4)` mask lands on four consecutive positions of the RAW (N, 4A, H, W) layout, i.e. on four
        # different pixels
        cap = (2 * self.p.rpn_batch + 4 * int(self.p.rpn_batch * self.p.rpn_pos_frac)) * N
        idx = torch.empty(cap, dtype=torch.int32, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(ops.rpn_active_pixels_workspace(c.geom, N), dtype=torch.uint8, device=dev)
        ops.rpn_active_pixels(c.geom, c.ghead, N, cap, idx, count, ws, self.err)
        G = torch.empty((cap, 1, 1, Ch), dtype=T, device=dev)
        Tm = torch.empty((cap, 1, 1, Cf), dtype=T, device=dev)
        X9 = torch.empty((cap, 1, 1, 9 * Cf), dtype=T, device=dev)
        ops.rpn_sparse_gather(c.geom, c.ghead, c.rpn_t, c.P, N, Cf, cap, idx, count, G, Tm, X9)
        g_t = ops.conv2d(G, W.wt("rpn_head_out"), mask=Tm)
        Y = ops.conv2d(g_t, W.wt_flat("proposal_generator.rpn_head.conv"))           # [cap][9][Cf]: contributions to the 3x3 neighbourhood
        return dict(cap=cap, idx=idx, count=count, G=G, Tm=Tm, X9=X9, g_t=g_t, Y=Y)

    def _rpn_sparse_finish(self, c: Ctx, gP_roi: List[torch.Tensor], sp: dict) -> List[torch.Tensor]:
        """queue the two weight gradients and scatter the rows into d(loss)/d(P_l) (ROIAlign's contribution gP_roi included: fp32
        maps are summed in fp32 and rounded once, bf16 maps take the rows by packed bf16 atomics)"""
        T, dev = self.dtype, self.device
        self._wgrad("rpn_head_out", sp["Tm"], sp["G"], temp_x=True)
        self._wgrad("proposal_generator.rpn_head.conv", sp["X9"], sp["g_t"], flat=True, temp_x=True)
        g32 = list(gP_roi) + [torch.zeros(c.P[4].shape, dtype=gP_roi[0].dtype, device=dev)]
        ops.rpn_sparse_scatter(c.geom, g32, sp["Y"], c.N, FPN_C, sp["cap"], sp["idx"], sp["count"])
        return g32 if g32[0].dtype == T else [ops.cast_from_f32(g, T) for g in g32]

    def _rpn_head_backward_sparse(self, c: Ctx, gP_roi: List[torch.Tensor]) -> List[torch.Tensor]:
        return self._rpn_sparse_finish(c, gP_roi, self._rpn_sparse_prepare(c))

    def _aux_stream(self):
        if not hasattr(self, "_aux_side"):
            self._aux_side = torch.cuda.Stream(device=self.device) if os.environ.get("ALDI_AUX_STREAM", "1") == "1" and torch.device(self.device).type == "cuda" else None
        return self._aux_side

    def _grads_final(self, names: List[str]):
        """tell the gradient exchange (if one is attached: data-parallel fused step) that these layers' gradients are
        complete for this step -- every kernel writing them has been enqueued."""
        self._flush_wgrads()
        cb = getattr(self, "grad_ready", None)
        if cb is None:
            return
        # Producer events instead of a join: the gradient exchange waits for the weight-gradient stream and for this point of
        # the main stream, the main stream itself keeps running beside the side stream (a join at each of the six reports
        # serialised the dgrad chain behind the wgrad kernels, i.e. removed the overlap the 1-GPU step relies on).
        evs = []
        side = getattr(self, "_wg_side", None)
        if side is not None and self._wgrad_pending:
            ev = torch.cuda.Event()
            ev.record(side)
            evs.append(ev)
        ev = torch.cuda.Event()
        ev.record()
        evs.append(ev)
        if len(inspect.signature(cb).parameters) >= 2:
            cb(self.wts.layout.ranges(names), evs)
        else:                                               # a plain callback(ranges): stream-ordered after everything
            self._join_wgrads()
            cb(self.wts.layout.ranges(names))

    def _wgrad(self, name: str, x: torch.Tensor, g: torch.Tensor, flat: bool = False, temp_x: bool = False):
        """weight (+ bias) gradient of one layer (`flat`: x holds im2col rows [S][KH*KW*Cin], the gradient is a plain GEMM).
        Nothing reads these results before the optimizer, so (a) in bf16 mode they are only QUEUED here and launched per layer
        group -- a stage's small GEMMs in one grouped launch, csrc/wgrad.hip -- when `_grads_final` / `_join_wgrads` flushes,
        and (b) they run on a second HIP stream beside the data-gradient chain."""
        W = self.wts
        p = W.layout.t[name]
        geo = dict(KH=1, KW=1, stride=1, pad=0) if flat else dict(KH=p.kk, KW=p.kk, stride=p.stride, pad=p.pad)
        geo["scale"] = W.scale(name)
        if (not flat and p.kk == 1 and p.stride == 2 and p.pad == 0 and self.dtype == torch.bfloat16 and getattr(self, "group_wgrad", False)
                and os.environ.get("ALDI_WGRAD_SUBSAMPLE", "1") == "1"):
            # a stride-2 1x1 conv (first block of res3..res5: conv1 and the shortcut read the same input) sees every other pixel:
            # gather those ONCE into a dense map and its weight gradient is a plain GEMM that joins the stage's grouped launch,
            # instead of two launches of the generic gather kernel per stage (6 x ~60 us)
            cache = self.__dict__.setdefault("_sub_cache", {})
            hit = cache.get(id(x))
            if hit is None or hit[0] is not x:
                side_ = self._wgrad_stream()
                if side_ is not None and os.environ.get("ALDI_WGRAD_SUBSAMPLE_SIDE", "1") == "1":
                    # only the weight gradients read the gathered map: on THEIR stream, not as one more dependent launch of the
                    # data-gradient chain (x is a saved activation: complete long before this point of the main stream)
                    ev_ = torch.cuda.Event()
                    ev_.record()
                    side_.wait_event(ev_)
                    with torch.cuda.stream(side_):
                        hit = (x, ops.subsample2(x))
                else:
                    hit = (x, ops.subsample2(x))
                cache.clear()
                cache[id(x)] = hit
            x, temp_x = hit[1], True
            geo.update(stride=1)
        if getattr(self, "group_wgrad", False) and self.dtype == torch.bfloat16:
            fused_bias = p.bias and os.environ.get("ALDI_WGRAD_BIAS_FUSED", "1") == "1"    # (0: the separate column-sum launches, for A/B runs)
            if fused_bias:
                geo["db"] = W.gb(name)
            self._wg_queue.append((x, g, W.gw(name), geo, W.gb(name) if p.bias and not fused_bias else None, temp_x))
            return
        side = self._wgrad_stream()
        if p.bias:
            geo["db"] = W.gb(name)                   # the bias gradient rides in the same launch (ones column; csrc/wgrad.hip)
        if side is None:
            ops.conv_wgrad(x, g, W.gw(name), **geo)
            return
        ev = torch.cuda.Event()
        ev.record()                                  # g (and x) are complete once the main stream gets here
        side.wait_event(ev)
        g.record_stream(side)                        # g is a temporary of the main stream's allocator
        if temp_x:
            x.record_stream(side)                    # x is a temporary too (gathered rows), not a saved activation
        with torch.cuda.stream(side):
            ops.conv_wgrad(x, g, W.gw(name), **geo)
        self._wgrad_pending = True

    def _flush_wgrads(self):
        """launch the queued weight gradients: one grouped launch (+ the bias sums) on the weight-gradient stream"""
        q = getattr(self, "_wg_queue", None)
        if not q:
            return
        self._wg_queue = []
        side = self._wgrad_stream()

        def launch():
            ops.conv_wgrad_group([(x, g, dw, geo) for x, g, dw, geo, _, _ in q])
            for _, g, _, _, gb, _ in q:
                if gb is not None:
                    ops.bias_grad(g, gb)
        if side is None:
            launch()
            return
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        for x, g, _, _, _, temp_x in q:
            g.record_stream(side)
            if temp_x:
                x.record_stream(side)
        with torch.cuda.stream(side):
            launch()
        self._wgrad_pending = True

    def _wgrad_stream(self):
        if not hasattr(self, "_wg_side"):
            self._wg_side = (torch.cuda.Stream(device=self.device, priority=int(os.environ.get("ALDI_WGRAD_PRIO", "0")))
                             if os.environ.get("ALDI_WGRAD_STREAM", "1") == "1" else None)
            self._wgrad_pending = False
        return self._wg_side

    def _join_wgrads(self):
        """main stream waits for every weight-gradient kernel issued so far"""
        self._flush_wgrads()
        if getattr(self, "_wg_side", None) is not None and self._wgrad_pending:
            torch.cuda.current_stream().wait_stream(self._wg_side)
            self._wgrad_pending = False
