"""Strong augmentation on the device (reference: aldi/aug.py:39-60 `build_strong_augmentation`, :80-171 transforms).

The reference derives the strong view on the CPU (numpy + scipy's `gaussian_filter`, tens of ms per Cityscapes frame).
Here the weak view lives in HBM as an HWC uint8 tensor and every transform is a HIP kernel (`csrc/aug.hip`); only the
RANDOM DRAWS stay on the host, consumed from the same generators, in the same order, as the reference:

* `np.random.uniform` -- the `RandomApply` gates and the colour weights (detectron2 `Augmentation._rand_range`,
  `RandomContrast/Brightness/Saturation.get_transform`),
* python `random`     -- the blur sigma (drawn inside `RandomBlurTransform.apply_image`, aldi/aug.py:86) and the erase
  geometry (:116-123),
* `np.random.rand`    -- erase fills (:125) and the MIC block mask (:162).

Same class names / constructor arguments as the reference; `apply_image` takes and returns a CUDA uint8 HWC tensor.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops


def _p(t):
    return t.data_ptr() if t is not None else None


def _check(img: torch.Tensor):
    if not (img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_contiguous()):
        raise ValueError("device augmentation expects a contiguous CUDA uint8 HWC image with 3 channels")


# ------------------------------------------------------------------------------------------------ colour (detectron2 names)
class _Blend:
    mode = -1

    def __init__(self, intensity_min: float, intensity_max: float):
        self.intensity_min, self.intensity_max = intensity_min, intensity_max

    def draw(self) -> float:
        return np.random.uniform(self.intensity_min, self.intensity_max)

    def apply_image(self, img: torch.Tensor, w: Optional[float] = None) -> torch.Tensor:
        _check(img)
        if w is None:
            w = self.draw()
        H, W, _ = img.shape
        out = img.clone()
        s = None
        if self.mode == 0:
            s = torch.empty(1, dtype=torch.int64, device=img.device)
            L.call("aldi_aug_sum_u8", _p(out), out.numel(), _p(s), ops.stream_ptr())
        L.call("aldi_aug_blend", _p(out), H, W, self.mode, float(w), _p(s), ops.stream_ptr())
        return out


class RandomContrast(_Blend):
    mode = 0


class RandomBrightness(_Blend):
    mode = 1


class RandomSaturation(_Blend):
    mode = 2


# ------------------------------------------------------------------------------------------------ ALDI-owned transforms
def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """the taps scipy's gaussian_filter builds (float64): radius int(truncate * sigma + 0.5)"""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


class RandomBlurTransform:
    def __init__(self, sigma: Sequence[float]):
        self.sigma = sigma

    def apply_image(self, img: torch.Tensor, sigma: Optional[float] = None) -> torch.Tensor:
        _check(img)
        if sigma is None:
            sigma = random.uniform(self.sigma[0], self.sigma[1])
        H, W, _ = img.shape
        w = gaussian_weights(sigma)
        wd = torch.from_numpy(w).to(img.device)
        tmp0 = torch.empty(img.numel(), dtype=torch.float32, device=img.device)
        tmp1 = torch.empty_like(tmp0)
        out = torch.empty_like(img)
        L.call("aldi_aug_blur", _p(img), _p(out), _p(tmp0), _p(tmp1), H, W, _p(wd), (len(w) - 1) // 2, ops.stream_ptr())
        return out


class RandomEraseTransform:
    """scale=(sl, sh) of the image area, aspect ratio in (r1, r2), value="random" (uniform noise), as the reference."""
    def __init__(self, sl=0.02, sh=0.4, r1=0.3, r2=3.3, value="random"):
        self.sl, self.sh, self.r1, self.r2, self.value = sl, sh, r1, r2, value

    def draw(self, imgh: int, imgw: int) -> Optional[Tuple[int, int, int, int]]:
        for _ in range(100):
            area = imgw * imgh
            target_area = random.uniform(self.sl, self.sh) * area
            aspect_ratio = random.uniform(self.r1, self.r2)
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w > 1 and h > 1 and w < imgw and h < imgh:
                h0 = random.randint(0, imgh - h - 1)
                w0 = random.randint(0, imgw - w - 1)
                return h0, w0, h, w
        return None

    def apply_image(self, img: torch.Tensor, rect=None, fill: Optional[np.ndarray] = None) -> torch.Tensor:
        _check(img)
        H, W, C = img.shape
        if rect is None:
            rect = self.draw(H, W)
        if rect is None:
            return img
        h0, w0, h, w = rect
        if fill is None:
            fill = np.random.rand(h, w, C) if self.value == "random" else np.full((h, w, C), float(self.value))
        f32 = torch.from_numpy(np.ascontiguousarray(fill, dtype=np.float32)).to(img.device)   # the cast the reference's assignment does
        out = img.clone()
        L.call("aldi_aug_erase", _p(out), H, W, h0, w0, h, w, _p(f32), ops.stream_ptr())
        return out


class MICTransform:
    def __init__(self, ratio: float, block_size: int):
        self.ratio, self.block_size = ratio, block_size

    def draw(self, H: int, W: int) -> np.ndarray:
        mh, mw = round(H / self.block_size), round(W / self.block_size)
        return np.random.rand(mh, mw) > self.ratio

    def apply_image(self, img: torch.Tensor, mask: Optional[np.ndarray] = None) -> torch.Tensor:
        _check(img)
        H, W, _ = img.shape
        if mask is None:
            mask = self.draw(H, W)
        m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(img.device)
        out = img.clone()
        L.call("aldi_aug_mic", _p(out), H, W, _p(m), int(m.shape[0]), int(m.shape[1]), ops.stream_ptr())
        return out


# ------------------------------------------------------------------------------------------------ the chain
class RandomApply:
    """detectron2 RandomApply: one np.random.uniform draw decides; `aug` is a transform or a list applied in order."""
    def __init__(self, aug, prob: float = 0.5):
        self.aug, self.prob = aug, prob

    def apply_image(self, img: torch.Tensor) -> torch.Tensor:
        if np.random.uniform(0, 1.0) < self.prob:
            for a in (self.aug if isinstance(self.aug, (list, tuple)) else [self.aug]):
                img = a.apply_image(img)
        return img


def build_strong_augmentation(include_erasing: bool = True) -> List[RandomApply]:
    """aldi/aug.py:39-60, same probabilities and ranges"""
    augs = [
        RandomApply([RandomContrast(0.6, 1.4), RandomBrightness(0.6, 1.4), RandomSaturation(0.6, 1.4)], prob=0.8),
        RandomApply(RandomSaturation(0, 0), prob=0.2),                 # random grayscale
        RandomApply(RandomBlurTransform((0.1, 2.0)), prob=0.5),
    ]
    if include_erasing:
        augs += [
            RandomApply(RandomEraseTransform(sl=0.05, sh=0.2, r1=0.3, r2=3.3, value="random"), prob=0.7),
            RandomApply(RandomEraseTransform(sl=0.02, sh=0.2, r1=0.1, r2=6, value="random"), prob=0.5),
            RandomApply(RandomEraseTransform(sl=0.02, sh=0.2, r1=0.05, r2=8, value="random"), prob=0.3),
        ]
    return augs


def get_strong_augs(cfg, labeled: bool) -> List[RandomApply]:
    """the strong part of `get_augs` (aldi/aug.py:26-35): erasing / MIC switches from cfg.AUG"""
    erasing = (labeled and cfg.AUG.LABELED_INCLUDE_RANDOM_ERASING) or (not labeled and cfg.AUG.UNLABELED_INCLUDE_RANDOM_ERASING)
    augs = build_strong_augmentation(include_erasing=erasing)
    if (labeled and cfg.AUG.LABELED_MIC_AUG) or (not labeled and cfg.AUG.UNLABELED_MIC_AUG):
        augs.append(RandomApply(MICTransform(cfg.AUG.MIC_RATIO, cfg.AUG.MIC_BLOCK_SIZE), prob=1.0))
    return augs


def strong_view(img_weak_hwc: torch.Tensor, augs: Sequence[RandomApply], chw: bool = True) -> torch.Tensor:
    """weak view (HWC uint8, device) -> strong view; `chw` returns the (3, H, W) layout dataset dicts carry"""
    img = img_weak_hwc
    for a in augs:
        img = a.apply_image(img)
    if not chw:
        return img
    H, W, _ = img.shape
    out = torch.empty((3, H, W), dtype=torch.uint8, device=img.device)
    L.call("aldi_aug_hwc_to_chw", _p(img), _p(out), H, W, ops.stream_ptr())
    return out
