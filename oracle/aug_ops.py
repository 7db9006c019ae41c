"""CPU restatement of the reference's STRONG AUGMENTATION chain (aldi/aug.py).

TEST INFRASTRUCTURE ONLY -- nothing in aldi_amd/ imports this file.

Two provenance classes:

* ALDI-owned transforms -- `RandomBlurTransform` (aldi/aug.py:80-91), `RandomEraseTransform` (:103-138),
  `MICTransform` (:149-171): **pinned** by golden g9 (tests/golden/g9_aug.npz), produced by running the reference's own
  classes (scipy's `gaussian_filter` is the real one; `cv2.resize(..., INTER_NEAREST)` is absent from the image and
  restated below -- that one call is unpinned).
* Detectron2 colour transforms reached through `build_strong_augmentation` (aldi/aug.py:39-60): `RandomContrast`,
  `RandomBrightness`, `RandomSaturation`, `RandomApply`, `AugmentationList`, `BlendTransform` -- detectron2 is absent
  (un-vendored, unpinned fork): restated from the published v0.6 behaviour, **parity unpinned**.

All arithmetic follows the numpy the reference would run under THIS image's numpy (2.x, NEP 50 promotion): a float64
scalar/array times a float32 image gives float64 (contrast, saturation), a python float times a float32 image stays
float32 (brightness, MIC, erase).
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------- detectron2 (unpinned)
def blend(img_u8: np.ndarray, src_image, src_weight: float, dst_weight: float) -> np.ndarray:
    """BlendTransform.apply_image for uint8 input: float32 copy, w_src*src + w_dst*img, clip, truncate to uint8."""
    img = img_u8.astype(np.float32)
    img = src_weight * src_image + dst_weight * img
    return np.clip(img, 0, 255).astype(np.uint8)


def contrast(img_u8: np.ndarray, w: float) -> np.ndarray:
    """RandomContrast.get_transform(image) with the drawn weight: blend with the image mean (np.float64 scalar)."""
    return blend(img_u8, img_u8.mean(), 1 - w, w)


def brightness(img_u8: np.ndarray, w: float) -> np.ndarray:
    return blend(img_u8, 0, 1 - w, w)


def saturation(img_u8: np.ndarray, w: float) -> np.ndarray:
    """RandomSaturation: blend with image.dot([0.299, 0.587, 0.114]) -- applied to the channels AS STORED (BGR here)."""
    gray = img_u8.dot([0.299, 0.587, 0.114])[:, :, np.newaxis]
    return blend(img_u8, gray, 1 - w, w)


# ------------------------------------------------------------------------------------------- ALDI-owned (pinned by g9)
def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d(order 0) as used by gaussian_filter: radius int(truncate*sigma+0.5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """scipy 'reflect' (d c b a | a b c d | d c b a)"""
    p = np.mod(i, 2 * n)
    return np.where(p >= n, 2 * n - 1 - p, p)


def correlate1d_symmetric(a: np.ndarray, w: np.ndarray, axis: int) -> np.ndarray:
    """NI_Correlate1D's symmetric branch: double line buffer, tmp = x[l]*w0; for j=-r..-1: tmp += (x[l+j] + x[l-j])*w[j];
    result cast to the array dtype (float32)."""
    r = (len(w) - 1) // 2
    a = np.moveaxis(a, axis, -1)
    n = a.shape[-1]
    x = a.astype(np.float64)
    idx = np.arange(n)
    tmp = x * w[r]
    for j in range(-r, 0):
        lo = x[..., _reflect_index(idx + j, n)]
        hi = x[..., _reflect_index(idx - j, n)]
        tmp = tmp + (lo + hi) * w[j + r]
    return np.moveaxis(tmp.astype(np.float32), -1, axis)


def gaussian_blur(img_u8: np.ndarray, sigma: float) -> np.ndarray:
    """RandomBlurTransform.apply_image (aldi/aug.py:85-91): scipy gaussian_filter over ALL THREE axes of the HWC float32 image
    (yes, the channel axis too), clip, truncate."""
    a = img_u8.astype(np.float32)
    w = gaussian_weights(sigma)
    for axis in range(3):
        a = correlate1d_symmetric(a, w, axis)
    return np.clip(a, 0, 255).astype(np.uint8)


def erase_params(imgh: int, imgw: int, sl: float, sh: float, r1: float, r2: float) -> Optional[Tuple[int, int, int, int]]:
    """the rejection loop of RandomEraseTransform.apply_image (:113-124): python `random` draws, up to 100 attempts"""
    for _ in range(100):
        area = imgw * imgh
        target_area = random.uniform(sl, sh) * area
        aspect_ratio = random.uniform(r1, r2)
        h = int(round(math.sqrt(target_area * aspect_ratio)))
        w = int(round(math.sqrt(target_area / aspect_ratio)))
        if w > 1 and h > 1 and w < imgw and h < imgh:
            h0 = random.randint(0, imgh - h - 1)
            w0 = random.randint(0, imgw - w - 1)
            return h0, w0, h, w
    return None


def erase(img_u8: np.ndarray, rect: Tuple[int, int, int, int], fill: np.ndarray) -> np.ndarray:
    """value="random": img[rect] = fill (np.random.rand(h, w, c) cast to float32 on assignment), *= 255, clip, truncate"""
    h0, w0, h, w = rect
    img = img_u8.astype(np.float32)
    img[h0:h0 + h, w0:w0 + w, :] = fill
    img[h0:h0 + h, w0:w0 + w, :] *= 255
    return np.clip(img, 0, 255).astype(np.uint8)


def resize_nearest_u8(m: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(m, (W, H), interpolation=INTER_NEAREST) (UNPINNED: cv2 absent): src = min(floor(dst * src/dst), src-1)"""
    mh, mw = m.shape
    ys = np.minimum(np.floor(np.arange(H) * (mh / H)).astype(np.int64), mh - 1)
    xs = np.minimum(np.floor(np.arange(W) * (mw / W)).astype(np.int64), mw - 1)
    return m[ys][:, xs]


def mic_mask(img_u8: np.ndarray, mask_small: np.ndarray) -> np.ndarray:
    """MICTransform.apply_image (:154-171) given the drawn block mask (np.random.rand(mh, mw) > ratio)"""
    H, W, C = img_u8.shape
    img = img_u8.astype(np.float32)
    big = resize_nearest_u8(np.asarray(mask_small, dtype="uint8"), W, H)
    masked = img * np.repeat(big[..., np.newaxis], C, axis=-1)
    return np.clip(masked, 0, 255).astype(np.uint8)


def mic_grid(H: int, W: int, block_size: int) -> Tuple[int, int]:
    return round(H / block_size), round(W / block_size)


# ------------------------------------------------------------------------------------------- the chain + its RNG streams
def draw_strong_params(H: int, W: int, include_erasing: bool, mic: Optional[Tuple[float, int]] = None) -> List[tuple]:
    """Consume the global numpy / python RNG streams exactly as `build_strong_augmentation` (+ MIC) would for one image
    (aldi/aug.py:39-60, 31-33): RandomApply gates and colour weights come from np.random.uniform, blur sigma and the erase
    geometry from python `random`, erase fills and the MIC mask from np.random.rand.  Returns a list of ops."""
    ops: List[tuple] = []
    if np.random.uniform(0, 1.0) < 0.8:                     # RandomApply(AugmentationList([...]), prob=0.8)
        ops.append(("contrast", np.random.uniform(0.6, 1.4)))
        ops.append(("brightness", np.random.uniform(0.6, 1.4)))
        ops.append(("saturation", np.random.uniform(0.6, 1.4)))
    if np.random.uniform(0, 1.0) < 0.2:                     # random grayscale = RandomSaturation(0, 0)
        ops.append(("saturation", np.random.uniform(0, 0)))
    if np.random.uniform(0, 1.0) < 0.5:
        ops.append(("blur", random.uniform(0.1, 2.0)))      # sigma is drawn inside apply_image
    if include_erasing:
        for prob, (sl, sh, r1, r2) in ((0.7, (0.05, 0.2, 0.3, 3.3)), (0.5, (0.02, 0.2, 0.1, 6)), (0.3, (0.02, 0.2, 0.05, 8))):
            if np.random.uniform(0, 1.0) < prob:
                rect = erase_params(H, W, sl, sh, r1, r2)
                if rect is not None:
                    ops.append(("erase", rect, np.random.rand(rect[2], rect[3], 3)))
    if mic is not None:
        ratio, block = mic
        if np.random.uniform(0, 1.0) < 1.0:                 # RandomApply(MICTransform, prob=1.0) still draws
            mh, mw = mic_grid(H, W, block)
            ops.append(("mic", np.random.rand(mh, mw) > ratio))
    return ops


def apply_ops(img_u8: np.ndarray, ops: List[tuple]) -> np.ndarray:
    img = img_u8
    for op in ops:
        if op[0] == "contrast":
            img = contrast(img, op[1])
        elif op[0] == "brightness":
            img = brightness(img, op[1])
        elif op[0] == "saturation":
            img = saturation(img, op[1])
        elif op[0] == "blur":
            img = gaussian_blur(img, op[1])
        elif op[0] == "erase":
            img = erase(img, op[1], op[2])
        elif op[0] == "mic":
            img = mic_mask(img, op[1])
        else:
            raise ValueError(op[0])
    return img
