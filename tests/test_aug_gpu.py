"""Device-side strong augmentation (aldi_amd/aug.py -> csrc/aug.hip) vs the reference's own outputs (golden g9) and vs the
oracle on the full chain with identical RNG streams.  Byte work: every comparison is bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_aug.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_alidi_owned_transforms_vs_reference_golden(g9, tag):
    from aldi_amd import aug
    img = g9[f"img_{tag}"]
    H, W, _ = img.shape
    d = dev(img)
    for i in range(3):
        random.seed(int(g9[f"blur_{tag}{i}_seed"]))
        out = aug.RandomBlurTransform((0.1, 2.0)).apply_image(d)             # draws sigma like the reference
        assert np.array_equal(out.cpu().numpy(), g9[f"blur_{tag}{i}"]), ("blur", tag, i)
    for i in range(3):
        seed, sl, sh, r1, r2 = g9[f"erase_{tag}{i}_cfg"]
        random.seed(int(seed))
        np.random.seed(int(seed))
        out = aug.RandomEraseTransform(sl=sl, sh=sh, r1=r1, r2=r2, value="random").apply_image(d)
        assert np.array_equal(out.cpu().numpy(), g9[f"erase_{tag}{i}"]), ("erase", tag, i)
    for i in range(2):
        seed, ratio, block = g9[f"mic_{tag}{i}_cfg"]
        np.random.seed(int(seed))
        out = aug.MICTransform(ratio, int(block)).apply_image(d)
        assert np.array_equal(out.cpu().numpy(), g9[f"mic_{tag}{i}"]), ("mic", tag, i)
    assert np.array_equal(d.cpu().numpy(), img)                              # inputs are never modified in place


@pytest.mark.parametrize("shape", [(64, 96), (211, 333), (800, 1333)])
def test_colour_and_blur_vs_oracle(shape):
    from aldi_amd import aug
    from oracle import aug_ops as ao
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, (*shape, 3), dtype=np.uint8)
    img[: shape[0] // 4] = 250                                              # saturating region
    img[shape[0] // 4: shape[0] // 2, :, 1] = 3
    d = dev(img)
    for w in (0.6, 0.873, 1.0, 1.4):
        assert np.array_equal(aug.RandomContrast(0.6, 1.4).apply_image(d, w).cpu().numpy(), ao.contrast(img, w)), ("contrast", w)
        assert np.array_equal(aug.RandomBrightness(0.6, 1.4).apply_image(d, w).cpu().numpy(), ao.brightness(img, w)), ("brightness", w)
        assert np.array_equal(aug.RandomSaturation(0.6, 1.4).apply_image(d, w).cpu().numpy(), ao.saturation(img, w)), ("saturation", w)
    assert np.array_equal(aug.RandomSaturation(0, 0).apply_image(d, 0.0).cpu().numpy(), ao.saturation(img, 0.0))
    for sigma in (0.1, 0.124, 0.9, 2.0):                                    # radius 0 (identity taps) .. 8
        assert np.array_equal(aug.RandomBlurTransform((0.1, 2.0)).apply_image(d, sigma).cpu().numpy(), ao.gaussian_blur(img, sigma)), ("blur", sigma)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_full_strong_chain_vs_oracle_same_rng_streams(seed):
    """build_strong_augmentation (+ MIC) end to end: the device chain consumes np.random / random exactly like the oracle's
    restatement of the reference chain, and produces the same bytes."""
    from aldi_amd import aug
    from oracle import aug_ops as ao
    H, W = 160 + 7 * seed, 224 + 3 * seed
    img = np.random.default_rng(100 + seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
    np.random.seed(seed); random.seed(seed)
    ops_ = ao.draw_strong_params(H, W, include_erasing=True, mic=(0.5, 32))
    state_np, state_py = np.random.get_state()[1].copy(), random.getstate()
    ref = ao.apply_ops(img, ops_)
    np.random.seed(seed); random.seed(seed)
    augs = aug.build_strong_augmentation(include_erasing=True) + [aug.RandomApply(aug.MICTransform(0.5, 32), prob=1.0)]
    out = aug.strong_view(dev(img), augs, chw=True)
    assert np.array_equal(np.random.get_state()[1], state_np) and random.getstate() == state_py     # same draws consumed
    assert np.array_equal(out.cpu().numpy(), ref.transpose(2, 0, 1)), [o[0] for o in ops_]


def test_argument_errors():
    from aldi_amd import aug
    with pytest.raises(ValueError):
        aug.RandomBrightness(0.6, 1.4).apply_image(torch.zeros(3, 8, 8, dtype=torch.uint8, device=DEV))   # CHW is not accepted
    with pytest.raises(Exception):
        aug.RandomEraseTransform().apply_image(torch.zeros(8, 8, 3, dtype=torch.uint8, device=DEV), rect=(4, 4, 8, 8), fill=np.zeros((8, 8, 3)))
