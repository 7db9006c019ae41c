"""Measurement for the strong-augmentation row: strong views per second, weak view resident in HBM,
device chain (aldi_amd/aug.py) vs the reference's own CPU arithmetic (numpy + scipy.ndimage.gaussian_filter, one core --
the reference runs it inside single-threaded dataloader workers).  Prints ONE JSON line."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.ndimage import gaussian_filter

from aldi_amd import aug

H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (800, 1333)))
N = 64
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
dimgs = [torch.from_numpy(i).cuda() for i in imgs]
augs = aug.build_strong_augmentation(include_erasing=True) + [aug.RandomApply(aug.MICTransform(0.5, 32), prob=1.0)]


def run_device(n):
    for i in range(n):
        aug.strong_view(dimgs[i % 4], augs)


np.random.seed(0); random.seed(0)
run_device(8)
torch.cuda.synchronize()
np.random.seed(1); random.seed(1)
t = time.perf_counter()
run_device(N)
torch.cuda.synchronize()
dt = time.perf_counter() - t

# per-transform device times (always applied)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
per = {
    "contrast_us": timed(lambda: aug.RandomContrast(0.6, 1.4).apply_image(dimgs[0], 0.9)),
    "saturation_us": timed(lambda: aug.RandomSaturation(0.6, 1.4).apply_image(dimgs[0], 0.9)),
    "blur_sigma2_us": timed(lambda: aug.RandomBlurTransform((0.1, 2.0)).apply_image(dimgs[0], 2.0)),
    "mic_us": timed(lambda: aug.MICTransform(0.5, 32).apply_image(dimgs[0])),
}

# CPU: the reference's arithmetic for the same chain (scipy for the blur), bounded sample
def cpu_chain(img):
    def blend(im, src, sw, dw):
        a = im.astype(np.float32); a = sw * src + dw * a
        return np.clip(a, 0, 255).astype(np.uint8)
    if np.random.uniform(0, 1.0) < 0.8:
        w = np.random.uniform(0.6, 1.4); img = blend(img, img.mean(), 1 - w, w)
        w = np.random.uniform(0.6, 1.4); img = blend(img, 0, 1 - w, w)
        w = np.random.uniform(0.6, 1.4); img = blend(img, img.dot([0.299, 0.587, 0.114])[:, :, None], 1 - w, w)
    if np.random.uniform(0, 1.0) < 0.2:
        w = np.random.uniform(0, 0); img = blend(img, img.dot([0.299, 0.587, 0.114])[:, :, None], 1 - w, w)
    if np.random.uniform(0, 1.0) < 0.5:
        img = np.clip(gaussian_filter(img.astype(np.float32), sigma=random.uniform(0.1, 2.0)), 0, 255).astype(np.uint8)
    for prob, frac in ((0.7, 0.12), (0.5, 0.1), (0.3, 0.1)):
        if np.random.uniform(0, 1.0) < prob:
            h, w_ = int(H * frac ** 0.5), int(W * frac ** 0.5)
            a = img.astype(np.float32); a[10:10 + h, 10:10 + w_, :] = np.random.rand(h, w_, 3); a[10:10 + h, 10:10 + w_, :] *= 255
            img = np.clip(a, 0, 255).astype(np.uint8)
    m = np.random.rand(round(H / 32), round(W / 32)) > 0.5
    ys = np.minimum((np.arange(H) * (m.shape[0] / H)).astype(int), m.shape[0] - 1); xs = np.minimum((np.arange(W) * (m.shape[1] / W)).astype(int), m.shape[1] - 1)
    a = img.astype(np.float32) * np.repeat(m[ys][:, xs].astype(np.uint8)[..., None], 3, axis=-1)
    return np.clip(a, 0, 255).astype(np.uint8)

torch.set_num_threads(1)
np.random.seed(1); random.seed(1)
t = time.perf_counter(); n_cpu = 0
while time.perf_counter() - t < 10.0:
    cpu_chain(imgs[n_cpu % 4]); n_cpu += 1
dcpu = time.perf_counter() - t
print(json.dumps({"metric": "strong views/sec (ALDI strong augmentation chain, %dx%d, weak view resident in HBM)" % (W, H), "value": round(N / dt, 1),
                  "unit": "images/sec", "n_gpus": 1, "dtype": "u8 (f32/f64 arithmetic as numpy)", "data": "synthetic", "ms_per_image": round(dt / N * 1e3, 3),
                  "per_transform": {k: round(v, 1) for k, v in per.items()},
                  "cpu_baseline": {"value": round(n_cpu / dcpu, 2), "unit": "images/sec", "cores": 1, "kind": "port",
                                   "sample": "%d images in %.1f s, numpy + scipy.ndimage.gaussian_filter (the reference's own arithmetic), one core" % (n_cpu, dcpu)}}))
