"""The oracle's C helpers (roi_align / nms) agree with its pure-torch statements. CPU only."""
import os
import subprocess

import pytest
import torch

from oracle import d2_rcnn as d2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    d2._CLIB = False
    assert d2._clib() is not None


def test_roi_align_c_vs_torch():
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 6, 20, 28, generator=g, requires_grad=True)
    x1 = torch.rand(40, generator=g) * 80 - 5
    y1 = torch.rand(40, generator=g) * 60 - 5
    rois = torch.stack([torch.randint(0, 2, (40,), generator=g).float(), x1, y1, x1 + torch.rand(40, generator=g) * 60 + 0.5,
                        y1 + torch.rand(40, generator=g) * 50 + 0.5], 1)
    a = d2.roi_align(feat, rois, 7, 0.25)
    ga, = torch.autograd.grad((a * torch.arange(a.numel()).view_as(a).float().sin()).sum(), feat)
    b = d2.roi_align_torch(feat, rois, 7, 0.25)
    gb, = torch.autograd.grad((b * torch.arange(b.numel()).view_as(b).float().sin()).sum(), feat)
    assert (a - b).abs().max() < 1e-5
    assert (ga - gb).abs().max() < 1e-4


def test_nms_c_vs_torch():
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(700, 2, generator=g) * 100
    wh = torch.rand(700, 2, generator=g) * 40 + 1
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(700, generator=g)
    scores[10] = scores[20]            # a tie: lower index first
    for thr in (0.3, 0.7):
        assert torch.equal(d2.nms(boxes, scores, thr), d2.nms_torch(boxes, scores, thr))
