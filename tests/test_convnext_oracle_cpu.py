"""Pins oracle/d2_convnext.py against golden g10: outputs and gradients of the REFERENCE's own ConvNeXt class (aldi/backbone.py)."""
import os

import numpy as np
import torch


def test_oracle_convnext_matches_reference_golden(golden_dir):
    from oracle import d2_convnext as oc
    G = np.load(os.path.join(golden_dir, "g10_convnext.npz"))
    pre = "backbone.bottom_up."
    sd = {pre + str(k): torch.from_numpy(G["sd." + str(k)]).clone().requires_grad_(True) for k in G["keys"]}
    img = torch.from_numpy(G["img"]).float()
    x = img - torch.tensor([103.530, 116.280, 123.675]).view(1, 3, 1, 1)
    outs = oc.convnext_forward(dict(depths=(1, 1, 2, 1)), sd, x)
    for i in range(4):
        ref = torch.from_numpy(G[f"out{i}"])
        assert (outs[i] - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item()), i
    torch.autograd.backward(outs, [torch.from_numpy(G[f"gout{i}"]) for i in range(4)])
    for k in G["keys"]:
        ref = torch.from_numpy(G["grad." + str(k)])
        got = sd[pre + str(k)].grad
        assert (got - ref).abs().max().item() < 2e-4 * max(1e-3, ref.abs().max().item()), str(k)
