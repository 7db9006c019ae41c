"""Pins oracle/aldi_ops.py (the ALDI-owned arithmetic) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import aldi_ops as ao


def T(a):
    return torch.from_numpy(np.asarray(a))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_discriminators(golden_dir):
    g = load(golden_dir, "g1_discriminators.npz")
    assert int(g["conv_nparams_default"]) == 590337 and int(g["fc_nparams_default"]) == 1050625
    for tag, fn, keys in (("conv", ao.conv_discriminator, ("model.0", "model.4")), ("fc", ao.fc_discriminator, ("model.1", "model.3"))):
        P = [T(g[f"{tag}_sd.{k}.{s}"]).clone().requires_grad_(True) for k in keys for s in ("weight", "bias")]
        for labeled in (1, 0):
            x = T(g[f"{tag}_x"]).clone().requires_grad_(True)
            for p in P:
                p.grad = None
            preds = fn(ao.grad_reverse(x), *P)
            loss = ao.domain_loss(preds, bool(labeled), 0.01)
            loss.backward()
            np.testing.assert_allclose(preds.detach().numpy(), g[f"{tag}_preds"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss_l{labeled}"]), rtol=1e-6)
            np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_dx_l{labeled}"], rtol=1e-4, atol=1e-9)
            i = 0
            for k in keys:
                for s in ("weight", "bias"):
                    np.testing.assert_allclose(P[i].grad.numpy(), g[f"{tag}_grad_l{labeled}.{k}.{s}"], rtol=1e-4, atol=1e-8)
                    i += 1


@pytest.mark.parametrize("tag", ["r256k8", "r64k80", "nofg"])
def test_g2_roih_losses(golden_dir, tag):
    g = load(golden_dir, "g2_roih_losses.npz")
    for lt in ("CE", "KL"):
        for Tm in (1.0, 0.5):
            s1 = T(g[f"{tag}_s_logits"]).clone().requires_grad_(True)
            s2 = T(g[f"{tag}_s_deltas"]).clone().requires_grad_(True)
            L = ao.roih_distill_losses(s1, s2, T(g[f"{tag}_t_logits"]), T(g[f"{tag}_t_deltas"]), Tm, lt)
            (L["loss_cls_ce"] + L["loss_roih_l1"]).backward()
            key = f"{tag}_{lt}_T{Tm}"
            np.testing.assert_allclose(float(L["loss_cls_ce"]), float(g[key + "_loss_cls_ce"]), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(float(L["loss_roih_l1"]), float(g[key + "_loss_roih_l1"]), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(s1.grad.numpy(), g[key + "_dlogits"], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(s2.grad.numpy(), g[key + "_ddeltas"], rtol=1e-5, atol=1e-8)
    with pytest.raises(ValueError):
        ao.roih_distill_losses(s1, s2, s1, s2, 1.0, "XX")


@pytest.mark.parametrize("tag", ["mix", "zerofg"])
def test_g3_rpn_losses_index_quirk(golden_dir, tag):
    g = load(golden_dir, "g3_rpn_losses.npz")
    nl = len(g["shapes"])
    for Tm in (1.0, 0.5):
        s_lo = [T(g[f"{tag}_s_logits{l}"]).clone().requires_grad_(True) for l in range(nl)]
        s_de = [T(g[f"{tag}_s_deltas{l}"]).clone().requires_grad_(True) for l in range(nl)]
        t_lo = [T(g[f"{tag}_t_logits{l}"]) for l in range(nl)]
        t_de = [T(g[f"{tag}_t_deltas{l}"]) for l in range(nl)]
        L = ao.rpn_distill_losses(s_lo, s_de, t_lo, t_de, T(g[f"{tag}_labels"]), Tm)
        (L["loss_obj_bce"] + L["loss_rpn_l1"]).backward()
        key = f"{tag}_T{Tm}"
        np.testing.assert_allclose(float(L["loss_obj_bce"]), float(g[key + "_loss_obj_bce"]), rtol=1e-6)
        np.testing.assert_allclose(float(L["loss_rpn_l1"]), float(g[key + "_loss_rpn_l1"]), rtol=1e-6, atol=1e-8)
        for l in range(nl):
            gl = s_lo[l].grad if s_lo[l].grad is not None else torch.zeros_like(s_lo[l])
            gd = s_de[l].grad if s_de[l].grad is not None else torch.zeros_like(s_de[l])
            np.testing.assert_allclose(gl.numpy(), g[f"{key}_dlogits{l}"], rtol=1e-5, atol=1e-9)
            np.testing.assert_allclose(gd.numpy(), g[f"{key}_ddeltas{l}"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["thr08", "empty"])
def test_g4_process_bbox(golden_dir, tag):
    g = load(golden_dir, "g4_process_bbox.npz")
    pred = {"image_size": (100, 200), "scores": T(g[f"{tag}_in_scores"]), "pred_boxes": T(g[f"{tag}_in_boxes"]),
            "pred_classes": T(g[f"{tag}_in_classes"])}
    r = ao.process_bbox(pred, 0.8)
    assert np.array_equal(r["gt_boxes"].numpy(), g[f"{tag}_gt_boxes"])
    assert np.array_equal(r["gt_classes"].numpy(), g[f"{tag}_gt_classes"])
    assert np.array_equal(r["scores"].numpy(), g[f"{tag}_scores"])
    assert tuple(g[f"{tag}_image_size"]) == r["image_size"]


def test_g5_ema(golden_dir):
    g = load(golden_dir, "g5_ema.npz")
    keys = [k[3:] for k in g.files if k.startswith("t0.")]
    t0 = OrderedDict((k, T(g["t0." + k])) for k in keys)
    s = OrderedDict((k, T(g["s." + k])) for k in keys)
    t1 = ao.ema_update(t0, s, float(g["alpha"]), it=3, start_iter=2)
    for k in keys:
        assert np.array_equal(t1[k].numpy(), g["t_after_ema." + k]), k        # bit-exact
    t2 = ao.ema_update(t1, s, float(g["alpha"]), it=2, start_iter=2)
    for k in keys:
        assert np.array_equal(t2[k].numpy(), g["t_after_copy." + k]), k
    with pytest.raises(Exception):
        ao.ema_update(OrderedDict(x=torch.ones(1)), OrderedDict(), 0.5, 5, 0)


def test_g6_step_trace(golden_dir):
    traces = json.load(open(os.path.join(golden_dir, "g6_step_trace.json")))
    for name, tr in traces.items():
        log = []

        class Model:
            n = 0

            def __call__(self, data, **kw):
                self.n += 1
                log.append(["model", [d["id"] for d in data], dict(sorted(kw.items()))])
                base = float(self.n)
                out = {"loss_cls": torch.tensor(base, requires_grad=True) * 1.0,
                       "loss_rpn_cls": torch.tensor(base + 0.25, requires_grad=True) * 1.0}
                if kw.get("do_align"):
                    out["loss_da_img"] = torch.tensor(base + 0.5, requires_grad=True) * 1.0
                elif tr["align"]:
                    out["_da"] = torch.tensor(0.0, requires_grad=True) * 1.0
                return out

        class Dist:
            def distill_enabled(self):
                return tr["distill"]

            def __call__(self, t, s):
                log.append(["distiller", [d["id"] for d in t], [d["id"] for d in s]])
                return {"loss_cls": torch.tensor(7.0, requires_grad=True) * 0.0,
                        "loss_cls_ce": torch.tensor(3.0, requires_grad=True) * 1.0}
        n = tr["nper"]
        lab = [{"id": f"L{i}"} for i in range(n)]
        unl = [{"id": f"U{i}"} for i in range(n)]
        c = tr["contents"]
        has_unl = any(x.startswith("unlabeled") for x in c)
        data = (lab if "labeled_weak" in c else None, lab if "labeled_strong" in c else None,
                unl if has_unl else None, unl if "unlabeled_strong" in c else None)
        ld = ao.run_model_labeled_unlabeled(Model(), Dist(), lambda l: log.append(["backward", round(float(l), 6)]), *data,
                                            do_align=tr["align"], backward_at_end=tr["backward_at_end"],
                                            model_batch_size=tr["ims_per_gpu"])
        assert log == tr["log"], name
        assert {k: round(float(v), 6) for k, v in ld.items()} == tr["loss_dict"], name
        assert list(ld.keys()) == sorted(ld.keys(), key=list(ld.keys()).index)
        assert {k: bool(getattr(v, "requires_grad", False)) for k, v in ld.items()} == tr["requires_grad"], name


def test_g8_hard_mask(golden_dir):
    res = json.load(open(os.path.join(golden_dir, "g8_hard_mask.json")))
    hard = {"loss_cls": torch.tensor(1.5), "loss_box_reg": torch.tensor(2.5),
            "loss_rpn_cls": torch.tensor(3.5), "loss_rpn_loc": torch.tensor(4.5)}
    for key in ("0000", "1010", "1111"):
        f = [c == "1" for c in key]
        out = ao.mask_hard_losses(hard, f[0], f[1], f[2], f[3])
        out.update({"loss_obj_bce": torch.tensor(0.1), "loss_rpn_l1": torch.tensor(0.2)})
        out.update({"loss_cls_ce": torch.tensor(0.3), "loss_roih_l1": torch.tensor(0.4)})
        assert list(out.keys()) == res[key]["keys"]
        assert {k: round(float(v), 6) for k, v in out.items()} == res[key]["values"]


@pytest.mark.parametrize("tag", ["conv2", "conv0", "fc2", "fc0"])
def test_g11_discriminators_with_other_hidden_dims(golden_dir, tag):
    """two hidden layers and none at all (aldi/align.py:103-135 accept any hidden_dims list)"""
    g = load(golden_dir, "g11_discriminators_deep.npz")
    keys = [str(k) for k in g[f"{tag}_keys"]]
    assert keys == {"conv2": ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias", "model.6.weight", "model.6.bias"],
                    "conv0": ["model.2.weight", "model.2.bias"],
                    "fc2": ["model.1.weight", "model.1.bias", "model.3.weight", "model.3.bias", "model.5.weight", "model.5.bias"],
                    "fc0": ["model.1.weight", "model.1.bias"]}[tag]
    from aldi_amd.arch import disc_convs                      # the engine's layout uses the same nn.Sequential indices
    spec = {"conv2": dict(img=dict(input_dim=32, hidden_dims=[16, 32]), ins=False), "conv0": dict(img=dict(input_dim=32, hidden_dims=[]), ins=False),
            "fc2": dict(img=False, ins=dict(input_dim=64, hidden_dims=[32, 16])), "fc0": dict(img=False, ins=dict(input_dim=64, hidden_dims=[]))}[tag]
    pre = "img_align." if tag.startswith("conv") else "ins_align."
    assert [pre + k.rsplit(".", 1)[0] for k in keys[::2]] == list(disc_convs(spec["img"], spec["ins"]))
    fn = ao.conv_discriminator if tag.startswith("conv") else ao.fc_discriminator
    P = [T(g[f"{tag}_sd.{k}"]).clone().requires_grad_(True) for k in keys]
    for labeled in (1, 0):
        x = T(g[f"{tag}_x"]).clone().requires_grad_(True)
        for p in P:
            p.grad = None
        preds = fn(ao.grad_reverse(x), *P)
        loss = ao.domain_loss(preds, bool(labeled), 0.01)
        loss.backward()
        np.testing.assert_allclose(preds.detach().numpy(), g[f"{tag}_preds"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss_l{labeled}"]), rtol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_dx_l{labeled}"], rtol=1e-4, atol=1e-9)
        for p, k in zip(P, keys):
            np.testing.assert_allclose(p.grad.numpy(), g[f"{tag}_grad_l{labeled}.{k}"], rtol=1e-4, atol=1e-8)
