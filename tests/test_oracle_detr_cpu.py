"""oracle/deformable_detr.py (the Deformable-DETR detector after its backbone, restated from the published algorithm) against the golden
vectors tests/golden/make_detr_golden.py generated with transformers' DeformableDetrForObjectDetection, and -- when transformers is
importable -- against a live run on fresh inputs.  CPU only."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "g12_deformable_detr.npz")
CFG = dict(d_model=64, num_levels=4, enc_layers=2, dec_layers=3, n_heads=4, enc_points=4, dec_points=4)


def _load():
    z = np.load(GOLD)
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")}
    feats = [torch.from_numpy(z[f"feat{l}"]) for l in range(3)]
    targets = [{"labels": torch.from_numpy(z[f"tgt{i}/labels"]), "boxes": torch.from_numpy(z[f"tgt{i}/boxes"])} for i in range(2)]
    return z, p, feats, torch.from_numpy(z["image_mask"]), targets


def test_forward_matches_the_golden_vectors():
    from oracle import deformable_detr as D
    z, p, feats, image_mask, _ = _load()
    logits, boxes = D.forward(p, feats, image_mask, **CFG)
    assert logits.shape == z["logits"].shape and boxes.shape == z["boxes"].shape
    assert np.abs(logits.numpy() - z["logits"]).max() <= 2e-4 * max(1.0, np.abs(z["logits"]).max())
    assert np.abs(boxes.numpy() - z["boxes"]).max() <= 1e-5
    # the padded image exercises the masks: its valid ratios differ from 1 and the padded rows of the value maps are zeroed
    src, pos, mask, shapes, vr = D.prepare_levels(p, feats, image_mask, CFG["d_model"], CFG["num_levels"])
    assert shapes == [(12, 16), (6, 8), (3, 4), (2, 2)] and bool(mask[1].any()) and not bool(mask[0].any())
    assert float(vr[0].min()) == 1.0 and float(vr[1].min()) < 1.0


def test_losses_match_the_golden_vectors():
    from oracle import deformable_detr as D
    z, p, feats, image_mask, targets = _load()
    logits, boxes = torch.from_numpy(z["logits"]), torch.from_numpy(z["boxes"])
    d, total = D.criterion(logits, boxes, targets, weights=(1.0, 5.0, 2.0))          # (transformers weighs the classification loss 1)
    for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_ce_0", "loss_bbox_1", "loss_giou_1"):
        assert abs(float(d[k]) - float(z["loss/" + k])) <= 1e-4 * max(1.0, abs(float(z["loss/" + k]))), (k, float(d[k]), float(z["loss/" + k]))
    assert abs(float(total) - float(z["loss"])) <= 1e-4 * float(z["loss"])
    # the reference's own coefficients (configs/Base-DETR.yaml:29-34: CLS 2, BBOX 5, GIOU 2)
    d2, total2 = D.criterion(logits, boxes, targets, weights=(2.0, 5.0, 2.0))
    ce = sum(float(v) for k, v in d2.items() if k.startswith("loss_ce"))
    assert abs(float(total2) - (float(total) + ce)) <= 1e-4 * float(total2)


def test_matching_is_one_to_one_and_empty_targets_are_handled():
    from oracle import deformable_detr as D
    z, p, feats, image_mask, targets = _load()
    logits, boxes = torch.from_numpy(z["logits"])[-1], torch.from_numpy(z["boxes"])[-1]
    idx = D.hungarian_match(logits, boxes, targets)
    for (i, j), t in zip(idx, targets):
        assert len(i) == len(j) == len(t["labels"]) and len(set(i.tolist())) == len(i) and sorted(j.tolist()) == list(range(len(j)))
    none = [{"labels": torch.zeros(0, dtype=torch.long), "boxes": torch.zeros(0, 4)} for _ in range(2)]
    d, _ = D.set_losses(logits, boxes, none)
    assert float(d["loss_bbox"]) == 0.0 and float(d["loss_giou"]) == 0.0 and float(d["loss_ce"]) > 0.0
    scores, labels, xyxy = D.post_process(logits, boxes, [(96, 128), (80, 100)], topk=10)
    assert scores.shape == (2, 10) and bool((scores[:, :-1] >= scores[:, 1:]).all()) and xyxy.shape == (2, 10, 4)


def test_live_transformers_run_on_fresh_inputs():
    pytest.importorskip("transformers")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_detr_golden", os.path.join(ROOT, "tests", "golden", "make_detr_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    from oracle import deformable_detr as D
    m, cfg = G.build(seed=5)
    x, pixel_mask, labels, fmaps, logits, boxes, out = G.run(m, seed=7)
    p = G.rename({k: v.detach() for k, v in m.state_dict().items()})
    lo, bo = D.forward(p, fmaps, pixel_mask == 0, **CFG)
    assert (lo - logits).abs().max().item() <= 2e-4 * max(1.0, logits.abs().max().item())
    assert (bo - boxes).abs().max().item() <= 1e-5
    targets = [{"labels": t["class_labels"], "boxes": t["boxes"]} for t in labels]
    _, total = D.criterion(lo, bo, targets, weights=(1.0, 5.0, 2.0))
    assert abs(float(total) - float(out.loss)) <= 1e-4 * float(out.loss)


def test_dropout_sites_and_order_against_live_transformers():
    """the oracle's dropout hook (`drop(site, tensor)`: <layer>.dropout1..4 and the decoder self attention's probabilities) sits where
    transformers' layers call nn.functional.dropout, in the same order: both runs take their keep masks from one call counter and must
    agree.  (transformers' layers carry `dropout` / `activation_dropout` / the attention's `dropout`; the authors use one rate for all, as
    configs/Base-DETR.yaml's TRANSFORMER.DROPOUT.  Its encoder's extra dropout on the input embeddings -- not in the authors' code --
    stays at p = 0.)"""
    pytest.importorskip("transformers")
    import importlib.util
    import torch.nn.functional as F
    spec = importlib.util.spec_from_file_location("make_detr_golden", os.path.join(ROOT, "tests", "golden", "make_detr_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    from oracle import deformable_detr as D
    m, cfg = G.build(seed=11)
    m.config._attn_implementation = "eager"                       # (the fused attention paths draw their own masks)
    for layer in list(m.model.encoder.layers) + list(m.model.decoder.layers):
        layer.training = True
        layer.dropout, layer.activation_dropout = 0.3, 0.3
        mlp = getattr(layer, "mlp", None)                         # (newer transformers keep the feed-forward block's two rates on a sub-module)
        if mlp is not None:
            mlp.training = True
            mlp.dropout, mlp.activation_dropout = 0.3, 0.3
        att = layer.self_attn                                     # the decoder's multi-head attention (the deformable attention has no dropout)
        for name in ("attention_dropout", "dropout"):
            if isinstance(getattr(att, name, None), float):
                att.training = True
                setattr(att, name, 0.3)
                if hasattr(att, "config"):
                    att.config._attn_implementation = "eager"

    def mask_of(i, shape):
        n = 1
        for s_ in shape:
            n *= s_
        return ((torch.rand(n, generator=torch.Generator().manual_seed(1000 + i)) > 0.3).float() / 0.7).view(shape)
    calls = []
    orig = F.dropout

    def fake(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        calls.append(tuple(x.shape))
        return x * mask_of(len(calls) - 1, x.shape)
    F.dropout = fake
    try:
        x, pixel_mask, labels, fmaps, logits, boxes, out = G.run(m, seed=13)
    finally:
        F.dropout = orig
    assert len(calls) == 2 * 3 + 3 * 5, calls
    p = G.rename({k: v.detach() for k, v in m.state_dict().items()})
    sites = []

    def drop(name, t):
        sites.append(name)
        return t * mask_of(len(sites) - 1, t.shape)
    lo, bo = D.forward(p, fmaps, pixel_mask == 0, drop=drop, **CFG)
    assert len(sites) == len(calls)
    assert sites[:3] == ["transformer.encoder.layers.0.dropout1", "transformer.encoder.layers.0.dropout2", "transformer.encoder.layers.0.dropout3"]
    assert sites[6:11] == ["transformer.decoder.layers.0.self_attn.attn", "transformer.decoder.layers.0.dropout2", "transformer.decoder.layers.0.dropout1",
                           "transformer.decoder.layers.0.dropout3", "transformer.decoder.layers.0.dropout4"]
    assert (lo - logits).abs().max().item() <= 2e-4 * max(1.0, logits.abs().max().item())
    assert (bo - boxes).abs().max().item() <= 1e-5
    lo0, _ = D.forward(p, fmaps, pixel_mask == 0, **CFG)
    assert (lo0 - logits).abs().max().item() > 1e-2              # the masks really changed the result
