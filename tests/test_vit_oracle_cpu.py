"""Pins the ViTDet oracle (oracle/d2_vitdet.py) on CPU: its trunk against transformers' VitDetModel (an independent
implementation of detectron2's ViT present in the image), and the host-side layout logic of aldi_amd.vit against both."""
import torch


def _hf(cfg, drop_path=0.0):
    from transformers import VitDetConfig, VitDetModel
    hc = VitDetConfig(hidden_size=cfg["embed"], num_hidden_layers=cfg["depth"], num_attention_heads=cfg["heads"], mlp_ratio=4,
                      hidden_act="gelu", dropout_prob=0.0, layer_norm_eps=cfg["ln_eps"], image_size=cfg["rel_input"] * cfg["patch"],
                      pretrain_image_size=cfg["pretrain_grid"] * cfg["patch"], patch_size=cfg["patch"], num_channels=3, qkv_bias=True,
                      drop_path_rate=drop_path, window_block_indices=[i for i in range(cfg["depth"]) if i not in cfg["global_blocks"]],
                      residual_block_indices=[], use_absolute_position_embeddings=True, use_relative_position_embeddings=True,
                      window_size=cfg["window"])
    torch.manual_seed(3)
    m = VitDetModel(hc).float().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn_like(p) * (0.2 if ("rel_pos" in n or "position" in n) else 0.05) + (1.0 if "norm" in n and n.endswith("weight") else 0.0))
    return m


def _to_d2(m, cfg, prefix="backbone.net."):
    sd, out = m.state_dict(), {}
    out[prefix + "pos_embed"] = sd["embeddings.position_embeddings"]
    out[prefix + "patch_embed.proj.weight"] = sd["embeddings.projection.weight"]
    out[prefix + "patch_embed.proj.bias"] = sd["embeddings.projection.bias"]
    ren = {"attn.rel_pos_h": "attention.rel_pos_h", "attn.rel_pos_w": "attention.rel_pos_w", "attn.qkv.weight": "attention.qkv.weight",
           "attn.qkv.bias": "attention.qkv.bias", "attn.proj.weight": "attention.proj.weight", "attn.proj.bias": "attention.proj.bias"}
    for i in range(cfg["depth"]):
        for a in ("norm1.weight", "norm1.bias", "attn.rel_pos_h", "attn.rel_pos_w", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                  "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"):
            out[f"{prefix}blocks.{i}.{a}"] = sd[f"encoder.layer.{i}.{ren.get(a, a)}"]
    return out


def test_oracle_vit_matches_transformers_vitdet():
    from oracle import d2_vitdet as ov
    vc = dict(embed=128, depth=4, heads=2, patch=16, window=7, global_blocks=(1, 3), pretrain_grid=4, rel_input=10, ln_eps=1e-6)
    m = _hf(vc)
    sd = _to_d2(m, vc)
    torch.manual_seed(4)
    x = torch.randn(2, 3, 96, 144)                  # 6 x 9 tokens: windows pad to 7 x 14; abs-pos and global rel-pos are resized
    xr = x.clone().requires_grad_(True)
    ref = m(xr).last_hidden_state
    xo = x.clone().requires_grad_(True)
    out = ov.vit_forward(vc, sd, xo)
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
    g = torch.randn_like(ref)
    ref.backward(g)
    out.backward(g)
    assert (xo.grad - xr.grad).abs().max().item() < 2e-4 * xr.grad.abs().max().item()


def test_window_maps_match_oracle_partition():
    from aldi_amd.vit import window_maps
    from oracle import d2_vitdet as ov
    N, gh, gw, ws, E = 2, 9, 17, 7, 4
    x = torch.arange(N * gh * gw * E, dtype=torch.float32).view(N, gh, gw, E) + 1
    win, inv, nW = window_maps(N, gh, gw, ws, "cpu")
    w, padded = ov.to_windows(x, ws)
    assert nW == w.shape[0]
    flat = x.view(-1, E)
    mine = torch.where(win[:, None] >= 0, flat[win.clamp_min(0).long()], torch.zeros(1))
    assert torch.equal(mine, w.reshape(-1, E))
    back = ov.from_windows(w, ws, padded, (gh, gw)).reshape(-1, E)
    assert torch.equal(w.reshape(-1, E)[inv.long()], back)


def test_vit_params_layout_roundtrip_and_packs():
    """detectron2-shaped state_dict -> flat NHWC / packed layout -> back, bit for bit; packed head tensors are contiguous rows"""
    from aldi_amd.vit import VitConfig, VitParams
    cfg = VitConfig(embed=128, depth=2, heads=2, window=7, global_blocks=(1,), pretrain_grid=4, rel_input=10, sfp=True, num_classes=8,
                    fc_dim=64)
    P = VitParams(cfg, "cpu")
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(shape, generator=g) for k, (shape, _) in P.spec.items()}
    flat = P.flatten(sd)
    back = P.state_dict_like(flat)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    C = cfg.fpn_channels
    o, n = P.pack_off["rpn_head_out.weight"]
    rows = flat[o:o + n].view(16, C)
    assert torch.equal(rows[:3], sd["proposal_generator.rpn_head.objectness_logits.weight"].view(3, C))
    assert torch.equal(rows[3:15], sd["proposal_generator.rpn_head.anchor_deltas.weight"].view(12, C))
    assert rows[15].abs().max() == 0
    o, n = P.pack_off["box_pred.bias"]
    assert torch.equal(flat[o:o + 9], sd["roi_heads.box_predictor.cls_score.bias"]) and n == 48
    # conv kernels are channel-last in the flat buffer, the fc1 input is (7, 7, C)
    w = sd["roi_heads.box_head.conv2.weight"]
    assert torch.equal(flat[P.off["roi_heads.box_head.conv2.weight"]:][:w.numel()].view(C, 3, 3, C), w.permute(0, 2, 3, 1))
    f = sd["roi_heads.box_head.fc1.weight"]
    assert torch.equal(flat[P.off["roi_heads.box_head.fc1.weight"]:][:f.numel()].view(64, 7, 7, C), f.view(64, C, 7, 7).permute(0, 2, 3, 1))
    # weight decay split: norms of the transformer blocks and pos_embed sit in the undecayed tail
    assert all((P.off[k] >= P.n_decay) == (not d) for k, (_, d) in P.spec.items())
