"""Golden vectors for oracle/deformable_detr.py from transformers' DeformableDetrForObjectDetection (the reference's own detector
source is an absent submodule; transformers' port of the authors' implementation is the independent statement available in this image).

A small configuration (d_model 64, 4 heads, 2 + 3 layers, 20 queries, 4 levels, 8 classes) with seeded random weights runs on two
images, the second one padded (so masks, valid ratios and the extra level's mask matter); stored: the backbone's three feature maps, the
image mask, every weight after the backbone under the ORIGINAL implementation's names, the logits / boxes of every decoder layer, the
Hungarian assignment's losses and the weighted total.      python tests/golden/make_detr_golden.py   -> tests/golden/g12_deformable_detr.npz"""
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g12_deformable_detr.npz")


def rename(hf):
    """transformers' parameter names -> the authors' (oracle/deformable_detr.py's)"""
    out = {}
    for k, v in hf.items():
        if "backbone" in k:
            continue
        k = k[len("model."):] if k.startswith("model.") else k
        if k == "level_embed":
            out["transformer.level_embed"] = v
        elif k == "query_position_embeddings.weight":
            out["query_embed.weight"] = v
        elif k.startswith("reference_points."):
            out["transformer." + k] = v
        elif k.startswith("input_proj."):
            out[k] = v
        elif k.startswith("class_embed.0."):
            out["class_embed." + k.split(".", 2)[2]] = v
        elif k.startswith("bbox_embed.0."):
            out["bbox_embed." + k.split(".", 2)[2]] = v
        elif k.startswith("class_embed.") or k.startswith("bbox_embed."):
            continue                                              # tied clones of head 0 (no box refinement)
        elif k.startswith("encoder.layers."):
            k = k.replace("self_attn_layer_norm", "norm1").replace("final_layer_norm", "norm2").replace("mlp.fc1", "linear1").replace("mlp.fc2", "linear2")
            k = k.replace(".fc1.", ".linear1.").replace(".fc2.", ".linear2.")
            out["transformer." + k] = v
        elif k.startswith("decoder.layers."):
            k = k.replace("encoder_attn_layer_norm", "norm1").replace("self_attn_layer_norm", "norm2").replace("final_layer_norm", "norm3")
            k = k.replace("encoder_attn", "cross_attn").replace("mlp.fc1", "linear1").replace("mlp.fc2", "linear2").replace(".fc1.", ".linear1.").replace(".fc2.", ".linear2.")
            out["transformer." + k] = v
        else:
            raise KeyError(k)
    # separate q / k / v projections -> nn.MultiheadAttention's packed in_proj
    for k in [k for k in list(out) if k.endswith("self_attn.q_proj.weight") and ".decoder." in k]:
        pre = k[: -len("q_proj.weight")]
        out[pre + "in_proj_weight"] = torch.cat([out.pop(pre + "q_proj.weight"), out.pop(pre + "k_proj.weight"), out.pop(pre + "v_proj.weight")])
        out[pre + "in_proj_bias"] = torch.cat([out.pop(pre + "q_proj.bias"), out.pop(pre + "k_proj.bias"), out.pop(pre + "v_proj.bias")])
        out[pre + "out_proj.weight"] = out.pop(pre + "o_proj.weight")
        out[pre + "out_proj.bias"] = out.pop(pre + "o_proj.bias")
    return out


def build(seed=0):
    from transformers import DeformableDetrConfig, DeformableDetrForObjectDetection, ResNetConfig
    torch.manual_seed(seed)
    bc = ResNetConfig(embedding_size=16, hidden_sizes=[16, 32, 64, 128], depths=[1, 1, 1, 1], layer_type="bottleneck", out_features=["stage2", "stage3", "stage4"])
    cfg = DeformableDetrConfig(use_timm_backbone=False, backbone_config=bc, use_pretrained_backbone=False, d_model=64, encoder_layers=2, decoder_layers=3,
                               encoder_attention_heads=4, decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_queries=20,
                               num_feature_levels=4, encoder_n_points=4, decoder_n_points=4, num_labels=8, dropout=0.0, attention_dropout=0.0,
                               activation_dropout=0.0, auxiliary_loss=True, with_box_refine=False, two_stage=False,
                               class_cost=2, bbox_cost=5, giou_cost=2, bbox_loss_coefficient=5, giou_loss_coefficient=2, focal_alpha=0.25)
    m = DeformableDetrForObjectDetection(cfg).eval()
    with torch.no_grad():                                          # the default init leaves the heads near zero: make every path matter
        for n, p in m.named_parameters():
            if "backbone" not in n:
                p.add_(torch.randn_like(p) * 0.05)
    return m, cfg


def run(m, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 3, 96, 128, generator=g)
    pixel_mask = torch.ones(2, 96, 128, dtype=torch.long)
    pixel_mask[1, 80:, :] = 0
    pixel_mask[1, :, 100:] = 0
    x = x * pixel_mask[:, None]
    labels = [{"class_labels": torch.tensor([1, 3]), "boxes": torch.tensor([[0.5, 0.5, 0.2, 0.3], [0.3, 0.6, 0.1, 0.2]])},
              {"class_labels": torch.tensor([2]), "boxes": torch.tensor([[0.4, 0.4, 0.3, 0.3]])}]
    feats = {}
    h = m.model.backbone.register_forward_hook(lambda mod, i, o: feats.setdefault("f", o))
    with torch.no_grad():
        out = m(pixel_values=x, pixel_mask=pixel_mask, labels=labels)
    h.remove()
    fmaps = [f for f, _ in feats["f"]]
    logits = torch.stack([a["logits"] for a in out.auxiliary_outputs] + [out.logits])
    boxes = torch.stack([a["pred_boxes"] for a in out.auxiliary_outputs] + [out.pred_boxes])
    return x, pixel_mask, labels, fmaps, logits, boxes, out


def main():
    m, cfg = build()
    x, pixel_mask, labels, fmaps, logits, boxes, out = run(m)
    arrs = {"image_mask": (pixel_mask == 0).numpy(), "logits": logits.numpy(), "boxes": boxes.numpy(), "loss": np.float32(out.loss.item())}
    for l, f in enumerate(fmaps):
        arrs[f"feat{l}"] = f.numpy()
    for k, v in out.loss_dict.items():
        arrs["loss/" + k] = np.float32(float(v))
    for i, t in enumerate(labels):
        arrs[f"tgt{i}/labels"], arrs[f"tgt{i}/boxes"] = t["class_labels"].numpy(), t["boxes"].numpy()
    for k, v in rename({k: v.detach() for k, v in m.state_dict().items()}).items():
        arrs["p/" + k] = v.numpy()
    np.savez_compressed(OUT, **arrs)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024), "loss", float(out.loss))


if __name__ == "__main__":
    sys.exit(main())
