import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_detr_gpu as T
from aldi_amd.model import build_aldi
from oracle import d2_rcnn as d2, deformable_detr as D
cfg = T._detr_cfg(); model = build_aldi(cfg)
gen = torch.Generator().manual_seed(5); data = T._detr_batch(gen)
model.weights.zero_grad(); ld = model(data); sum(ld.values()).backward(); torch.cuda.synchronize()
W = model.weights
sd_b = {k: v.detach().cpu().double() for k, v in W.backbone.state_dict().items()}
p_t = {k: v.detach().cpu().double().requires_grad_(True) for k, v in W.tr.state_dict().items()}
ocfg = d2.make_cfg(pixel_mean=list(cfg.MODEL.PIXEL_MEAN), pixel_std=list(cfg.MODEL.PIXEL_STD))
x, sizes = d2.preprocess(ocfg, [d["image"] for d in data]); stages = []
d2.resnet_fpn(ocfg, sd_b, x.double(), stages)
mask = torch.ones(len(data), x.shape[2], x.shape[3], dtype=torch.bool)
for i, (h, w) in enumerate(sizes): mask[i, :h, :w] = False
dims = dict(d_model=256, num_levels=4, enc_layers=2, dec_layers=2, n_heads=8, enc_points=4, dec_points=4)
lo, bo = D.forward(p_t, stages[1:4], mask, **dims)
targets = [{"labels": t["labels"], "boxes": t["boxes"].double()} for t in model._targets([d["instances"] for d in data], sizes)]
ref, tot = D.criterion(lo, bo, targets, weights=(2.0, 5.0, 2.0)); tot.backward()
got = W.tr.state_dict(W.tr.grad)
for k, v in p_t.items():
    if "encoder.layers" in k and ("sampling" in k or "attention_w" in k or "value" in k):
        e = (got[k].cpu().double() - v.grad).abs().max().item(); m = v.grad.abs().max().item()
        print(k, "err %.3e max %.3e" % (e, m))
k = "transformer.encoder.layers.0.self_attn.sampling_offsets.weight"
d = (got[k].cpu().double() - p_t[k].grad)
print("rows with error:", (d.abs().max(1)[0] > 1e-3 * p_t[k].grad.abs().max()).nonzero().view(-1).tolist()[:40])
