"""Debug harness: one training forward/backward on HIP vs the CPU oracle (small image, fp32 by default)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import d2_rcnn as d2
from aldi_amd import synthetic as syn
from aldi_amd.arch import ParamLayout
from aldi_amd.engine import Weights, RCNN

dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32
H, W = (192, 256)
K = 8
cfg = d2.make_cfg(num_classes=K)
sd = syn.init_state_dict(K, seed=1)
_, data, _, _ = syn.make_batch(2, 0, H, W, K, seed=0, boxes_per_image=(3, 6))

# ---- oracle
osd = {k: v.clone() for k, v in sd.items()}
for k in d2.trainable_keys(cfg, osd):
    osd[k].requires_grad_(True)
torch.manual_seed(123)
cap = d2.Captured()
t = time.time()
ol = d2.forward_train(cfg, osd, data, roi_seed=77, cap=cap)
sum(ol.values()).backward()
print("oracle", {k: round(float(v), 5) for k, v in ol.items()}, "%.1fs" % (time.time() - t))

# ---- HIP
dev = torch.device("cuda")
layout = ParamLayout(K)
wts = Weights(layout, dev, dtype, trainable=True)
wts.load_state_dict(sd)
m = RCNN(wts, K)
torch.manual_seed(123)
scales = {k: 1.0 for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")}
c = m.forward_train([d["image"] for d in data], [d["instances"] for d in data], roi_seed=77)
m.backward(c, scales)
torch.cuda.synchronize()
hl = {k: float(v) for k, v in m.loss_dict(c).items()}
print("hip   ", {k: round(v, 5) for k, v in hl.items()}, "err flag", int(m.err))

def cmp(name, a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    e = (a - b).abs().max().item(); s = b.abs().max().item()
    print(f"  {name:34s} max|d|={e:.3e}  max|ref|={s:.3e}  rel={e / max(s, 1e-12):.2e}")

for i, k in enumerate(("p2", "p3", "p4", "p5", "p6")):
    cmp(k, c.P[i].permute(0, 3, 1, 2), cap["features"][k])
for l in range(5):
    cmp(f"rpn_logits{l}", c.head[l][..., :3].permute(0, 3, 1, 2), cap["rpn_logits"][l])
    cmp(f"rpn_deltas{l}", c.head[l][..., 3:15].permute(0, 3, 1, 2), cap["rpn_deltas"][l])
lab_o = torch.stack(cap["rpn_gt_labels"]).to(torch.int32)
print("  rpn labels equal:", bool((c.rpn_labels.cpu() == lab_o).all()), int((lab_o == 1).sum()), int((lab_o == 0).sum()))
for n in range(2):
    po = cap["proposals"][n]["proposal_boxes"]
    cnt = int(c.prop_count[n])
    print(f"  proposals img{n}: count hip={cnt} oracle={len(po)}")
    k = min(cnt, len(po))
    cmp(f"proposal boxes {n}", c.props[n, :k], po[:k])
    cmp(f"proposal logits {n}", c.prop_scores[n, :k], cap["proposals"][n]["objectness_logits"][:k])
sidx_o = torch.cat([s["sampled_idxs"] for s in cap["sampled"]]).to(torch.int32)
print("  sampled idx equal:", bool(c.r_idx.cpu().shape == sidx_o.shape and (c.r_idx.cpu() == sidx_o).all()), c.R, len(sidx_o))
cls_o = torch.cat([s["gt_classes"] for s in cap["sampled"]]).to(torch.int32)
print("  sampled cls equal:", bool((c.r_cls.cpu()[:len(cls_o)] == cls_o).all()))
cmp("pooled", c.pooled.permute(0, 3, 1, 2), cap["pooled"])
cmp("box_head_out", c.fc2.view(c.R, -1), cap["box_head_out"])
cmp("box_scores", c.pred[:, :K + 1], cap["box_scores"])
cmp("box_deltas", c.pred[:, K + 1:K + 1 + 4 * K], cap["box_deltas"])
flat = torch.zeros(layout.n_total)
flat[:layout.n_train] = wts.grad.cpu()
g = layout.unpack(flat)
for k in ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.bias", "roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.weight",
          "proposal_generator.rpn_head.objectness_logits.weight", "proposal_generator.rpn_head.anchor_deltas.weight", "proposal_generator.rpn_head.conv.weight",
          "proposal_generator.rpn_head.conv.bias", "backbone.fpn_output2.weight", "backbone.fpn_output5.weight", "backbone.fpn_lateral2.weight", "backbone.fpn_lateral4.weight", "backbone.fpn_lateral5.bias",
          "backbone.bottom_up.res5.2.conv3.weight", "backbone.bottom_up.res5.0.shortcut.weight", "backbone.bottom_up.res5.0.conv1.weight", "backbone.bottom_up.res4.5.conv2.weight",
          "backbone.bottom_up.res4.0.conv1.weight", "backbone.bottom_up.res3.3.conv1.weight", "backbone.bottom_up.res3.0.conv2.weight", "backbone.bottom_up.res3.0.shortcut.weight"):
    cmp("grad " + k.replace("backbone.bottom_up.", "").replace("proposal_generator.", "").replace("roi_heads.", ""), g[k], osd[k].grad)
