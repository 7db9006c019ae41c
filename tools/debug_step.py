"""Debug harness: full ALDI iterations (EMA tick, source step, [align], distill step, SGD) on HIP vs the CPU oracle."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import d2_rcnn as d2, aldi_ops as ao
from aldi_amd import synthetic as syn
from aldi_amd.config import get_cfg, add_aldi_config
from aldi_amd.trainer import ALDITrainer

ALIGN = "align" in sys.argv
BF16 = "bf16" in sys.argv
H, W, K = 192, 256, 8
STEPS = 2


def make_cfg():
    cfg = get_cfg(); add_aldi_config(cfg)
    cfg.merge_from_list(["MODEL.ROI_HEADS.NUM_CLASSES", K, "DATASETS.BATCH_CONTENTS", ("labeled_strong", "unlabeled_strong"),
                         "DATASETS.BATCH_RATIOS", (1, 1), "SOLVER.IMS_PER_BATCH", 4, "SOLVER.IMS_PER_GPU", 2, "SOLVER.BACKWARD_AT_END", False,
                         "SOLVER.BASE_LR", 0.02, "SOLVER.WARMUP_ITERS", 0, "SOLVER.STEPS", (1000,), "SOLVER.AMP.ENABLED", BF16,
                         "EMA.ENABLED", True, "EMA.ALPHA", 0.9, "DOMAIN_ADAPT.TEACHER.ENABLED", True,
                         "DOMAIN_ADAPT.DISTILL.ROIH_CLS_ENABLED", True, "DOMAIN_ADAPT.DISTILL.OBJ_ENABLED", True,
                         "DOMAIN_ADAPT.DISTILL.ROIH_REG_ENABLED", True, "DOMAIN_ADAPT.DISTILL.RPN_REG_ENABLED", True,
                         "DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED", ALIGN, "DOMAIN_ADAPT.ALIGN.INS_DA_ENABLED", ALIGN, "SEED", 1])
    cfg.SYNTHETIC = type(cfg)({"HEIGHT": H, "WIDTH": W, "FIXED": False})
    return cfg


cfg = make_cfg()
# ---------------- HIP
random.seed(0)
torch.manual_seed(123)
tr = ALDITrainer(cfg)
loader_for_oracle = iter(ALDITrainer.build_train_loader(cfg))
hip_losses = []
recorded = []
_pl = tr._trainer.distiller.pseudo_labeler
_orig_call = type(_pl).__call__
def _rec(self, weak, strong):
    c = _orig_call(self, weak, strong)
    cnt = c.pseudo["count"].tolist()
    recorded.append([{"image_size": c.sizes[i], "gt_boxes": c.pseudo["boxes"][i, :n].cpu().clone(), "gt_classes": c.pseudo["classes"][i, :n].to(torch.int64).cpu(),
                      "scores": c.pseudo["scores"][i, :n].cpu().clone()} for i, n in enumerate(cnt)])
    return c
type(_pl).__call__ = _rec
for it in range(STEPS):
    tr.iter = it
    tr.before_step(); tr.run_step(); tr.after_step()
    torch.cuda.synchronize()
    hip_losses.append({k: float(v) for k, v in tr._trainer.last_loss_dict.items()})
    print("hip step", it, {k: round(v, 5) for k, v in hip_losses[-1].items()}, "err", int(tr.model.engine.err))
    if it == 0:
        g0 = tr.model.weights.grad.clone()
        ti = tr.ema.model._last_inference
        hp = {k: v.clone().cpu() for k, v in ti.pseudo.items()}
        hdet = {k: v.clone().cpu() for k, v in ti.det.items()}
        hc = tr.model._last.ctx
        hroi = (hc.rois.cpu().clone(), hc.r_cls.cpu().clone(), hc.distill["labels"].cpu().clone(), hc.rpn_labels.cpu().clone())

# ---------------- oracle
sd = syn.init_state_dict(K, seed=1, img_da=ALIGN, ins_da=ALIGN)
ocfg = d2.make_cfg(num_classes=K)
align = None
if ALIGN:
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("img_align") or k.startswith("ins_align")}
    align = dict(img=True, ins=True, img_w=0.01, ins_w=0.01, params=params)
osd = {k: v for k, v in sd.items() if not (k.startswith("img_align") or k.startswith("ins_align"))}
orc = ao.OracleALDI(ocfg, osd, ema_alpha=0.9, lr=0.02, align=align, ims_per_gpu=2, backward_at_end=False, py_seed=0)
# mirror EMA semantics for discriminator params as part of the teacher state (buffers+params)
torch.manual_seed(123)
orc.pseudo_override = recorded
for it in range(STEPS):
    data = next(loader_for_oracle)
    t = time.time()
    ol = orc.step(*data)
    print("orc step", it, {k: round(v, 5) for k, v in ol.items()}, "%.1fs" % (time.time() - t))
    if it == 0:
        for n in range(2):
            pl = orc.last["pseudo_own"][n]
            cnt = int(hp["count"][n])
            print(f"   pseudo img{n}: hip {cnt} oracle {len(pl['scores'])}")
            m = min(cnt, len(pl["scores"]))
            print("     scores hip", hp["scores"][n, :m][:8].tolist())
            print("     scores orc", pl["scores"][:8].tolist())
            print("     box diff", float((hp["boxes"][n, :m] - pl["gt_boxes"][:m]).abs().max()) if m else None, "cls eq", bool((hp["classes"][n, :m] == pl["gt_classes"][:m]).all()))
        scap = orc.last["student_cap"]
        orois = torch.cat([s_["proposal_boxes"] for s_ in scap["sampled"]])
        print("   student rois equal:", tuple(hroi[0].shape), tuple(orois.shape), float((hroi[0][:, 1:] - orois).abs().max()) if hroi[0].shape[0] == orois.shape[0] else "shape")
        ocls = torch.cat([s_["gt_classes"] for s_ in scap["sampled"]])
        print("   student roi cls equal:", bool((hroi[1][:len(ocls)] == ocls.to(torch.int32)).all()))
        print("   rpn labels (student hard) equal:", bool((hroi[3] == torch.stack(scap["rpn_gt_labels"]).to(torch.int32)).all()))
        print("   rpn distill labels equal:", bool((hroi[2] == orc.last["rpn_distill_labels"].to(torch.int32)).all()),
              int((hroi[2] == 1).sum()), int((orc.last["rpn_distill_labels"] == 1).sum()))
    if it == 0:
        flat = torch.zeros(tr.model.layout.n_total); flat[:tr.model.layout.n_train] = g0.cpu()
        hg = tr.model.layout.unpack(flat)
        for k in ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_head.fc2.weight",
                  "proposal_generator.rpn_head.objectness_logits.weight", "proposal_generator.rpn_head.anchor_deltas.weight",
                  "proposal_generator.rpn_head.conv.weight", "backbone.fpn_output2.weight", "backbone.fpn_output4.weight", "backbone.bottom_up.res4.2.conv1.weight"):
            a, b = hg[k], orc.last["grads"][k]
            print(f"   grad0 {k:55s} max|d|={(a-b).abs().max():.3e} max|ref|={b.abs().max():.3e}")
        d_ = (hroi[0][:, 1:] - orois).abs().max(1)[0]
        print("   roi rows differing:", int((d_ > 1e-2).sum()), "first", (d_ > 1e-2).nonzero()[:5].flatten().tolist())
    worst = max(abs(ol[k] - hip_losses[it][k]) for k in ol)
    assert set(ol) == set(hip_losses[it]), (sorted(ol), sorted(hip_losses[it]))
    print("   max |loss diff| = %.3e" % worst)

def cmp(name, a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    e = (a - b).abs().max().item(); s = b.abs().max().item()
    print(f"  {name:44s} max|d|={e:.3e}  max|ref|={s:.3e}  rel={e / max(s, 1e-12):.2e}")

hs = tr.model.state_dict(); ht = tr.ema.model.state_dict()
for k in ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_head.fc1.weight", "proposal_generator.rpn_head.conv.weight",
          "backbone.fpn_output2.weight", "backbone.bottom_up.res5.0.conv1.weight", "backbone.bottom_up.res3.0.conv2.weight",
          "backbone.bottom_up.res2.0.conv1.weight", "backbone.bottom_up.res4.2.conv1.norm.running_var"):
    cmp("student " + k, hs[k], orc.sd[k])
    cmp("teacher " + k, ht[k], orc.teacher[k])
if ALIGN:
    for k in params:
        cmp("student " + k, hs[k], params[k])
