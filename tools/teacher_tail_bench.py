"""the teacher's inference pass (2 images, 1333 x 800) alone on the chip, for a kernel trace: which of its latency-bound kernels (proposals, RoIAlign, box head,
detections) are long by themselves, and which only in the step (beside the student's one-workgroup-per-CU 3x3 kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aldi_amd import synthetic as syn
from aldi_amd.arch import ParamLayout
from aldi_amd.engine import RCNN, Weights
K = 8
lay = ParamLayout(K)
w = Weights(lay, torch.device("cuda"), torch.bfloat16, trainable=False)
w.load_state_dict(syn.init_state_dict(K, seed=1))
m = RCNN(w, K)
_, _, uw, _ = syn.make_batch(0, 2, 800, 1333, K, seed=3)
imgs = [d["image"].cuda() for d in uw]
for _ in range(int(os.environ.get("REPS", "6"))):
    t = m.inference(imgs, 0.8)
torch.cuda.synchronize()
print("detections per image", t.det.count.tolist())
