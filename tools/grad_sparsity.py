"""How sparse is d(loss)/d(P_l) -- the input of the FPN output convs' data / weight gradients -- in the benchmark step?  (ROIAlign's
backward touches only pixels under sampled ROIs, the sparse RPN-head backward <= 1024 pixels per image.)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aldi_amd import synthetic as syn
from aldi_amd.engine import RCNN
from aldi_amd.trainer import ALDITrainer
cfg = bench.make_cfg(1, 800, 1333, False)
cfg.SOLVER.FUSED_STEP = True
cfg.SOLVER.STEP_GRAPH = False
random.seed(1234); torch.manual_seed(100)
tr = ALDITrainer(cfg)
data = syn.make_batch(2, 2, 800, 1333, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=100)
tr._trainer.data_loader = bench.FixedGpuLoader(data, torch.device("cuda")); tr._trainer._data_loader_iter_obj = None
orig = RCNN._rpn_sparse_finish
def spy(self, c, gP_roi, sp):
    gP = orig(self, c, gP_roi, sp)
    torch.cuda.synchronize()
    for l, g in enumerate(gP):
        nz = (g != 0).any(-1)                                   # [N][H][W]
        N, H, W = nz.shape
        flat = nz.reshape(-1)
        line = "p%d: %7d px, nonzero %.3f" % (l + 2, flat.numel(), float(flat.float().mean()))
        for T in (64, 128, 256):
            # a tile of T consecutive pixels needs work if any pixel of the tile or of its 3x3 neighbourhood rows (dgrad) is nonzero
            dil = torch.nn.functional.max_pool2d(nz.float().unsqueeze(1), 3, 1, 1).squeeze(1).reshape(-1)
            pad = (-dil.numel()) % T
            d2 = torch.nn.functional.pad(dil, (0, pad)).view(-1, T).amax(1)
            f2 = torch.nn.functional.pad(flat.float(), (0, pad)).view(-1, T).amax(1)
            line += " | T=%d: dgrad tiles %.3f, wgrad slabs %.3f" % (T, float(d2.mean()), float(f2.mean()))
        print(line, flush=True)
    return gP
RCNN._rpn_sparse_finish = spy
for it in range(2):
    tr.iter = it; tr.before_step(); tr.run_step(); tr.after_step()
    print("---")
