"""N>1 path on CPU: world_size-2 gloo run of the one-all-reduce-per-step gradient exchange
(SimpleTrainer.after_backward) -- identical averaged gradients on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _W:
    def __init__(self, rank):
        self.grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        self._gscale = 1.0

    def scale_grad(self, f):
        self._gscale *= f


class _M:
    training = True

    def __init__(self, rank):
        self.weights = _W(rank)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.trainer import SimpleTrainer
    t = SimpleTrainer(_M(rank), None, None)
    t.after_backward()
    g = t.model.weights.grad * t.model.weights._gscale
    q.put((rank, g[:5].tolist(), float(g.sum())))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    ref = torch.arange(1000, dtype=torch.float32) * 1.5          # mean of 1x and 2x
    for rank, head, total in res:
        assert head == ref[:5].tolist()
        assert abs(total - float(ref.sum())) < 1e-2


def _worker_bucketed(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.reduce import BucketedReducer
    n = 3_000_000
    g = torch.Generator().manual_seed(7 + rank)
    grad = torch.randn(n, generator=g)
    ref = grad.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    red = BucketedReducer(grad)
    # reports arrive out of order, overlap, include pieces below the launch threshold, and leave gaps for finish()
    red.ready([(2_000_000, 2_900_000), (2_900_000, 2_900_100)])
    red.ready([(100, 200)])
    red.ready([(500_000, 1_200_000), (1_100_000, 1_500_000), (200, 300_000)])
    red.finish()
    red.finish()                                                       # idempotent (fused step + after_backward both call it)
    exact = bool(torch.equal(grad, ref))
    # bf16 payload (SOLVER.GRAD_PAYLOAD): every piece is rounded to bf16 once, summed, written back into the fp32 buffer
    grad2 = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank))
    red2 = BucketedReducer(grad2, payload="bf16")
    red2.ready([(2_000_000, 2_900_000)])
    red2.finish()
    rel = float((grad2 - ref).abs().max() / ref.abs().max())
    # SOLVER.GRAD_EXCHANGE "rs_ag": every piece as reduce_scatter_tensor + all_gather_into_tensor (odd lengths: the tail shorter than
    # the world size goes through all_reduce) == the all-reduce of the whole buffer; two ranks: one addition per element, so bit-exact
    grad3 = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank))
    red3 = BucketedReducer(grad3, exchange="rs_ag")
    red3.ready([(2_000_000, 2_900_001), (2_900_001, 2_900_100)])
    red3.ready([(101, 200)])
    red3.ready([(500_000, 1_200_000), (1_100_000, 1_500_003), (200, 300_000)])
    red3.finish()
    exact = exact and bool(torch.equal(grad3, ref))
    grad4 = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank))
    red4 = BucketedReducer(grad4, payload="bf16", exchange="rs_ag")
    red4.ready([(2_000_000, 2_900_001)])
    red4.finish()
    rel4 = float((grad4 - ref).abs().max() / ref.abs().max())
    q.put((rank, exact and 0 < rel < 2e-2 and 0 < rel4 < 2e-2, float((grad - ref).abs().max()) + rel))
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_world2_gloo():
    """the fused step's overlapped exchange == one all-reduce of the whole buffer (every element reduced exactly once)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_bucketed, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    for rank, same, err in res:
        assert same, (rank, err)


def _worker_rs_ag8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.reduce import BucketedReducer
    n = 400_003
    grad = torch.randn(n, generator=torch.Generator().manual_seed(11 + rank))
    ref = grad.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    red = BucketedReducer(grad, exchange="rs_ag")
    red.ready([(300_000, 390_001), (390_001, 390_006)])            # lengths that are not multiples of 8; a piece shorter than the world size
    red.ready([(13, 100_000)])
    red.ready([(100_000, 250_007), (240_000, 300_000)])
    red.finish()
    err = float((grad - ref).abs().max() / ref.abs().max())
    same_everywhere = grad.clone()
    dist.broadcast(same_everywhere, src=0)
    q.put((rank, err, bool(torch.equal(same_everywhere, grad))))
    dist.destroy_process_group()


def test_bucketed_rs_ag_world8_gloo():
    """SOLVER.GRAD_EXCHANGE "auto" picks reduce-scatter + all-gather on the 8 ranks of one node: the exchange at THAT world size -- ragged
    pieces, a piece shorter than the world size, overlapping reports -- equals the all-reduce of the whole buffer (eight addends: fp32 sum
    order may differ) and leaves every rank with the same bits"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_rs_ag8, args=(r, 8, port, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(8))
    for p in ps:
        p.join(timeout=60)
    for rank, err, same in res:
        assert err <= 2e-6 and same, (rank, err, same)


def test_range_helpers():
    from aldi_amd.reduce import complement, merge_ranges
    assert merge_ranges([(5, 9), (0, 3), (3, 4), (8, 12), (20, 20)]) == [(0, 4), (5, 12)]
    assert complement([(5, 9), (0, 3)], 12) == [(3, 5), (9, 12)]
    assert complement([], 4) == [(0, 4)]
    assert complement([(0, 4)], 4) == []


def _count_collectives(n, reports):
    """run BucketedReducer's bookkeeping over a sequence of ready() reports with the collective stubbed out"""
    from aldi_amd.reduce import BucketedReducer

    class Counting(BucketedReducer):
        def __init__(self, n_):
            self.n_, self.done, self.pending, self.works, self.sizes = n_, [], [], [], []
            self.launch_stream, self.events = None, []

        def _launch(self, lo, hi):
            self.sizes.append(hi - lo)
            self.done.append((lo, hi))

        def finish(self):
            from aldi_amd.reduce import complement
            for lo, hi in complement(self.done, self.n_):
                self._launch(lo, hi)
    r = Counting(n)
    for rep in reports:
        r.ready(rep)
    r.finish()
    return r.sizes


def test_exchange_launches_few_large_collectives():
    """the overlapped exchange must not dissolve into hundreds of latency-bound collectives: reported ranges carry their layout
    padding, so each layer group is ONE contiguous range and finish() has only the never-reported remainder left"""
    from aldi_amd.arch import STAGE_BLOCKS, ParamLayout
    lay = ParamLayout(8)
    bu = "backbone.bottom_up."
    groups = [["box_pred", "roi_heads.box_head.fc2", "roi_heads.box_head.fc1"],
              ["rpn_head_out", "proposal_generator.rpn_head.conv"] + [f"backbone.fpn_output{l}" for l in (2, 3, 4, 5)] + [f"backbone.fpn_lateral{l}" for l in (2, 3, 4, 5)]]
    for si in (3, 2, 1):
        names = []
        for b in range(STAGE_BLOCKS[si]):
            p = f"{bu}res{si + 2}.{b}."
            names += [p + "conv3", p + "conv2", p + "conv1"] + ([p + "shortcut"] if b == 0 else [])
        groups.append(names)
    sizes = _count_collectives(lay.n_train, [lay.ranges(g) for g in groups])
    assert sum(sizes) == lay.n_train and len(sizes) <= 12, (len(sizes), sorted(sizes)[:10])
    assert min(sizes) >= 4096 or len(sizes) <= 8

    from aldi_amd.vit import VitConfig, VitParams
    cfg = VitConfig(sfp=True, num_classes=8)
    P = VitParams(cfg, "cpu")
    reports = [P.ranges([n for n in P.spec if n.startswith(("proposal_generator.", "roi_heads."))]),
               P.ranges([n for n in P.spec if n.startswith("backbone.simfp_")])]
    for i in reversed(range(cfg.depth)):
        reports.append(P.ranges([n for n in P.spec if n.startswith(f"{cfg.prefix}blocks.{i}.")]))
    reports.append(P.ranges([cfg.prefix + "pos_embed", cfg.prefix + "patch_embed.proj.weight", cfg.prefix + "patch_embed.proj.bias"]))
    sizes = _count_collectives(P.n, reports)
    assert sum(sizes) == P.n and len(sizes) <= 20, (len(sizes), sorted(sizes)[:10])


# ---- distributed evaluation: the test set is sharded by rank and gathered once (detections were counted world times)
class _FakeDetector:
    """stand-in for the EMA model in ALDITrainer.test: noisy ground truth + false positives, a pure function of the image id"""
    training = False

    def __init__(self, records):
        self.records = {r["image_id"]: r for r in records}

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def __call__(self, inputs):
        from aldi_amd.structures import Boxes, Instances
        out = []
        for inp in inputs:
            r = self.records[inp["image_id"]]
            g = torch.Generator().manual_seed(100 + inp["image_id"])
            b = torch.tensor([a["bbox"] for a in r["annotations"]], dtype=torch.float32).reshape(-1, 4)
            c = torch.tensor([a["category_id"] for a in r["annotations"]], dtype=torch.int64)
            keep = torch.rand(len(b), generator=g) > 0.3
            b, c = b[keep] + torch.randn(int(keep.sum()), 4, generator=g) * 3.0, c[keep]
            fp = torch.rand(4, 4, generator=g) * 100
            fp[:, 2:] += fp[:, :2] + 20
            boxes = torch.cat([b, fp, b[:2] + 1.0])                 # duplicates of matched GT must count as false positives ONCE
            cls = torch.cat([c, torch.randint(0, 8, (4,), generator=g), c[:2]])
            sc = torch.rand(len(boxes), generator=g)
            out.append({"instances": Instances((inp["height"], inp["width"]), pred_boxes=Boxes(boxes), scores=sc, pred_classes=cls)})
        return out


def _eval_cfg():
    from aldi_amd.config import add_aldi_config, get_cfg
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_list(["MODEL.ROI_HEADS.NUM_CLASSES", 8, "SYNTHETIC.HEIGHT", 96, "SYNTHETIC.WIDTH", 128, "SYNTHETIC.VAL_IMAGES", 7])
    return cfg


def _worker_eval(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.trainer import ALDITrainer
    cfg = _eval_cfg()
    _, records = ALDITrainer.build_test_loader(cfg, "synthetic_val")
    res = ALDITrainer.test(cfg, _FakeDetector(records))
    q.put((rank, dict(res.get("bbox", {})) if res else {}))
    dist.destroy_process_group()


def test_distributed_evaluation_equals_single_process():
    from aldi_amd.trainer import ALDITrainer
    cfg = _eval_cfg()
    _, records = ALDITrainer.build_test_loader(cfg, "synthetic_val")
    ref = ALDITrainer.test(cfg, _FakeDetector(records))["bbox"]
    assert 0.0 < ref["AP50"] < 100.0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_eval, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert res[1] == {}                                           # only the main process reports
    for k in ("AP", "AP50", "AP75"):
        assert abs(res[0][k] - ref[k]) < 1e-9, (k, res[0][k], ref[k])


def _worker_detr_norm(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.detr.criterion import _world_mean
    dev = torch.device("cpu")
    q.put((rank, _world_mean(3.0 if rank == 0 else 5.0, dev), _world_mean(0.0, dev), _world_mean(1.0 if rank == 0 else 0.0, dev)))
    dist.destroy_process_group()


def test_detr_target_count_is_the_world_mean_gloo():
    """the Deformable-DETR set criterion's normaliser under data parallelism: the ranks' target counts summed / world size, at least 1
    (3 and 5 targets -> 4 on both ranks; none anywhere -> 1; one target on one rank -> 0.5 -> 1)"""
    from aldi_amd.detr.criterion import _world_mean
    assert _world_mean(7.0, torch.device("cpu")) == 7.0 and _world_mean(0.0, torch.device("cpu")) == 1.0       # no process group: this rank's count
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_detr_norm, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert [r[1:] for r in res] == [(4.0, 1.0, 1.0), (4.0, 1.0, 1.0)], res
