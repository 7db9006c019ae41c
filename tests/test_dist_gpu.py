"""The real data-parallel step on 2 ranks (VERDICT r01 missing #2 / SURVEY 8(e) "test without a cluster"): two processes
share cuda:0 and exchange gradients over gloo, each running `ALDITrainer` iterations with the fused schedule and the
overlapped `BucketedReducer`.  Checked:
  (a) identical student weights on both ranks after the steps;
  (b) world = 2 with per-rank batches == world = 1 processing both ranks' batches and averaging the gradients (fp32 mode);
  (c) the EMA teacher is bit-identical across ranks (it is never communicated: reference aldi/ema.py:19-27);
and `bench.py --gpus N` launches its own ranks (and refuses to run on fewer GPUs than asked).

Reference: DDP wrap aldi/dropin.py:53,84-85; per-rank loaders aldi/trainer.py:214-238."""
import json
import os
import random
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, H, W = 8, 192, 256
ITERS = 2


def _cfg(world, align):
    from aldi_amd.config import add_aldi_config, get_cfg
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-Cityscapes.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4 * world, "SOLVER.AMP.ENABLED", False, "SOLVER.BASE_LR", 0.002, "SOLVER.WARMUP_ITERS", 0,
                         "SEED", 1, "EMA.ALPHA", 0.9, "SYNTHETIC.HEIGHT", H, "SYNTHETIC.WIDTH", W,
                         "DOMAIN_ADAPT.ALIGN.IMG_DA_ENABLED", align, "DOMAIN_ADAPT.ALIGN.INS_DA_ENABLED", align])
    cfg.SOLVER.FUSED_STEP = True
    cfg.SOLVER.STEP_GRAPH = False
    return cfg


def _rank_data(rank, it):
    from aldi_amd import synthetic as syn
    return syn.make_batch(2, 2, H, W, K, seed=1000 * it + 31 * rank + 5)


class _ListLoader:
    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)


def _rank_main(rank, world, port, align, out_path):
    """one rank of the 2-process run (executed via `python tests/test_dist_gpu.py rank ...`)"""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd.trainer import ALDITrainer
    random.seed(1234)
    torch.manual_seed(9)
    tr = ALDITrainer(_cfg(world, align))
    t = tr._trainer
    t.data_loader = _ListLoader([_rank_data(rank, it) for it in range(ITERS)])
    t._data_loader_iter_obj = None
    random.seed(99)
    torch.manual_seed(500 + rank)
    fused, losses, first = [], [], None
    for it in range(ITERS):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        fused.append(bool(t._fused_done))
        losses.append({k: float(v) for k, v in t.last_loss_dict.items()})
        if it == 0:
            first = tr.model.weights.master.cpu()
    torch.cuda.synchronize()
    torch.save(dict(student=tr.model.weights.master.cpu(), teacher=tr.ema.model.weights.master.cpu(), fused=fused, losses=losses, student_it0=first,
                    err=int(tr.model.engine.err) | int(tr.ema.model.engine.err)), out_path)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(tmp_path, align):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs = [], []
    for r in range(2):
        outs.append(str(tmp_path / f"rank{r}.pt"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(r), "2", str(port), str(int(align)), outs[-1]],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), logs
    return [torch.load(o) for o in outs]


def _world1_emulation(align):
    """ONE process: both ranks' batches through the same fused step, gradients accumulated and averaged, one SGD.  Each rank's
    host RNG streams (python `random`, torch CPU generator, the distiller's ManualSeed) are kept separately and swapped in
    around that rank's pass, exactly as the two processes own them."""
    from aldi_amd.trainer import ALDITrainer
    random.seed(1234)
    torch.manual_seed(9)
    tr = ALDITrainer(_cfg(1, align))
    t = tr._trainer
    seeder = t.distiller.seeder
    st = []
    for r in range(2):
        random.seed(99)
        torch.manual_seed(500 + r)
        st.append(dict(py=random.getstate(), th=torch.get_rng_state(), seed=seeder.seed))
    losses = []
    for it in range(1):       # ONE iteration: it starts from identical weights, so every discrete decision (top-k, NMS, sampling)
        tr.iter = it          # is identical and only the fp32 summation order differs; later iterations may flip a near-tie
        tr.before_step()                                  # EMA tick
        t.optimizer.zero_grad()
        per_rank = []
        for r in range(2):
            random.setstate(st[r]["py"])
            torch.set_rng_state(st[r]["th"])
            seeder.seed = st[r]["seed"]
            ld = t.run_model(_rank_data(r, it))
            assert t._fused_done
            per_rank.append({k: float(v) for k, v in ld.items()})
            st[r] = dict(py=random.getstate(), th=torch.get_rng_state(), seed=seeder.seed)
        tr.model.weights.scale_grad(0.5)                  # mean over the two ranks
        t.optimizer.step()
        tr.after_step()
        losses.append(per_rank)
    torch.cuda.synchronize()
    return tr.model.weights.master.cpu(), tr.ema.model.weights.master.cpu(), losses, tr.model.layout


@pytest.mark.parametrize("align", [False, True])
def test_two_ranks_equal_one_rank_with_both_batches(tmp_path, align):
    res = _run_two_ranks(tmp_path, align)
    assert all(r["err"] == 0 for r in res)
    assert all(all(r["fused"]) for r in res), "the data-parallel step must take the fused + overlapped-exchange path"
    # (a) identical weights on every rank, (c) identical teacher
    assert torch.equal(res[0]["student"], res[1]["student"])
    assert torch.equal(res[0]["teacher"], res[1]["teacher"])
    # (b) == one rank that processes both batches and averages
    w1, t1, losses1, lay = _world1_emulation(align)
    for r in range(2):
        a, b = res[r]["losses"][0], losses1[0][r]
        assert list(a) == list(b)
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(b[k])), (r, k, a[k], b[k])
    w2 = res[0]["student_it0"]
    n = lay.n_train
    assert torch.equal(w2[n:], w1[n:])                                    # frozen part
    d = (w2[:n] - w1[:n]).abs().max().item()
    assert d <= 1e-6 * max(1.0, w1[:n].abs().max().item()), d             # fp32: atomics / summation order only
    upd = (w1[:n] - res[0]["student"][:n]).abs().max().item()            # (the second iteration moved them further)
    assert upd > 100 * d
    # and the step did move the weights
    from aldi_amd.trainer import ALDITrainer
    random.seed(1234)
    torch.manual_seed(9)
    w0 = ALDITrainer(_cfg(1, align)).model.weights.master.cpu()
    assert (w2[:n] - w0[:n]).abs().max().item() > 100 * max(d, 1e-9)


def _detr_cfg():
    from aldi_amd.config import add_aldi_config, get_cfg
    cfg = get_cfg()
    add_aldi_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cityscapes", "ALDI-Best-DETR-Cityscapes.yaml"))
    cfg.merge_from_list(["MODEL.DEFORMABLE_DETR.TRANSFORMER.NUM_QUERIES", 40, "MODEL.DEFORMABLE_DETR.TRANSFORMER.ENC_LAYERS", 2,
                         "MODEL.DEFORMABLE_DETR.TRANSFORMER.DEC_LAYERS", 2, "SEED", 3, "SOLVER.IMS_PER_BATCH", 8, "SOLVER.IMS_PER_GPU", 2, "SOLVER.WARMUP_ITERS", 0,
                         "SYNTHETIC.HEIGHT", 160, "SYNTHETIC.WIDTH", 224, "EMA.ALPHA", 0.9, "DOMAIN_ADAPT.TEACHER.THRESHOLD", 0.011, "SOLVER.BASE_LR", 1e-3])
    return cfg


def _detr_main(rank, world, port, out_path):
    """one rank of the Deformable-DETR data-parallel run: ALDITrainer iterations (HardDistiller), each rank on its own batches"""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aldi_amd import synthetic as syn
    from aldi_amd.detr.criterion import SetCriterion
    from aldi_amd.trainer import ALDITrainer
    # the set criterion's normaliser: the ranks' target counts summed / world size (3 and 5 targets -> 4 on both ranks)
    g = torch.Generator().manual_seed(40 + rank)
    n = 3 if rank == 0 else 5
    logits, boxes = torch.randn(2, 1, 30, 8, generator=g).cuda(), (torch.rand(2, 1, 30, 4, generator=g) * 0.4 + 0.2).cuda()
    tg = [{"labels": torch.randint(0, 8, (n,), generator=g), "boxes": torch.rand(n, 4, generator=g) * 0.4 + 0.2}]
    crit = SetCriterion()
    l_dp = {k: float(v) for k, v in crit(logits, boxes, tg)[0].items()}
    l_4 = {k: float(v) for k, v in crit(logits, boxes, tg, num_boxes=4.0)[0].items()}
    random.seed(1234)
    torch.manual_seed(9)
    tr = ALDITrainer(_detr_cfg())
    t = tr._trainer
    t.data_loader = _ListLoader([syn.make_batch(2, 2, 160, 224, 8, seed=1000 * it + 31 * rank + 5) for it in range(ITERS)])
    t._data_loader_iter_obj = None
    w0 = tr.model.weights.master.cpu()
    losses = []
    for it in range(ITERS):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        losses.append({k: float(v) for k, v in t.last_loss_dict.items()})
    torch.cuda.synchronize()
    # the backward's re-run of the loss kernel (gradient accumulation: scale 1 / accum != 1) divides by the normaliser the forward USED --
    # the world mean -- not by this rank's own target count
    from aldi_amd.detr.criterion import _world_mean
    batch = syn.make_batch(2, 2, 160, 224, 8, seed=77 + 13 * rank)[1]
    tr.model(batch)
    nb_local = float(sum(len(b["instances"]["gt_classes"] if isinstance(b["instances"], dict) else b["instances"].gt_classes) for b in batch))
    nb = dict(ctx=float(tr.model._last.ctx.num_boxes), local=max(nb_local, 1.0), world=_world_mean(nb_local, "cuda"))
    torch.save(dict(student=tr.model.weights.master.cpu(), teacher=tr.ema.model.weights.master.cpu(), initial=w0, losses=losses, l_dp=l_dp, l_4=l_4, nb=nb), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_deformable_detr_two_ranks(tmp_path):
    """BASELINE configs[4] under data parallelism (reference: DDP wrap aldi/dropin.py:53): two ranks with different batches end every
    iteration with bit-identical student weights (one all-reduce of the flat gradient before the clip and the AdamW step) and
    bit-identical teachers; the criterion normalises by the world's mean target count"""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs = [], []
    for r in range(2):
        outs.append(str(tmp_path / f"detr{r}.pt"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "detr", str(r), "2", str(port), outs[-1]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), logs
    res = [torch.load(o) for o in outs]
    assert torch.equal(res[0]["student"], res[1]["student"]) and torch.equal(res[0]["teacher"], res[1]["teacher"])
    assert torch.equal(res[0]["initial"], res[1]["initial"]) and not torch.equal(res[0]["student"], res[0]["initial"])
    assert res[0]["losses"] != res[1]["losses"]                     # different batches per rank
    assert all(v == v and abs(v) < 1e4 for r in res for d in r["losses"] for v in d.values())
    assert all(r["nb"]["ctx"] == r["nb"]["world"] for r in res) and res[0]["nb"]["world"] == res[1]["nb"]["world"], [r["nb"] for r in res]
    assert any(r["nb"]["local"] != r["nb"]["world"] for r in res), [r["nb"] for r in res]       # (the two ranks' batches hold different counts)
    for r in res:
        assert r["l_dp"].keys() == r["l_4"].keys()
        for k in r["l_dp"]:
            assert abs(r["l_dp"][k] - r["l_4"][k]) <= 1e-6 * max(1.0, abs(r["l_4"][k])), (k, r["l_dp"][k], r["l_4"][k])


def _bench(args, extra_env=None, timeout=900):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return p.returncode, p.stdout.decode(errors="replace")


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher) spawns 2 ranks; on this 1-GPU box they share cuda:0 over gloo through the test hook"""
    rc, out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--height", str(H), "--width", str(W), "--no-cpu-baseline", "--no-profile"],
                     {"ALDI_BENCH_BACKEND": "gloo", "ALDI_BENCH_DEVICE": "0"})
    assert rc == 0, out[-3000:]
    line = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(line) == 1, out[-3000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 8 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    assert j["config"]["error_flag"] == 0 and j["value"] > 0


def test_bench_refuses_fewer_gpus_than_asked():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than requested")
    rc, out = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--height", str(H), "--width", str(W), "--no-cpu-baseline", "--no-profile"])
    assert rc != 0 and '"metric"' not in out, out[-2000:]


def _rccl_main(mode, port, payload, out_path):
    """six iterations of the fused + graph-replayed step in ONE process: mode "rccl" = a world-size-1 process group on the REAL
    backend (torch "nccl" = RCCL) with ALDI_DP_FORCE=1, i.e. the whole data-parallel code path -- bucketed exchange on its launch
    stream, producer events, collectives recorded into the phase-B graph -- with RCCL doing the (single-rank) all-reduces;
    mode "plain" = no process group."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    if mode == "rccl":
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ALDI_DP_FORCE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from aldi_amd.trainer import ALDITrainer
    cfg = _cfg(1, False)
    cfg.SOLVER.STEP_GRAPH = True
    payload, _, exchange = payload.partition("+")
    cfg.SOLVER.GRAD_PAYLOAD = payload
    if exchange:
        cfg.SOLVER.GRAD_EXCHANGE = exchange
    random.seed(1234)
    torch.manual_seed(9)
    tr = ALDITrainer(cfg)
    t = tr._trainer
    n_it = 6
    t.data_loader = _ListLoader([_rank_data(0, it % 2) for it in range(n_it)])
    t._data_loader_iter_obj = None
    losses, w0, first = [], tr.model.weights.master.cpu(), None
    for it in range(n_it):
        tr.iter = it
        tr.before_step()
        tr.run_step()
        tr.after_step()
        losses.append({k: float(v) for k, v in t.last_loss_dict.items()})
        if it == 0:
            first = tr.model.weights.master.cpu()
    torch.cuda.synchronize()
    stats = dict(t._fused_step.stats)
    torch.save(dict(student=tr.model.weights.master.cpu(), student_it0=first, initial=w0, losses=losses, stats=stats, dp_graph_ok=bool(t._fused_step.dp_graph_ok),
                    backend=(dist.get_backend() if mode == "rccl" else None)), out_path)
    if mode == "rccl":
        dist.destroy_process_group()


def _run_single(tmp_path, mode, payload="fp32"):
    out = str(tmp_path / f"{mode}_{payload}.pt")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "rccl", mode, str(_free_port()), payload, out],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    return torch.load(out)


def test_rccl_exchange_inside_the_phase_b_graph(tmp_path):
    """RCCL is initialised and drives the gradient exchange of the REAL step on this 1-GPU box (world size 1, data-parallel path
    forced): the collectives are recorded into the phase-B hipGraph and replayed (`replays_b_dp`), and with one rank the exchange
    is the identity, so the run must equal the same run without a process group -- fp32 payload: weights bit for bit (same kernels,
    same graphs); bf16 payload: gradients rounded to bf16 once, so close but not equal."""
    plain = _run_single(tmp_path, "plain")
    rccl = _run_single(tmp_path, "rccl")
    assert rccl["backend"] == "nccl"
    st = rccl["stats"]
    assert st["replays_a"] >= 2 and st["replays_b"] >= 2, st
    assert rccl["dp_graph_ok"] and st.get("replays_b_dp", 0) == st["replays_b"], ("phase B was not replayed with its collectives", st)
    # the first iteration starts from identical state: everything but the fp32 summation order of the weight-gradient atomics is
    # identical; later iterations may flip a near-tie of a discrete stage (as in the two-rank test above)
    a, b = plain["losses"][0], rccl["losses"][0]
    assert list(a) == list(b)
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), (k, a[k], b[k])
    step = (plain["student_it0"] - plain["initial"]).abs().max().item()
    d = (plain["student_it0"] - rccl["student_it0"]).abs().max().item()
    assert step > 0 and d <= 1e-3 * step, (d, step)
    for a, b in zip(plain["losses"], rccl["losses"]):              # (random-init training at lr 2e-3: trajectories separate after a flip)
        assert list(a) == list(b) and all(v == v and abs(v) < 1e4 for v in b.values()), (a, b)
    half = _run_single(tmp_path, "rccl", "bf16")
    assert half["stats"].get("replays_b_dp", 0) >= 2
    moved = (plain["student_it0"] - half["student_it0"]).abs().max().item()
    assert d < moved <= 2e-2 * step, (moved, d, step)              # gradients rounded to bf16 once: visible, and small against the update
    # SOLVER.GRAD_EXCHANGE "rs_ag": the same exchange as RCCL reduce_scatter_tensor + all_gather_into_tensor per bucket (in place on the
    # flat gradient), recorded and replayed inside phase B as well; one rank: the identity again
    rsag = _run_single(tmp_path, "rccl", "fp32+rs_ag")
    assert rsag["dp_graph_ok"] and rsag["stats"].get("replays_b_dp", 0) >= 2, rsag["stats"]
    d2 = (plain["student_it0"] - rsag["student_it0"]).abs().max().item()
    assert d2 <= 1e-3 * step, (d2, step)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rank":
    _rank_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5])), sys.argv[6])
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rccl":
    _rccl_main(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "detr":
    _detr_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
