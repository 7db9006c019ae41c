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
