"""Measurement tooling and host-side scheduling helpers that the bench line and the fused step rely on (CPU only)."""
import csv
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _write_counter_csv(d, ctr, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "p_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (name, val) in enumerate(rows):
            w.writerow({"Dispatch_Id": i + 1, "Kernel_Name": name, "Counter_Name": ctr, "Counter_Value": val})


def test_pmc_traffic_classifier_counts_every_igemm_kernel(tmp_path, capsys):
    """`tools/rocprof_summary.py pmc` matched three kernel names and dropped `igemm_halo64_group_kernel` / `igemm_ws_kernel`
    (the round's two new kernels: 23 of 158 launches); the bench line then divided the partial byte total by all launches.  Every kernel of the
    family counts now."""
    import rocprof_summary as R
    step = [("stage_images_kernel", 0), ("stage_images_kernel", 0),
            ("void igemm_kernel<unsigned short, 128, 64>(ConvDev)", 100), ("igemm_halo64_group_kernel<256, 256, 4, 2, true, true>(ConvGroup)", 1000),
            ("igemm_ws_kernel<4, 2, 4>(ConvDev, int, int)", 10), ("igemm_group_kernel<...>", 1), ("splitk_finalize_kernel", 5),
            ("wgrad_bf16_big64_group_kernel<false>(WgGroup)", 7000), ("wgrad_finalize_kernel(WgFin)", 30), ("sgd_kernel<unsigned short>", 9)]
    rows = step * 3
    _write_counter_csv(str(tmp_path / "fetch"), "FETCH_SIZE", rows)
    _write_counter_csv(str(tmp_path / "write"), "WRITE_SIZE", rows)
    R.pmc(str(tmp_path / "fetch"), str(tmp_path / "write"), "abc", "def")
    out = json.loads(capsys.readouterr().out)
    assert out["igemm"]["kernel_launches_per_step"] == 5
    assert out["igemm"]["hbm_bytes_per_step"] == (2.0 * 1116 + 1116) * 1024.0
    assert out["wgrad"]["kernel_launches_per_step"] == 2
    assert out["source_sha256"] == "abc"


def test_bench_traffic_is_null_when_the_counter_file_misses_launches():
    sys.path.insert(0, ROOT)
    import bench
    tj = {"igemm": {"hbm_bytes_per_step": 1.0e9, "hbm_bytes_per_launch": 5.0e6, "kernel_launches_per_step": 138}}
    assert bench._traffic_per_launch(tj, "igemm", 158) is None              # r05's file: 138 of 158 launches
    tj["igemm"]["kernel_launches_per_step"] = 161
    assert bench._traffic_per_launch(tj, "igemm", 158) == round(1.0e9 / 158)
    assert bench._traffic_per_launch(None, "igemm", 158) is None


class _FakeTrainer:
    """the batch-fetching part of the trainer (`_fetch_batch`), with and without the fused step's look-ahead"""
    from aldi_amd.trainer import SimpleTrainer as _S
    _fetch_batch = _S._fetch_batch

    def __init__(self, it, ahead):
        self._data_loader_iter = iter(it)
        self._ahead = ahead

    def _wants_lookahead(self):
        return self._ahead


def test_batch_lookahead_keeps_the_order_and_ends_cleanly():
    """SOLVER.PIPELINE_PREFIX: the trainer fetches batch k + 1 before it runs step k (the reference's loader is a batch ahead anyway,
    /root/reference/aldi/trainer.py:211-238): same batches in the same order, the next one visible in `_next_batch`, a finite loader ends cleanly,
    and switching the look-ahead off mid-run does not drop the batch already fetched."""
    t = _FakeTrainer(range(5), True)
    seen = []
    for k in range(5):
        b = t._fetch_batch()
        seen.append((b, t._next_batch))
    assert seen == [(0, 1), (1, 2), (2, 3), (3, 4), (4, None)]
    t = _FakeTrainer(range(4), True)
    assert t._fetch_batch() == 0 and t._next_batch == 1
    t._ahead = False
    assert t._fetch_batch() == 1 and t._next_batch is None           # the batch fetched ahead is the next one, not skipped
    assert t._fetch_batch() == 2
    t = _FakeTrainer(range(3), False)
    assert [t._fetch_batch() for _ in range(3)] == [0, 1, 2] and t._next_batch is None
