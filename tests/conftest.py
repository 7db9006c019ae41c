import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """the CPU oracle's convolutions (torch-CPU, small batches) stop scaling around 32 threads and slow down by an order of magnitude
    when every hardware thread of a 256-thread host is used (measured in bench.py's cpu_baseline): bound them for the whole session"""
    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
