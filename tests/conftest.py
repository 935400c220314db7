import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is the slow side of every parity test.  On the GPU box's host (128+ cores) torch's default of one thread per core makes
    # it ~3x SLOWER than 32 threads (profiles/r03_bench.json cpu_baseline: 9.0 s at 32, 14.3 s at 64, 26.2 s at all cores per bs=8 pass).
    try:
        import torch
        if (os.cpu_count() or 1) > 32 and "OMP_NUM_THREADS" not in os.environ:
            torch.set_num_threads(32)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
