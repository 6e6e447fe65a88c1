"""pytest configuration: the `gpu` marker, repo-root imports and shared fixtures."""
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    # The oracle (PyTorch CPU) is the checker of most GPU tests.  On the GPU box torch defaults to one thread per logical core
    # (128), where the oracle's small convolutions run ~3x slower than on 32 threads (tools/cpu_threads.py; bench.py's
    # cpu_baseline uses 32 for the same reason): the suite's wall time was mostly that.
    import os

    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
