import os
import sys

import pytest

# The oracle (OpenMP) and torch size their thread pools from the logical CPUs they SEE; a GPU box shows all host cores but
# grants a 16-CPU share, and an oversubscribed, spinning OpenMP team made this suite take 12 minutes instead of one.
_NT = str(max(1, min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_NUM_THREADS", _NT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("MKL_NUM_THREADS", _NT)

# (no UMX_HIP_RUNTIME here: libumx's default binding -- the runtime an installed PyTorch bundles, located without importing torch --
# is what the test processes run on, whatever order they import torch and umx in; tests/test_host_logic.py covers the policy)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
