import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
    return load


# Run order of the GPU suite: per-kernel parity files first, then the model-level parity files, then the long full-size /
# determinism property tests -- so that a red run still shows which kernels are good (pytest -x stops at the first failure).
_LATE = {"test_model_gpu.py": 1, "test_fullsize_gpu.py": 2, "test_fullsize_parity_gpu.py": 2, "test_determinism_gpu.py": 3}


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=lambda it: _LATE.get(os.path.basename(str(it.fspath)), 0))      # stable: file order otherwise unchanged
