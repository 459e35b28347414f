import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def zk():
    """The product package (HIP library loaded; fails loudly if it is not built)."""
    import phase2_bn254_amd as pkg

    pkg.lib.load()
    return pkg


@pytest.fixture(scope="session")
def worker(zk):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return zk.Worker(0)
