import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import cpu
    cpu.build(ref=True)
    return cpu


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
