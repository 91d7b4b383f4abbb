import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-generation-models_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _seeded():
    """Every test starts from the same generator state (CPU and, when there is one, the GPU): a test that draws an operand without seeding
    it sees the same operand on every box and in every order of execution, so a tolerance that holds once holds on the driver's run."""
    import torch
    torch.manual_seed(20240905)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(20240905)
    yield
