import os
import sys

import pytest

# The Glow-stack tests run PyTorch-ROCm convolutions around the unit; MIOpen's default exhaustive find costs
# minutes per new shape on a fresh box.  Immediate-mode heuristics are enough for a parity test.
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
