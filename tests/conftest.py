import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """HIP device for `-m gpu` tests.  No silent skipping: a gpu-marked test without a device fails."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran on a machine without a HIP device")
    from librecommender_amd import _lib

    _lib.load()  # raises HipExtensionMissing if the extension was not built
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
