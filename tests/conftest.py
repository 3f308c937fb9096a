import os
import sys
from pathlib import Path

import pytest

# The multi-process tests start 2-4 rank processes of a few seconds of work each: with the default (one OpenMP thread per
# core in EVERY process) they only fight over the cores — `tests/test_dist_api_cpu.py` took 333 s with it and takes 105 s with
# two threads per process.  Inherited by the spawned ranks; an explicit setting of the caller wins.
os.environ.setdefault("OMP_NUM_THREADS", "2")

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
