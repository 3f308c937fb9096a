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


@pytest.fixture
def f32_chain():
    """Pin the fused first layer to the exact f32 fma chain (csrc/deepfm_l1.hip) for tests of THOSE kernels; the default
    arithmetic where K = 64 / H1 = 128 is the split-bf16 form (csrc/deepfm_l1_sb.hip, tests/test_l1_split_bf16_gpu.py)."""
    from librecommender_amd import ops

    prev = ops.set_l1_arith("f32_chain")
    yield
    ops.set_l1_arith(prev)


@pytest.fixture(params=["split_bf16", "f32_chain"])
def l1_arith(request):
    """Run a model-level test under both arithmetics of the fused first layer."""
    from librecommender_amd import ops

    prev = ops.set_l1_arith(request.param)
    yield request.param
    ops.set_l1_arith(prev)
