"""The reference's OWN test files, run unmodified against this package.

Only in the build container (the reference checkout is read-only at /root/reference and does not travel): the files are
copied to a scratch directory outside the repository, `libreco` is aliased to `librecommender_amd` by a `sys.meta_path`
finder in a scratch `conftest.py`, and pytest runs them in a subprocess.  Covered: the host-side files whose imports stay
inside the seam of SURVEY 8(b) (`tests/test_data.py`, `test_split_data.py`, `test_misc.py`, `test_consumed.py`).  Files that import the reference's
private helpers or out-of-scope models are covered by transcribed known-answer tests instead
(`tests/test_reference_data_kat_cpu.py`); `tests/test_rank_reco.py` needs the device (`tests/test_rank_seam_gpu.py`)."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

REF = Path(os.environ.get("LIBRECO_REFERENCE", "/root/reference"))
REPO = Path(__file__).resolve().parent.parent
FILES = ["test_data.py", "test_split_data.py", "test_misc.py", "test_consumed.py"]

CONFTEST = '''
import importlib, importlib.abc, importlib.util, sys
sys.path.insert(0, {repo!r})


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`import libreco[.x.y]` -> the module object of `librecommender_amd[.x.y]`."""

    def find_spec(self, name, path, target=None):
        if name == "libreco" or name.startswith("libreco."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("librecommender_amd" + spec.name[len("libreco"):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Alias())
'''


@pytest.mark.skipif(not (REF / "tests" / "test_data.py").exists(), reason="reference checkout not present (GPU box)")
def test_reference_host_test_files_pass_against_this_package(tmp_path):
    tests = tmp_path / "tests"
    tests.mkdir()
    for name in ["__init__.py", "utils_data.py", *FILES]:
        shutil.copy(REF / "tests" / name, tests / name)
    shutil.copytree(REF / "tests" / "sample_data", tests / "sample_data")
    (tmp_path / "conftest.py").write_text(CONFTEST.format(repo=str(REPO)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", *[f"tests/{f}" for f in FILES]],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
