"""The driver's multi-GPU command line, end to end, on the one GPU of a test box: `bench.py --gpus 2` re-launches itself
as two ranks (127.0.0.1 rendezvous), which with `--backend gloo` share device 0 — the row-sharded DeepFM step, the
sharded recommend leg, the max-over-ranks timing and the single JSON line on stdout are the ones an 8-GPU run uses; only the
transport differs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--small", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--steady-seconds", "0", *extra]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]            # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_two_ranks_deepfm_line():
    r = _run([])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["scaling"] == "weak"
    assert "row-sharded" in r["config"]["parallelism"]
    assert r["config"]["global_batch"] == 2 * r["config"]["per_gpu_batch"]
    assert "error" not in (r.get("recommend") or {})


@pytest.mark.parametrize("workload", ["twotower", "lightgcn", "din"])
def test_two_ranks_other_sharded_workloads(workload):
    r = _run(["--workload", workload])
    assert r["n_gpus"] == 2 and r["value"] > 0
