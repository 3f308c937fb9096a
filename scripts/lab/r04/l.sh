#!/bin/bash
# r04l: SpMM with all workgroups on the long rows' chunks + chunk lists built once per graph: parity, then the cfg 5 line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04l
mkdir -p "$out"
timeout 900 python -m pytest tests/test_lightgcn_gpu.py tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_zz_ngcf_gpu.py tests/test_graph_fit_gpu.py "tests/test_feat_api_gpu.py::test_factorised_catalog_scores_equal_materialised_forward" -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest.log"
timeout 600 python bench.py --workload lightgcn --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_lightgcn.json" 2> "$out/bench_lightgcn.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04l/bench_lightgcn.json").read().strip().splitlines()[-1])
print("lightgcn ms", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mean_launch_ms"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
PY
