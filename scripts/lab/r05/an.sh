#!/bin/bash
# r05an: column-blocked / chunked sharded LightGCN on the HIP kernels + SpMM tests + the single-GPU LightGCN line
mkdir -p gpurun_out/r05an
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_ops_gpu.py tests/test_lightgcn_gpu.py tests/test_graph_fit_gpu.py -x -q -k "lightgcn or spmm or graph" 2>&1 | grep -E "passed|failed|error|Error" | tail -6 > gpurun_out/r05an/tests.log
cat gpurun_out/r05an/tests.log
timeout 900 python -m pytest tests/test_dist_api_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -4 > gpurun_out/r05an/tests_dist.log
cat gpurun_out/r05an/tests_dist.log
timeout 600 python bench.py --workload lightgcn --steps 5 --warmup 2 --no-cpu-baseline --steady-seconds 0 > gpurun_out/r05an/lgcn.json 2> gpurun_out/r05an/lgcn.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05an/lgcn.json").read().strip().splitlines()[-1])
print("lightgcn", d.get("ms_per_step"), {k:v.get("mean_ms") for k,v in d.get("kernels",{}).items()})
PY
