#!/bin/bash
# r05aj: the new sharded-TwoTower ssl test on the HIP kernels + DIN step with / without the forked segment build
mkdir -p gpurun_out/r05aj
timeout 600 python -m pytest tests/test_dist_api_gpu.py -x -q -k "ssl or two_tower" 2>&1 | grep -v -i "rccl\|nccl" | tail -8 > gpurun_out/r05aj/tests.log
cat gpurun_out/r05aj/tests.log
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_din_fused_gpu.py tests/test_fm_models_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r05aj/tests_fold.log
cat gpurun_out/r05aj/tests_fold.log
for f in 1 0; do
  LIBRECO_DIN_FORK=$f timeout 300 python bench.py --workload din --steps 40 --warmup 10 --no-cpu-baseline --steady-seconds 1 > gpurun_out/r05aj/din_fork$f.json 2> gpurun_out/r05aj/din_fork$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05aj/din_fork$f.json").read().strip().splitlines()[-1])
print("fork=$f", d.get("ms_per_step"), d.get("steady_state"))
PY
done
