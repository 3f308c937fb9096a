#!/bin/bash
out=gpurun_out/r05ac; mkdir -p $out
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_dist_api_gpu.py tests/test_bench_cli_gpu.py tests/test_lightgcn_gpu.py -x -q 2>&1 | tail -5 | tee $out/pytest.log
timeout 600 python bench.py --workload lightgcn --force-sharded --steps 6 --warmup 2 --no-cpu-baseline > $out/lgcn_sharded.json 2> $out/lgcn_sharded.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ac/lgcn_sharded.json').read().strip().splitlines()[-1])
print('lightgcn sharded w1', d['ms_per_step'], d['value'], d.get('kernels'))
PY
tail -2 $out/lgcn_sharded.err
