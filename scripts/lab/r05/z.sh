#!/bin/bash
out=gpurun_out/r05z0; mkdir -p $out
timeout 900 python -m pytest tests/test_bench_cli_gpu.py -x -q 2>&1 | tail -4 | tee $out/pytest.log
timeout 400 python bench.py --workload din --force-sharded --steps 10 --warmup 3 --no-cpu-baseline > $out/din_sharded.json 2> $out/din_sharded.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05z0/din_sharded.json').read().strip().splitlines()[-1])
print('din sharded w1', d['ms_per_step'], d['value'], d['config']['parallelism'][:60])
PY
tail -3 $out/din_sharded.err
