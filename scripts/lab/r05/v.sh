#!/bin/bash
# r05v: LightGCN with the batch-row bitmaps (last forward / first backward product)
out=gpurun_out/r05v; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_lightgcn_gpu.py tests/test_fullsize_cfg345_gpu.py -x -q -k "spmm or lightgcn or Lightgcn or fit or reference" 2>&1 | tail -6 | tee $out/pytest.log
timeout 600 python bench.py --workload lightgcn --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_lightgcn.json 2> $out/bench_lightgcn.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_lightgcn.json').read().strip().splitlines()[-1])
print('lightgcn ms/step', d['ms_per_step'], d['value'], d['kernels'])
PY
tail -3 $out/bench_lightgcn.err
