#!/bin/bash
# r05n: split-bf16 softmax-CE — parity (both arithmetics), the no-transposing-read variant if it fails, kernel bench, TwoTower line
out=gpurun_out/r05n; mkdir -p $out
timeout 600 python -m pytest tests/test_softmax_ce_gpu.py -x -q -s 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log
if ! grep -q " passed" $out/pytest.log || grep -q "failed" $out/pytest.log; then
  echo "== no-TR variant"
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sce_notr.so timeout 600 python -m pytest tests/test_softmax_ce_gpu.py -x -q 2>&1 | tail -8 | tee $out/pytest_notr.log
fi
timeout 300 python scripts/sce_bench.py 65536 128 5 2>&1 | tee $out/sce_bench.log
timeout 400 python bench.py --workload twotower --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_twotower.json 2> $out/bench_twotower.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05n/bench_twotower.json').read().strip().splitlines()[-1])
print('twotower ms/step', d['ms_per_step'], d['value'], {k:v['mean_ms'] for k,v in d.get('kernels',{}).items()})
print(d['roofline'])
PY
