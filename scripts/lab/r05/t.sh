#!/bin/bash
# r05t: TF1 dense Adam on the fused step: tests + the bench line
out=gpurun_out/r05t; mkdir -p $out
timeout 900 python -m pytest tests/test_dense_adam_fused_gpu.py tests/test_fm_models_gpu.py tests/test_deepfm_fused_gpu.py -x -q 2>&1 | tail -12 | tee $out/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --no-workloads --steady-seconds 0 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05t/bench.json').read().strip().splitlines()[-1])
print('deepfm ms/step', d['ms_per_step'], 'f32 chain', d.get('f32_chain_ms_per_step'))
print(json.dumps(d.get('dense_adam'), indent=1)[:2500])
print(json.dumps(d['roofline'], indent=1)[:1500])
PY
tail -5 $out/bench.err
