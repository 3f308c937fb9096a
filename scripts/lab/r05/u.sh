#!/bin/bash
# r05u: dense path after the streaming rewrite of the table pass + the test files it touches
out=gpurun_out/r05u; mkdir -p $out
timeout 900 python -m pytest tests/test_dense_adam_fused_gpu.py tests/test_fm_models_gpu.py tests/test_deepfm_fused_gpu.py tests/test_cfg1_movielens_gpu.py -x -q 2>&1 | tail -6 | tee $out/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --no-workloads --steady-seconds 0 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05u/bench.json').read().strip().splitlines()[-1])
print('deepfm ms/step', d['ms_per_step'], 'f32 chain', d.get('f32_chain_ms_per_step'))
da=d.get('dense_adam'); print(da['ms_per_step'], da.get('kernels'), da.get('roofline',{}).get('frac'))
PY
