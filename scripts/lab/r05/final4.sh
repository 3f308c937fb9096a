#!/bin/bash
# r05 final (fourth box, after the chunked sharded LightGCN): the full GPU suite and the driver's command once more on the last commit
out=gpurun_out/r05final4; mkdir -p $out
timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $out/pytest_gpu_summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05final4/bench_default.json').read().strip().splitlines()[-1])
print('deepfm', d['ms_per_step'], d['value'], 'f32 chain', d.get('f32_chain_ms_per_step'), 'steady', d.get('steady_ms_per_step'))
print('recommend', d['recommend'].get('ms_per_pass'), 'dense_adam', d['dense_adam'].get('ms_per_step'))
for k,v in d['workloads'].items(): print(k, v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('frac_from_profiles'), v.get('error'))
PY
