#!/bin/bash
# r05 final: the full GPU suite, smoke(), the driver's command and the row-sharded lines at world size 1 on the final tree
out=gpurun_out/r05final; mkdir -p $out
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
timeout 600 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend > $out/bench_force_sharded.json 2> $out/bench_force_sharded.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05final/bench_default.json').read().strip().splitlines()[-1])
print('deepfm', d['ms_per_step'], d['value'], 'f32 chain', d.get('f32_chain_ms_per_step'), 'steady', d.get('steady_ms_per_step'), d['dtype'])
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','frac','mean_launch_ms','profiles_avg_ms','frac_from_profiles','frac_dedup_min','frac_by_traffic')})
print('kernels', {k:v['mean_ms'] for k,v in d['kernels'].items() if v['mean_ms']>0.05})
print('recommend', d['recommend'].get('ms_per_pass'), d['recommend'].get('roofline',{}).get('frac'), d['recommend'].get('roofline',{}).get('frac_from_profiles'))
print('dense_adam', d['dense_adam'].get('ms_per_step'), d['dense_adam'].get('roofline',{}).get('frac'), d['dense_adam'].get('roofline',{}).get('frac_from_profiles'))
for k,v in d['workloads'].items():
    print(k, v.get('ms_per_step'), v.get('value'), v.get('f32_chain_ms_per_step'), {q:v.get('roofline',{}).get(q) for q in ('kernel','frac','frac_from_profiles','mean_launch_ms','profiles_avg_ms')}, v.get('error'))
print('cpu_baseline', d.get('cpu_baseline',{}).get('value'))
f=json.loads(open('gpurun_out/r05final/bench_force_sharded.json').read().strip().splitlines()[-1])
print('force-sharded deepfm', f['ms_per_step'], f['value'])
PY
