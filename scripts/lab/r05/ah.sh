#!/bin/bash
out=gpurun_out/r05ah; mkdir -p $out
for v in default nont; do
  lib=$PWD/build/lab/libreco_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  LIBRECO_HIP_LIB=$lib timeout 400 python bench.py --workload lightgcn --steps 8 --warmup 2 --no-cpu-baseline --steady-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k:v['mean_ms'] for k,v in d['kernels'].items() if 'spmm' in k})"
done | tee $out/variants.log
timeout 300 python -m pytest tests/test_lightgcn_gpu.py -q 2>&1 | grep -E "passed|failed"
