#!/bin/bash
out=gpurun_out/r05ag; mkdir -p $out
for v in default din_w3 din_w4; do
  lib=$PWD/build/lab/libreco_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  LIBRECO_HIP_LIB=$lib timeout 200 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['kernels']['lr_din_attn_pool_bwd_parts_f32'])"
done | tee $out/variants.log
