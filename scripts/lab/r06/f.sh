#!/bin/bash
# r06f: DIN with the backward reading the forward's saved hidden activations: tests, then the step with the product build
# (2 waves / SIMD), the 3-wave lab build, and the recomputing form (LIBRECO_DIN_SAVED_H=0)
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_din_gpu.py tests/test_din_fused_gpu.py tests/test_fullsize_cfg345_gpu.py -x -q -k "din" 2>&1 | tail -4
run() {
  env "$@" timeout 300 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('   ms_per_step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'), 'fwd', k['lr_din_attn_pool_fwd_f32']['mean_ms'], 'bwd', k['lr_din_attn_pool_bwd_parts_f32']['mean_ms'])"
}
echo "saved h, product build"; run X=1
echo "saved h, 3 waves/SIMD"; run LIBRECO_HIP_LIB=$PWD/build/lab/libreco_din_w3.so
echo "recompute"; run LIBRECO_DIN_SAVED_H=0
