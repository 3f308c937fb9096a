#!/bin/bash
# r06h: DIN with the balanced sample order: tests, then the step with / without it
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_din_gpu.py tests/test_din_fused_gpu.py tests/test_fullsize_cfg345_gpu.py tests/test_zz_din_device_loader_gpu.py tests/test_din_tower_models_gpu.py -x -q -k "din or DIN" 2>&1 | tail -4
run() {
  env "$@" timeout 300 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('   ms_per_step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'), 'fwd', k['lr_din_attn_pool_fwd_f32']['mean_ms'], 'bwd', k['lr_din_attn_pool_bwd_parts_f32']['mean_ms'])"
}
echo "balanced order + saved h"; run X=1
echo "identity order + saved h"; run LIBRECO_DIN_ORDER=0
echo "identity order, recompute"; run LIBRECO_DIN_ORDER=0 LIBRECO_DIN_SAVED_H=0
