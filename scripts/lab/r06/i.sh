#!/bin/bash
# r06i: merged BatchNorm-fold chain: tests, then DeepFM / DIN steps with it and with the four-launch chain
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_deepfm_fused_gpu.py tests/test_din_fused_gpu.py tests/test_l1_split_bf16_gpu.py tests/test_feat_block_gpu.py tests/test_dense_adam_fused_gpu.py tests/test_sharded_gpu.py -x -q 2>&1 | tail -4
run() {
  for w in din deepfm; do
  env "$@" timeout 300 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-workloads --no-recommend --no-dense-adam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ', d['config']['workload'][:10], 'ms_per_step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'))"
  done
}
echo "merged"; run X=1
echo "chain"; run LIBRECO_FOLD_CHAIN=chain
echo "merged"; run X=1
