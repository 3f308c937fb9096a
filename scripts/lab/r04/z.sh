#!/bin/bash
# r04z: wgrad with the ids requested one slab ahead: first-layer parity tests + step time
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04z
mkdir -p "$out"
timeout 900 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_l1_wide_gpu.py tests/test_din_fused_gpu.py tests/test_feat_block_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$out/pytest.log"
timeout 600 python bench.py --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line 2> "$out/bench.err" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('deepfm ms', d['ms_per_step'], d.get('steady_state',{}).get('ms_per_step'))
for n in ('lr_deepfm_l1_fwd_f32','lr_deepfm_l1_wgrad_f32','lr_deepfm_l1_dgrad_f32','lr_fm_rows_adam_f32'): print(n, d['kernels'][n]['mean_ms'])"
