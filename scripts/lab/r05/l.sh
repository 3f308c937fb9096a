#!/bin/bash
# r05l: tail staging with per-thread column parameters: phase marks, tests, bench legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05l; mkdir -p $out
timeout 120 python scripts/lab/r05/tail_marks.py 2>&1 | tail -36
timeout 600 python -m pytest tests/test_tail_fused_gpu.py tests/test_deepfm_fused_gpu.py tests/test_tail_dropout_gpu.py tests/test_din_fused_gpu.py tests/test_feat_block_gpu.py tests/test_din_tower_models_gpu.py -q -m gpu --timeout 200 > $out/tail_tests.log 2>&1; echo "tail tests rc=$?"; tail -3 $out/tail_tests.log | cut -c1-300
for mode in fused chain; do
  LIBRECO_TAIL=$mode timeout 300 python bench.py --steps 20 --warmup 5 --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line > $out/bench_deepfm_$mode.json 2> $out/bench_deepfm_$mode.err; echo "bench $mode rc=$?"
  LIBRECO_TAIL=$mode timeout 300 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_din_$mode.json 2> $out/bench_din_$mode.err; echo "din $mode rc=$?"
done
python - <<'PY'
import json
for f in ("bench_deepfm_fused","bench_deepfm_chain","bench_din_fused","bench_din_chain"):
    try:
        r=json.loads(open(f"gpurun_out/r05l/{f}.json").read().strip().splitlines()[-1])
        k=r["kernels"]
        print(f, r["ms_per_step"], r.get("steady_ms_per_step"), r.get("f32_chain_ms_per_step"), "tail3", k.get("lr_mlp_tail3_f32",{}).get("mean_ms"), "rows_adam", k.get("lr_fm_rows_adam_f32",{}).get("mean_ms"), "sum", r.get("sum_kernel_ms"))
    except Exception as e:
        print(f, "parse failed", e); print(open(f"gpurun_out/r05l/{f}.err").read()[-1500:])
PY
