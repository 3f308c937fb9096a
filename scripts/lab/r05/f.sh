#!/bin/bash
# r05f: the fused three-layer tail: bit-for-bit against the chain, the models' tests, DeepFM / DIN bench legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05f; mkdir -p $out
timeout 300 python -m pytest tests/test_tail_fused_gpu.py -q -x -m gpu --timeout 120 > $out/tail_tests.log 2>&1; echo "tail tests rc=$?"; tail -12 $out/tail_tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_tail_dropout_gpu.py tests/test_din_fused_gpu.py tests/test_graph_fit_gpu.py tests/test_feat_block_gpu.py -q -m gpu --timeout 200 > $out/model_tests.log 2>&1; echo "model tests rc=$?"; tail -6 $out/model_tests.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line > $out/bench_deepfm.json 2> $out/bench_deepfm.err; echo "bench rc=$?"
timeout 300 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_din.json 2> $out/bench_din.err; echo "din rc=$?"
python - <<'PY'
import json
for f in ("bench_deepfm","bench_din"):
    try:
        r=json.loads(open(f"gpurun_out/r05f/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:r.get(k) for k in ("value","ms_per_step","steady_ms_per_step","f32_chain_ms_per_step","sum_kernel_ms")})
    except Exception as e:
        print(f, "parse failed", e); print(open(f"gpurun_out/r05f/{f}.err").read()[-1500:])
PY
