#!/bin/bash
# One gpurun call that answers the open questions left by a round that ended without GPU budget:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/first_gpu_call.sh'
# Results land in gpurun_out/first_call/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/first_call
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/summary.txt"
# the tests written after the previous round's GPU budget was spent run first, one file at a time
for t in tests/test_zz_*_gpu.py; do
  timeout 600 python -m pytest "$t" -x -q > "$out/$(basename "$t" .py).log" 2>&1; echo "$t rc=$?" >> "$out/summary.txt"
done
timeout 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest -m gpu rc=$?" >> "$out/summary.txt"
common="--no-cpu-baseline --no-recommend --steps 30 --warmup 5"
timeout 300 python bench.py $common > "$out/bench_n1.json" 2> "$out/bench_n1.err"
timeout 300 python bench.py $common --force-sharded > "$out/bench_field_w1.json" 2> "$out/bench_field_w1.err"
timeout 300 python bench.py $common --force-sharded --parallel row > "$out/bench_row_w1.json" 2> "$out/bench_row_w1.err"
timeout 300 python scripts/field_parallel_gemm_bench.py > "$out/field_parallel_gemm.log" 2>&1
# functional: two ranks sharing the GPU over gloo (host-staged collectives), small shapes
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
  bench.py --gpus 2 --backend gloo --small --steps 3 --warmup 1 --no-recommend > "$out/bench_gloo_w2.json" 2> "$out/bench_gloo_w2.err"
echo "gloo w2 rc=$?" >> "$out/summary.txt"
tail -n 3 "$out"/*.log >> "$out/summary.txt" 2>/dev/null
cat "$out/summary.txt"
