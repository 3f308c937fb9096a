#!/bin/bash
# r04s: exchange plans off the field-wise sort: kernel tests, sharded suites, bench --force-sharded + its kernel timeline
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04s
mkdir -p "$out"
timeout 900 python -m pytest tests/test_owner_partition_gpu.py tests/test_edge_cases_gpu.py tests/test_sharded_gpu.py tests/test_dist_api_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest.log" | cut -c1-250
timeout 300 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > "$out/force_sharded.json" 2> "$out/force_sharded.err"; echo "force-sharded rc=$?"; grep -o '"ms_per_step": [0-9.]*' "$out/force_sharded.json" | head -1; tail -3 "$out/force_sharded.err" | cut -c1-300
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --force-sharded --steps 12 --warmup 3 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > $out/prof.log 2>&1)
f=$(find /tmp/prof_r04 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && gzip -c "$f" > "$out/kernel_trace_sharded.csv.gz"
