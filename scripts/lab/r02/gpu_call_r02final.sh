#!/bin/bash
# full GPU suite + the driver's bench command + profile of the round's final state
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02final
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$out/pytest_gpu.log" 2>&1; echo "pytest -m gpu rc=$?" >> "$out/summary.txt"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err"; echo "bench (driver cmd) rc=$?" >> "$out/summary.txt"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OLDPWD/$out/prof_stdout.log 2>&1)
tail -n 8 "$out/pytest_gpu.log" | cut -c1-300 >> "$out/summary.txt"
cut -c1-6000 "$out/bench_driver_cmd.json" >> "$out/summary.txt"
tail -n 3 "$out/bench_driver_cmd.err" >> "$out/summary.txt"
tail -n 60 "$out/summary.txt"
