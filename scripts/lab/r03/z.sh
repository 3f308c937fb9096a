#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03z
mkdir -p "$out"
timeout 900 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --steady-seconds 0 > "$out/w2_deepfm.json" 2> "$out/w2_deepfm.err"; echo "w2 deepfm rc=$?"
tail -c 1500 "$out/w2_deepfm.json" | cut -c1-1500
timeout 900 python bench.py --workload twotower --small --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --steady-seconds 0 > "$out/w2_tt.json" 2> "$out/w2_tt.err"; echo "w2 twotower rc=$?"
tail -c 1200 "$out/w2_tt.json" | cut -c1-1200
tail -3 "$out/w2_tt.err" | cut -c1-300
timeout 600 python bench.py --force-sharded --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --steady-seconds 0 > "$out/fs.json" 2> "$out/fs.err"; echo "force-sharded rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/fs.json"
