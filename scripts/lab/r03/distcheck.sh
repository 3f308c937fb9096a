#!/bin/bash
# final-tree check of every multi-rank bench leg that a one-GPU box can run (RCCL at world size 1; two ranks sharing the GPU over gloo)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03dist
mkdir -p "$out"
timeout 300 python bench.py --force-sharded --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --steady-seconds 0 > "$out/fs_deepfm.json" 2> "$out/fs_deepfm.err"; echo "fs deepfm rc=$? lines=$(wc -l < $out/fs_deepfm.json)"
timeout 300 python bench.py --force-sharded --parallel field --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --steady-seconds 0 > "$out/fs_field.json" 2> "$out/fs_field.err"; echo "fs field rc=$? lines=$(wc -l < $out/fs_field.json)"
timeout 300 python bench.py --gpus 2 --backend gloo --small --steps 2 --warmup 1 --no-cpu-baseline > "$out/gloo2.json" 2> "$out/gloo2.err"; echo "gloo2 rc=$? lines=$(wc -l < $out/gloo2.json)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --force-sharded --steps 5 --warmup 2 --no-cpu-baseline --steady-seconds 0 > "$out/torchrun1.json" 2> "$out/torchrun1.err"; echo "torchrun1 rc=$? lines=$(wc -l < $out/torchrun1.json)"
timeout 400 python bench.py --workload twotower --force-sharded --steps 3 --warmup 1 --no-cpu-baseline > "$out/fs_tt.json" 2> "$out/fs_tt.err"; echo "fs twotower rc=$? lines=$(wc -l < $out/fs_tt.json)"
for f in fs_deepfm fs_field gloo2 torchrun1 fs_tt; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $out/$f.json | head -1)"; done
