#!/bin/bash
# functional run of the N > 1 bench path: two ranks sharing the GPU over gloo (RCCL refuses two ranks per device)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02y
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 3 --small --backend gloo > "$out/bench2.json" 2> "$out/bench2.err"; echo "bench gloo x2 rc=$?" >> "$out/summary.txt"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 3 --small --backend gloo --parallel field > "$out/bench2f.json" 2> "$out/bench2f.err"; echo "bench gloo x2 field rc=$?" >> "$out/summary.txt"
cut -c1-1500 "$out/bench2.json" >> "$out/summary.txt"; tail -n 5 "$out/bench2.err" | cut -c1-300 >> "$out/summary.txt"
cut -c1-600 "$out/bench2f.json" >> "$out/summary.txt"; tail -n 5 "$out/bench2f.err" | cut -c1-300 >> "$out/summary.txt"
cat "$out/summary.txt"
timeout 120 python scripts/mfma_peak.py > "$out/mfma_peak.log" 2>&1; cat "$out/mfma_peak.log"
