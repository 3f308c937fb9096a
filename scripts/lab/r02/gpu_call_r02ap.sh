#!/bin/bash
# r02ap: bench.py's N>1 glue end to end with 2 ranks sharing the one GPU (gloo backend; RCCL needs one GPU per rank):
# row-sharded fused step, prefetch of the next plan, barrier + max-over-ranks timing, rank-0 JSON line
set -x
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 4 --warmup 2 --backend gloo --no-cpu-baseline > gpurun_out/r02ap_bench_w2_gloo.txt 2>&1
tail -5 gpurun_out/r02ap_bench_w2_gloo.txt
