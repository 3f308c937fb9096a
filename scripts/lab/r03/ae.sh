#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03ae
timeout 600 python bench.py --workload lightgcn --small --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03ae/lg_w2.json 2> gpurun_out/r03ae/lg_w2.err; echo "w2 rc=$?"
tail -c 900 gpurun_out/r03ae/lg_w2.json; echo; tail -3 gpurun_out/r03ae/lg_w2.err | cut -c1-300
timeout 900 python bench.py --workload lightgcn --force-sharded --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03ae/lg_fs.json 2> gpurun_out/r03ae/lg_fs.err; echo "fs rc=$?"
tail -c 1200 gpurun_out/r03ae/lg_fs.json; echo; tail -3 gpurun_out/r03ae/lg_fs.err | cut -c1-300
