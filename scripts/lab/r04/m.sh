#!/bin/bash
# r04m: SpMM long-row path: workgroups on the chunk list x loads in flight per row group (cfg 5 graph, exact Zipf(1.05) endpoints)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04m
mkdir -p "$out"
run() {
  LIBRECO_HIP_LIB=$1 timeout 600 python bench.py --workload lightgcn --steps 6 --warmup 2 --no-cpu-baseline --steady-seconds 0 > "$out/$2.json" 2> "$out/$2.err"
  python - "$out/$2.json" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "step ms", d["ms_per_step"], "spmm ms", d["kernels"]["lr_spmm_csr_bucketed_f32"]["mean_ms"], "frac", d["roofline"]["frac"])
PY
}
run $PWD/librecommender_amd/lib/liblibreco_hip.so base_256_0
for v in 256_1 512_0 512_1 1024_0 1024_1; do run $PWD/build/lab/libreco_sp_$v.so sp_$v; done
