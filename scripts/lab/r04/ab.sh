#!/bin/bash
# r04ab: rows_adam / rows_grad with the software-pipelined walk over the short runs vs the kernel as it stands (same box)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04ab
mkdir -p "$out"
run() {
  timeout 600 python bench.py --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line --steady-seconds 0 $2 2> /dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', 'step', d['ms_per_step'], {n: k[n]['mean_ms'] for n in k if 'rows' in n or 'peer' in n})"
}
run base ""
run base-sharded "--force-sharded"
export LIBRECO_HIP_LIB=$PWD/build/lab/libreco_fmpipe.so
run pipe ""
run pipe-sharded "--force-sharded"
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -2
unset LIBRECO_HIP_LIB
run base2 ""
