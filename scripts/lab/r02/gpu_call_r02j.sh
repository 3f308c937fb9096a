#!/bin/bash
# SpMM bucketing tests/bench + rocprof summaries (kernel suite, model suite) + PMC of DIN / softmax-CE / SpMM
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02j
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
ROOT=$PWD
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_lightgcn_gpu.py -m gpu -q -x --timeout 600 -k "spmm or lightgcn or LightGCN" > "$out/tests.log" 2>&1; echo "spmm tests rc=$?" >> "$out/summary.txt"
timeout 400 python scripts/kern_suite.py spmm > "$out/spmm.log" 2>&1; echo "spmm bench rc=$?" >> "$out/summary.txt"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$out/prof_kern -o ks -- python $ROOT/scripts/kern_suite.py din spmm gather scatter > $ROOT/$out/prof_kern.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$out/prof_model -o ms -- python $ROOT/scripts/model_suite.py din twotower lightgcn > $ROOT/$out/prof_model.log 2>&1)
bash scripts/pmc_cmd.sh din "python scripts/kern_suite.py din" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" > "$out/pmc_din.log" 2>&1
bash scripts/pmc_cmd.sh sce "python scripts/sce_bench.py 32768 128 2" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" > "$out/pmc_sce.log" 2>&1
bash scripts/pmc_cmd.sh spmm "python scripts/kern_suite.py spmm" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" > "$out/pmc_spmm.log" 2>&1
tail -n 6 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
cat "$out/spmm.log" >> "$out/summary.txt"
tail -n 8 "$out/prof_kern.log" "$out/prof_model.log" | cut -c1-200 >> "$out/summary.txt"
grep -h "lr::" "$out/pmc_din.log" "$out/pmc_sce.log" "$out/pmc_spmm.log" | cut -c1-700 >> "$out/summary.txt"
cat "$out/summary.txt"
