export PYTHONPATH=.
TCC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
bash scripts/pmc_cmd.sh tkf "PYTHONPATH=. python scripts/lab/r06/topk_filter_once.py" "$TCC" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" 2>&1 | grep -v "^$" | cut -c1-600
export TMPDIR=/tmp; ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/tkf_trace -o t -- bash -c "cd $ROOT && PYTHONPATH=. python scripts/lab/r06/topk_filter_once.py" > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tkf_trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "lr::" in r["Name"]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
