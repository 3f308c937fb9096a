#!/bin/bash
# r05ai: kernel timeline of one replayed DIN step and one DeepFM step on the final tree
out=gpurun_out/r05ai; mkdir -p $out
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $ROOT/$out/din -o kt -- bash -c "cd $ROOT && python bench.py --workload din --steps 12 --warmup 3 --no-cpu-baseline --steady-seconds 0 > /dev/null 2>&1") > $out/din.log 2>&1
python scripts/trace_timeline.py $(find $out/din -name "*kernel_trace.csv" | head -1) din_fwd_mfma 0 > $out/din_timeline.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $ROOT/$out/deepfm -o kt -- bash -c "cd $ROOT && python bench.py --steps 12 --warmup 3 --no-cpu-baseline --steady-seconds 0 --no-recommend --no-workloads --no-dense-adam-line > /dev/null 2>&1") > $out/deepfm.log 2>&1
python scripts/trace_timeline.py $(find $out/deepfm -name "*kernel_trace.csv" | head -1) idx_transpose 0 > $out/deepfm_timeline.txt 2>&1
find $out -name "*.csv" -delete
head -5 $out/din_timeline.txt; head -3 $out/deepfm_timeline.txt
