#!/bin/bash
# Everything bench.py's `roofline` objects cite from profiles/, re-collected on THIS tree in one GPU call:
#   per workload  (a) rocprofv3 --kernel-trace --stats of the bench command (+ the bench line of that traced run)
#                 (b) a separate PMC pass (kernel-trace + the four fabric request counters only) for `traffic`
# usage (GPU box):  bash scripts/profile_round.sh <round tag, e.g. r05> [workloads...]
#   then            python scripts/profiles_from_run.py gpurun_out/<tag>_profiles <tag>   (run at the end of this script)
# and copy gpurun_out/<tag>_profiles/<tag>_* into profiles/.
RND=${1:-r05}; shift || true
WL=${@:-deepfm dense_adam din twotower lightgcn recommend_100m deepfm_recommend}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${RND}_profiles
TCC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
COMMON="--no-cpu-baseline --steady-seconds 0"
for w in $WL; do
  mkdir -p $OUT/$w/trace $OUT/$w/pmc
  case $w in
    deepfm)         TR="python bench.py --steps 20 --warmup 5 $COMMON --no-workloads --no-dense-adam-line"
                    PM="python bench.py --steps 3 --warmup 1 $COMMON --no-workloads --no-dense-adam-line --no-recommend --no-graph" ;;
    dense_adam)     TR="python bench.py --steps 5 --warmup 3 $COMMON --no-workloads --no-recommend"
                    PM="python bench.py --steps 2 --warmup 1 $COMMON --no-workloads --no-recommend --no-graph" ;;
    recommend_100m) TR=""
                    PM="python scripts/score_topk_traffic.py --once --arith both" ;;
    deepfm_recommend) TR="python bench.py --workload deepfm_recommend $COMMON"
                    PM="python bench.py --workload deepfm_recommend $COMMON" ;;
    din)            TR="python bench.py --workload din --steps 20 --warmup 5 $COMMON"
                    PM="python bench.py --workload din --steps 3 --warmup 2 $COMMON --no-graph" ;;
    *)              TR="python bench.py --workload $w --steps 10 --warmup 3 $COMMON"
                    PM="python bench.py --workload $w --steps 2 --warmup 1 $COMMON" ;;
  esac
  if [ -n "$TR" ] && [ "$PROFILE_ONLY" != "pmc" ]; then
    echo "cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d <out> -o kt -- $TR" > $OUT/$w/cmd.txt
    (cd /tmp && timeout ${PROFILE_TIMEOUT:-420} rocprofv3 --kernel-trace --stats -f csv -d $OUT/$w/trace -o kt -- \
        bash -c "cd $ROOT && $TR > $OUT/$w/bench.json 2> $OUT/$w/bench.err") > $OUT/$w/trace.log 2>&1
    echo "[$w] trace exit $?"; tail -c 300 $OUT/$w/bench.json | head -c 300; echo
    # the kernel trace itself is large; the stats table is what is kept
    find $OUT/$w/trace -name "*kernel_trace.csv" -size +8M -delete
  fi
  [ "$PROFILE_ONLY" = "trace" ] && continue
  echo "cd /tmp && rocprofv3 --kernel-trace --pmc $TCC -f csv -d <out> -o pmc -- $PM" > $OUT/$w/pmc_cmd.txt
  (cd /tmp && timeout ${PROFILE_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $TCC -f csv -d $OUT/$w/pmc -o pmc -- \
      bash -c "cd $ROOT && $PM > $OUT/$w/pmc_bench.json 2> $OUT/$w/pmc_bench.err") > $OUT/$w/pmc.log 2>&1
  echo "[$w] pmc exit $?"
  find $OUT/$w/pmc -name "*kernel_trace.csv" -delete
done
python scripts/profiles_from_run.py $OUT $RND
# counter CSVs of the full-size passes are tens of MB: the per-kernel means (pmc_kernels.json beside them) are what is kept
find $OUT -name "*counter_collection.csv" -size +1M -delete
du -sh $OUT
