export PYTHONPATH=.
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q 2>&1 | tail -5
echo product; timeout 300 python scripts/lab/r06/topk_filter_only.py 2>&1 | tail -1
for t in $TAGS; do
  echo $t; LIBRECO_HIP_LIB=build/lab/libreco_tk_$t.so timeout 300 python scripts/lab/r06/topk_filter_only.py 2>&1 | tail -1
done
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -8
