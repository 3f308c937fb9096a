export PYTHONPATH=.
echo product; timeout 300 python scripts/lab/r06/topk_filter_only.py 2>&1 | tail -1
for t in $TAGS; do
  echo $t; LIBRECO_HIP_LIB=build/lab/libreco_tk_$t.so timeout 300 python scripts/lab/r06/topk_filter_only.py 2>&1 | tail -1
done

