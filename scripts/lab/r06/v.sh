export PYTHONPATH=.
for t in marks marksg2; do echo $t; LIBRECO_HIP_LIB=build/lab/libreco_tk_$t.so timeout 600 python scripts/lab/r06/topk_marks.py 2>&1 | tail -9; done
