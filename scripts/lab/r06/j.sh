set -x
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q 2>&1 | tail -15
timeout 600 python scripts/lab/r06/topk_filter_time.py 12500000 2>&1 | tail -8
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -8
