export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_score_topk_gpu.py -x -q 2>&1 | tail -4
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "score_topk" 2>&1 | tail -3
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -7
timeout 300 python scripts/lab/r06/topk_filter_only.py 12500000 2>&1 | tail -1
timeout 300 python scripts/lab/r06/topk_filter_skew.py 2>&1 | tail -5
