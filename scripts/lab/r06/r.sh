export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_score_topk_gpu.py -x -q 2>&1 | tail -4
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "score_topk" 2>&1 | tail -3
timeout 600 python scripts/lab/r06/topk_filter_fallback.py 2>&1 | tail -8
