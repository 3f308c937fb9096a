export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_score_topk_gpu.py tests/test_rank_seam_gpu.py -x -q 2>&1 | tail -4
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "score_topk" 2>&1 | tail -3
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -7
timeout 600 python scripts/lab/r06/topk_ties_time.py 2>&1 | tail -4
timeout 600 python scripts/lab/r06/topk_filter_time.py 1048576 2>&1 | tail -7 | head -5
timeout 600 python scripts/lab/r06/topk_filter_time.py 3000000 2>&1 | tail -7 | head -5
