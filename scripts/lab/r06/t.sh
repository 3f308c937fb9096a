export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_score_topk_gpu.py tests/test_rank_seam_gpu.py tests/test_fullsize_parity_gpu.py -x -q -k "score_topk or filter or tied or rank" 2>&1 | tail -5
timeout 900 python scripts/lab/r06/topk_small_b.py 2>&1 | tail -10
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -7 | head -5
