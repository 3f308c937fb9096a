export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_score_topk_gpu.py tests/test_fullsize_parity_gpu.py tests/test_bench_cli_gpu.py -x -q -k "score_topk or filter or tied or two_merge or single_gpu" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python scripts/lab/r06/topk_filter_time.py 100000000 2>&1 | tail -7 | head -5
