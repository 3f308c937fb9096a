export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_score_topk_gpu.py tests/test_bench_cli_gpu.py tests/test_serving_gpu.py tests/test_sharded_gpu.py -x -q 2>&1 | tail -5
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_fullsize_cfg345_gpu.py -x -q -k "score_topk or ranking or recommend" 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads > gpurun_out/o_bench.json 2> gpurun_out/o_bench.err; tail -3 gpurun_out/o_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/o_bench.json").read().strip().splitlines()[-1])
r = d["recommend"]
print(json.dumps({k: r[k] for k in ("value", "ms_per_pass", "roofline", "split_bf16", "f32_chain")}, indent=1)[:3000])
print(d["config"]["other_legs"])
PY
