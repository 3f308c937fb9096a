#!/bin/bash
out=gpurun_out/r05y; mkdir -p $out
timeout 900 python -m pytest tests/test_din_fused_gpu.py tests/test_din_tower_models_gpu.py tests/test_din_gpu.py tests/test_zz_din_device_loader_gpu.py tests/test_fullsize_cfg345_gpu.py tests/test_graph_nodes_gpu.py -x -q 2>&1 | tail -5 | tee $out/pytest.log
timeout 300 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_din.json 2> $out/bench_din.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05y/bench_din.json').read().strip().splitlines()[-1])
print('din', d['ms_per_step'], d.get('steady_state'), d['sum_kernel_ms'])
PY
