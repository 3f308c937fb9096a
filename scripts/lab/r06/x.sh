export PYTHONPATH=.
timeout 600 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-recommend --no-dense-adam-line > gpurun_out/x_sharded.json 2> gpurun_out/x_sharded.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/x_sharded.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"].get("parallelism"))
k = d.get("kernels") or {}
for n, v in sorted(k.items(), key=lambda kv: -kv[1].get("mean_ms", 0) * kv[1].get("launches", 0))[:25]:
    print(n, v)
PY
