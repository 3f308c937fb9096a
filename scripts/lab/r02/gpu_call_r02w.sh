#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02w
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_softmax_ce_gpu.py tests/test_retrain_gpu.py tests/test_din_tower_models_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python - > "$out/sce_sharded.log" 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from librecommender_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, D = 65536, 128
Y = torch.nn.functional.normalize(torch.randn((N, D), device=dev, generator=g), dim=1)
for B in (8192, 16384, 32768):
    X = torch.nn.functional.normalize(torch.randn((B, D), device=dev, generator=g), dim=1) / 0.1
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return min(ts)
    lse, pos, W = ops.softmax_ce_fwd(X, Y, pos0=B)
    gr = torch.full((B,), 1.0 / B, device=dev)
    tf = timed(lambda: ops.softmax_ce_fwd(X, Y, pos0=B))
    tb = timed(lambda: ops.softmax_ce_bwd_cols(X, Y, lse, gr, pos0=B))
    fl = 4.0 * B * N * D
    print(f"B_local {B} x N {N}: fwd+W {tf:.3f} ms ({fl / tf / 1e9:.1f} TF/s)  bwd cols {tb:.3f} ms ({fl / tb / 1e9:.1f} TF/s)")
PY
tail -n 6 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
cat "$out/sce_sharded.log" >> "$out/summary.txt"
cat "$out/summary.txt"
