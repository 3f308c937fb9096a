#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02ae
mkdir -p "$out"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_feat_api_gpu.py -m gpu -q -x --timeout 600 -k "pair_mlp or catalog" > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python - > "$out/bench.log" 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from librecommender_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for B, N in ((1, 1_000_000), (128, 1_000_000), (1024, 1_000_000)):
    H1, H2 = 128, 64
    P = torch.randn((B, H1), device=dev, generator=g); Q = torch.randn((N, H1), device=dev, generator=g)
    W2 = torch.randn((H1, H2), device=dev, generator=g) / 11; b2 = torch.randn(H2, device=dev, generator=g); v3 = torch.randn(H2, device=dev, generator=g)
    out = torch.zeros((B, N), device=dev)
    ops.pair_mlp(P, Q, W2, b2, v3, 0.1, out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.pair_mlp(P, Q, W2, b2, v3, 0.1, out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = min(ts)
    fl = 2.0 * B * N * H1 * H2
    print(f"pair_mlp B={B} N={N}: {ms:.3f} ms  {B * N / ms / 1e6:.1f} G pairs/s  {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")
PY
tail -n 6 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
cat "$out/bench.log" >> "$out/summary.txt"
cat "$out/summary.txt"
