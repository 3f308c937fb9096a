#!/usr/bin/env python
"""Streaming in-batch softmax-CE at cfg 4's batch (B = N = 65,536, D = 128): HIP-event times of the forward
(+W) sweep and the column-gradient sweep, against the materialised torch path at a size that fits.
usage: python scripts/sce_bench.py [B] [D] [reps]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
X = torch.nn.functional.normalize(torch.randn((B, D), device=dev, generator=g), dim=1) / 0.1
Y = torch.nn.functional.normalize(torch.randn((B, D), device=dev, generator=g), dim=1)
bias = -torch.log(torch.rand(B, device=dev, generator=g).clamp_(1e-6, 1.0))
ids = torch.randint(0, B * 4, (B,), device=dev, generator=g, dtype=torch.int32)
gr = torch.full((B,), 1.0 / B, device=dev)
if os.environ.get("SCE_BENCH_ZEROS"):        # operands of zeros: the same instruction stream with (almost) no switching in the MFMA datapath
    X.zero_(); Y.zero_()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts), min(ts)


flop1 = 2.0 * B * B * D
for arith in os.environ.get("SCE_BENCH_ARITHS", "split_bf16,f32_chain").split(","):
    ops.set_sce_arith(arith)
    lse, pos, W = ops.softmax_ce_fwd(X, Y, bias, ids, ids, 0)
    print(f"-- {arith}")
    for name, fn, nf in (("fwd (lse only)", lambda: ops.softmax_ce_fwd(X, Y, bias, ids, ids, 0, want_w=False), 1),
                         ("fwd + W", lambda: ops.softmax_ce_fwd(X, Y, bias, ids, ids, 0), 2),
                         ("bwd cols", lambda: ops.softmax_ce_bwd_cols(X, Y, lse, gr, bias, ids, ids, 0), 2)):
        mean, mn = timed(fn)
        print(f"{name:16s} ms: mean {mean:8.3f} min {mn:8.3f}  {nf * flop1 / mn / 1e9:7.1f} TFLOP/s f32-equivalent "
              f"({nf * flop1 / mn / 1e9 / 157.3 * 100:.1f}% of the f32 MFMA peak"
              + (f", {6 * nf * flop1 / mn / 1e9 / 2500 * 100:.1f}% of the bf16 MFMA peak by the six products)" if arith == "split_bf16" else ")"))
