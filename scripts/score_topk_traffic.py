#!/usr/bin/env python
"""lr_score_topk_f32 over an N x 128 catalogue (default 100 M items, 1,024 users, k = 100): time per pass, or — with
--once — a single pass for a rocprofv3 --pmc run.  Items and users are drawn on the device."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=100_000_000)
ap.add_argument("--users", type=int, default=1024)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--once", action="store_true")
ap.add_argument("--arith", choices=["filter", "split_bf16", "f32_chain", "both"], default=None,
                help="arithmetic of the score contraction (default: ops.TOPK_ARITH); both = one pass of each form (--once)")
a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
items = torch.empty((a.items, 128), dtype=torch.float32, device=dev)
for s in range(0, a.items, 10_000_000):
    items[s:s + 10_000_000].normal_(generator=g)
users = torch.randn((a.users, 128), generator=g, device=dev)
if a.arith == "both":
    ops.score_topk(users, items, a.k, arith="f32_chain")
    ops.score_topk(users, items, a.k, arith="split_bf16")
    ops.score_topk(users, items, a.k, arith="filter")
sc, ids = ops.score_topk(users, items, a.k, arith=None if a.arith in (None, "both") else a.arith)
torch.cuda.synchronize()
if not a.once:
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        sc, ids = ops.score_topk(users, items, a.k)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    ms = min(t) * 1e3
    print(f"items={a.items} users={a.users} k={a.k}: {ms:.2f} ms/pass, {2 * a.users * a.items * 128 / ms / 1e9:.1f} TFLOP/s "
          f"({2 * a.users * a.items * 128 / ms / 1e9 / 157.3:.3f} of the f32 MFMA peak)")
# spot check against a direct product on a slice that holds the winners of user 0
u0 = users[:1]
best = torch.topk((items[: min(a.items, 20_000_000)] @ u0.t()).view(-1), 5)
print("user 0 top ids", ids[0, :5].tolist(), "scores", [round(x, 4) for x in sc[0, :5].tolist()], "| first-20M slice best", best.indices.tolist()[:3])
