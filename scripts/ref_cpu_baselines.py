#!/usr/bin/env python
"""CPU baselines from the REFERENCE's own code (build container only: needs /root/reference), SURVEY 8(d)(i):
  * LightGCN.fit (training/torch_trainer.py:77-121) on the reference's sample MovieLens ratings (cfg 1's data),
    BPR loss, embed_size 16 and 64, 3 layers — train samples/s on the host cores;
  * recommend_from_embedding + rank_recommendations (recommendation/recommend.py:57-78, ranking.py:10-56) on a
    synthetic 1 M x 128 catalogue — item-scores/s.
usage: python scripts/ref_cpu_baselines.py > profiles/r04_cpu_reference_baselines.md   (also writes profiles/r04_cpu_reference_baselines.json,
which bench.py attaches to its line as `reference_checkout.reference_run_build_container`)"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref_loader  # noqa: E402

ref_loader.load()
import torch  # noqa: E402
from libreco.algorithms.lightgcn import LightGCN  # noqa: E402
from libreco.data import DatasetPure  # noqa: E402
from libreco.recommendation.ranking import rank_recommendations  # noqa: E402

J = {"where": "build container", "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "lightgcn_fit": [], "note": "the reference's own code imported through oracle/ref_loader.py (SURVEY Appendix A); not reproducible on the GPU box, where /root/reference does not exist"}
print("# r04 — CPU baselines measured with the reference's own code (build container, "
      f"{os.cpu_count()} cores, torch {torch.__version__}, {torch.get_num_threads()} threads)\n")
print("`python scripts/ref_cpu_baselines.py`; the reference checkout is imported through `oracle/ref_loader.py` "
      "(torch / numpy paths; TensorFlow models cannot run here).\n")
df = pd.read_csv(ref_loader.REFERENCE / "examples" / "sample_data" / "sample_movielens_rating.dat", sep="::", engine="python",
                 names=["user", "item", "label", "time"])[["user", "item", "label"]]
train, info = DatasetPure.build_trainset(df)
print(f"## LightGCN.fit on sample_movielens_rating.dat ({len(df)} interactions, {info.n_users} users, {info.n_items} items)\n")
print("| embed_size | batch | epochs | s / epoch | train samples/s (positives + negatives) |\n|---|---|---|---|---|")
for K, bs in ((16, 2048), (64, 2048), (64, 8192)):
    model = LightGCN("ranking", info, loss_type="bpr", embed_size=K, n_epochs=1, lr=1e-3, batch_size=bs, num_neg=1,
                     n_layers=3, device="cpu", seed=42)
    model.fit(train, neg_sampling=True, verbose=0)             # warm-up epoch (graph build, allocations)
    model.n_epochs = 2
    t0 = time.perf_counter()
    model.fit(train, neg_sampling=True, verbose=0)
    dt = (time.perf_counter() - t0) / 2
    print(f"| {K} | {bs} | 2 | {dt:.2f} | {2 * len(df) / dt:,.0f} |")
    J["lightgcn_fit"].append({"data": "examples/sample_data/sample_movielens_rating.dat", "embed_size": K, "n_layers": 3, "batch_size": bs,
                              "s_per_epoch": round(dt, 3), "samples_per_s": round(2 * len(df) / dt, 1), "unit": "samples/s (positives + negatives)"})
print()
rng = np.random.default_rng(0)
N, D, k, B = 1_000_000, 128, 100, 64
I = rng.standard_normal((N, D)).astype(np.float32)
consumed = {u: sorted(rng.integers(0, N, 50).tolist()) for u in range(B)}
t_tot, n = 0.0, 0
while t_tot < 20:
    U = rng.standard_normal((B, D)).astype(np.float32)
    t0 = time.perf_counter()
    preds = U @ I.T
    rank_recommendations("ranking", list(range(B)), preds, k, N, consumed, True, False)
    t_tot += time.perf_counter() - t0
    n += B
print(f"## recommend_from_embedding + rank_recommendations, {N:,} items x {D} dims, k = {k}, 50 consumed ids per user\n")
print(f"{n} users in {t_tot:.1f} s -> **{n * N / t_tot:,.0f} item-scores/s** on {os.cpu_count()} cores "
      f"(the MI355X path: bench.py `recommend`).")
J["recommend"] = {"function": "recommend_from_embedding's product + rank_recommendations (recommendation/recommend.py:57-78, ranking.py:10-56)",
                  "users": n, "items": N, "dims": D, "k": k, "value": round(n * N / t_tot, 1), "unit": "items/s"}
with open(ROOT / "profiles" / "r04_cpu_reference_baselines.json", "w") as fh:
    json.dump(J, fh, indent=1)
