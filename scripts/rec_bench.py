#!/usr/bin/env python
"""SURVEY f2 measurement: DeepFM `recommend_user` over the full catalog — factorised scorer vs the
reference-style materialised forward (chunked, on the same GPU)."""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd.algorithms import DeepFM  # noqa: E402
from librecommender_amd.data import DatasetFeat  # noqa: E402

rng = np.random.default_rng(0)
n, nu, ni = 400_000, 20_000, 100_000
df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.integers(0, ni, n), "label": 1})
ucols = [f"u{c}" for c in range(6)]
icols = [f"i{c}" for c in range(10)]
uf = {c: rng.integers(0, 50, nu) for c in ucols}
itf = {c: rng.integers(0, 200, ni) for c in icols}
for c in ucols:
    df[c] = uf[c][df["user"].values]
for c in icols:
    df[c] = itf[c][df["item"].values]
train, info = DatasetFeat.build_trainset(df, user_col=ucols, item_col=icols, sparse_col=ucols + icols, dense_col=[])
model = DeepFM("ranking", info, embed_size=32, n_epochs=1, lr=1e-3, batch_size=8192, hidden_units=(128, 64, 32))
model.fit(train, neg_sampling=True, verbose=0)
users = [info.id2user[u] for u in range(64)]
torch.cuda.synchronize()
t0 = time.perf_counter(); a = model.recommend_user(users, 10); torch.cuda.synchronize(); t1 = time.perf_counter()
t2 = time.perf_counter(); a = model.recommend_user(users, 10); torch.cuda.synchronize(); t3 = time.perf_counter()
model._catalog_scorer = lambda: None                      # reference-style path
t4 = time.perf_counter(); b = model.recommend_user(users[:8], 10); torch.cuda.synchronize(); t5 = time.perf_counter()
same = all(np.array_equal(a[u], b[u]) for u in users[:8])
N = info.n_items
print(f"items={N} fields={2 + len(ucols) + len(icols)} K=32")
print(f"factorised: first call (item side built) {t1 - t0:.3f}s, cached {t3 - t2:.3f}s for 64 users -> {64 * N / (t3 - t2):.3e} item-scores/s")
print(f"materialised forward: {t5 - t4:.3f}s for 8 users -> {8 * N / (t5 - t4):.3e} item-scores/s; same top-10: {same}")
