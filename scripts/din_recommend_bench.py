#!/usr/bin/env python
"""SURVEY f2 measurement: DIN `recommend_user` over the full catalog (no factorised form: attention over
(sequence, item) pairs) — feature rows assembled on the device vs built on the host per chunk (round-1 path)."""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd.algorithms import DIN  # noqa: E402
from librecommender_amd.bases.feat_base import merge_user_item_feats  # noqa: E402
from librecommender_amd.data import DatasetFeat  # noqa: E402


def host_rows_scores(model, uid):
    N = model.n_items
    out = torch.empty(N, dtype=torch.float32, device=model.device)
    seqs1, lens1 = model._seq_for(uid, None)
    for s in range(0, N, model.score_chunk):
        items = np.arange(s, min(N, s + model.score_chunk))
        users = np.full(len(items), uid)
        sparse, dense = merge_user_item_feats(model.data_info, users, items)
        out[s:s + len(items)] = model._forward(users, items, sparse, dense, np.repeat(seqs1, len(items), axis=0),
                                               np.repeat(lens1, len(items)))
    return out


rng = np.random.default_rng(0)
n, nu, ni = 400_000, 20_000, 100_000
base = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.integers(0, ni, n), "label": 1,
                     "time": rng.integers(0, 10**6, n)})
ucols = [f"u{c}" for c in range(6)]
for c in ucols:
    base[c] = rng.integers(0, 50, nu)[base["user"].values]
for tag, icols in (("pure items (fused attention kernel)", []), ("3 item columns (general path, K' = 128)", [f"i{c}" for c in range(3)]),
                   ("4 item columns (K' = 160 > 128: torch attention)", [f"i{c}" for c in range(4)])):
    df = base.copy()
    for c in icols:
        df[c] = rng.integers(0, 200, ni)[df["item"].values]
    train, info = DatasetFeat.build_trainset(df, user_col=ucols, item_col=icols, sparse_col=ucols + icols, dense_col=[])
    model = DIN("ranking", info, embed_size=32, n_epochs=1, lr=1e-3, batch_size=8192, hidden_units=(128, 64, 32), recent_num=50)
    model.fit(train, neg_sampling=True, verbose=0)
    N = info.n_items
    a = model._scores_all_items(0, None, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(8):
        a = model._scores_all_items(u, None, None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    b = host_rows_scores(model, 0); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for u in range(4):
        b = host_rows_scores(model, u)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    a3 = model._scores_all_items(3, None, None)
    same = torch.allclose(a3, b, rtol=1e-4, atol=1e-5)
    users = [info.id2user[u] for u in range(8)]
    t4 = time.perf_counter(); model.recommend_user(users, 10); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"DIN {tag}: items={N} L={model.max_seq_len} K=32 fused={model.net.fused}")
    print(f"  device-assembled rows: {(t1 - t0) / 8 * 1e3:.1f} ms/user -> {8 * N / (t1 - t0):.3e} item-scores/s")
    print(f"  host-assembled rows (round 1): {(t3 - t2) / 4 * 1e3:.1f} ms/user -> {4 * N / (t3 - t2):.3e} item-scores/s; same scores: {same}")
    print(f"  recommend_user(8 users, 10): {(t5 - t4) * 1e3:.1f} ms")
    if not icols:
        import cProfile
        import pstats
        pr = cProfile.Profile(); pr.enable(); model.recommend_user(users, 10); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
