"""Host loader throughput on a wide feature set (the pointwise collator's one-pass C merge):
    python scripts/loader_bench_wide.py      # 40 sparse columns (20 user + 20 item), 16,384-sample batches, one core
Set LIBRECO_NO_HOSTLIB=1 to time the numpy definition of the same collation."""
import sys, time, types
import numpy as np, pandas as pd
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import os
from librecommender_amd import _hostlib
if os.environ.get("LIBRECO_NO_HOSTLIB"):
    _hostlib._tried, _hostlib._lib = True, None
from librecommender_amd.batch import get_batch_loader
from librecommender_amd.data import DatasetFeat
rng = np.random.default_rng(0)
n, nu, ni, nf = 300_000, 200_000, 100_000, 20
df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.zipf(1.2, n) % ni, "label": 1})
ucols, icols = [f"u{c}" for c in range(nf)], [f"i{c}" for c in range(nf)]
for c in ucols: df[c] = rng.integers(0, 1000, nu)[df["user"].values]
for c in icols: df[c] = rng.integers(0, 1000, ni)[df["item"].values]
ts, info = DatasetFeat.build_trainset(df, user_col=ucols, item_col=icols, sparse_col=ucols + icols, dense_col=[])
print("dtypes", ts.sparse_indices.dtype, info.item_sparse_unique.dtype, ts.sparse_indices.flags.c_contiguous, info.item_sparse_unique.flags.c_contiguous, _hostlib.load() is not None)
m = types.SimpleNamespace(model_name="DeepFM", data_info=info, seed=42, task="ranking", sampler="random", num_neg=1,
                          loss_type="cross_entropy", uses_features=True, uses_sequence=False, graph_backend="tf")
for rep in range(3):
    loader = get_batch_loader(m, ts, True, batch_size=8192, shuffle=True, num_workers=0, seed=42)
    nb = 0
    for b in loader:
        nb += 1
        if nb == 5: t0 = time.perf_counter()
        if nb == 35: break
    dt = time.perf_counter() - t0
    print(f"{dt / 30 * 1e3:.2f} ms per batch")
