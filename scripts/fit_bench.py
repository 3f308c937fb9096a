#!/usr/bin/env python
"""End-to-end `DeepFM.fit` throughput (rows f1 + the hot path): epoch wall time through the product API — host loader
vs device-side sampling / collation, eager launches vs hipGraph replay (dedicated-stream replays, nets/din_fused.py:
GraphRunner).  Synthetic implicit data: 2 M interactions, 2 * FIT_NF plain sparse columns (FIT_NF user + FIT_NF item columns;
default 20 -> 42 fields, FIT_NF=100 -> the 202 fields of BASELINE cfg 2), K = 64, batch 16,384 samples (8,192 positives +
8,192 sampled negatives)."""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd.algorithms import DeepFM  # noqa: E402
from librecommender_amd.data import DatasetFeat  # noqa: E402

import os

rng = np.random.default_rng(0)
n, nu, ni, nf = int(os.environ.get("FIT_N", 2_000_000)), 200_000, 100_000, int(os.environ.get("FIT_NF", 20))
df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.zipf(1.2, n) % ni, "label": 1})
ucols, icols = [f"u{c}" for c in range(nf)], [f"i{c}" for c in range(nf)]
for c in ucols:
    df[c] = rng.integers(0, 1000, nu)[df["user"].values]
for c in icols:
    df[c] = rng.integers(0, 1000, ni)[df["item"].values]
t0 = time.perf_counter()
train, info = DatasetFeat.build_trainset(df, user_col=ucols, item_col=icols, sparse_col=ucols + icols, dense_col=[])
print(f"build_trainset {time.perf_counter() - t0:.1f} s; {len(train)} interactions, {2 + 2 * nf} fields")



def timed_epochs(model, train, epochs=2):
    """Mean wall time of `epochs` further epochs over ONE loader, as `Trainer.run` iterates it inside a fit (the loader —
    device-resident interaction columns, history CSR — is built once per fit: reported apart)."""
    from librecommender_amd.batch import get_batch_loader
    from librecommender_amd.batch.device_loader import DevicePointwiseLoader
    from librecommender_amd.nets.din_fused import lazy_join

    tr = model.trainer
    t0 = time.perf_counter()
    loader = get_batch_loader(model, train, True, tr.batch_size, True, 0, model.seed)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    for ep in range(epochs):
        with lazy_join(isinstance(loader, DevicePointwiseLoader)):          # as training/trainer.py does
            losses = [model.train_on_batch(b) for b in loader]
        model.on_epoch_end(ep + 2)
    float(torch.stack(losses).mean())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / epochs, setup


ONLY = os.environ.get("FIT_BENCH_ONLY")
for tag, kw in (("host loader, eager", dict(device_sampling=False, graph_step=False)),
                ("host loader, hipGraph", dict(device_sampling=False, graph_step=True)),
                ("device loader, eager", dict(device_sampling=True, graph_step=False)),
                ("device loader, hipGraph", dict(device_sampling=True, graph_step=True))):
    if ONLY and ONLY != tag:
        continue
    model = DeepFM("ranking", info, embed_size=64, n_epochs=1, lr=1e-3, batch_size=16384, num_neg=1,
                   hidden_units=(128, 64, 32), sampler="random", **kw)
    model.fit(train, neg_sampling=True, verbose=0)              # epoch 1: builds, warm-up, graph capture
    torch.cuda.synchronize()
    dt, setup = timed_epochs(model, train)
    steps = -(-len(train) // 8192)
    print(f"{tag:44s}: epoch {dt:6.2f} s = {2 * len(train) / dt / 1e6:6.2f} M samples/s  ({dt / steps * 1e3:6.2f} ms per step of 16,384 samples; loader set-up {setup:.2f} s once per fit)")
    del model
    torch.cuda.empty_cache()
