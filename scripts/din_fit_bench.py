#!/usr/bin/env python
"""End-to-end `DIN.fit` through the product API at a cfg-3-like shape (K = 128, L = 50, batch 8,192 samples): host loader
vs device-side loader, eager launches vs the fused step replayed as one hipGraph.  Second epoch timed; the graph and
eager runs of the device loader must end at identical tables (same seeds, counter-based sampler)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd.algorithms import DIN  # noqa: E402
from librecommender_amd.data import DatasetPure  # noqa: E402

rng = np.random.default_rng(0)
n, nu, ni = int(os.environ.get("FIT_N", 1_000_000)), 100_000, 500_000
df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.zipf(1.15, n) % ni, "label": 1, "time": np.arange(n)})
train, info = DatasetPure.build_trainset(df)
print(f"{len(train)} interactions, {info.n_users} users, {info.n_items} items", flush=True)
ONLY = os.environ.get("FIT_BENCH_ONLY")
tables = {}
for tag, kw in (("host loader, hipGraph", dict(device_sampling=False, graph_step=True)),
                ("device loader, eager", dict(device_sampling=True, graph_step=False)),
                ("device loader, hipGraph", dict(device_sampling=True, graph_step=True))):
    if ONLY and ONLY != tag:
        continue
    model = DIN("ranking", info, embed_size=128, n_epochs=1, lr=1e-3, batch_size=8192, num_neg=1, hidden_units=(128, 64, 32),
                recent_num=50, sampler="random", seed=3, **kw)
    model.fit(train, neg_sampling=True, verbose=0)
    assert model.net._fstep is not None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.trainer.run(train, True, 0, True, None, None, 10, 8192, None, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = -(-len(train) // 4096)
    print(f"{tag:28s}: epoch {dt:6.2f} s = {2 * len(train) / dt / 1e6:6.2f} M samples/s  ({dt / steps * 1e3:6.2f} ms per step of 8,192 samples)", flush=True)
    tables[tag] = model.net.tables.embed.clone()
    del model
    torch.cuda.empty_cache()
if "device loader, eager" in tables and "device loader, hipGraph" in tables:
    print("device loader: graph == eager tables:", bool(torch.equal(tables["device loader, eager"], tables["device loader, hipGraph"])), flush=True)
