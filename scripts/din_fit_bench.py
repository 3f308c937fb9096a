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
    dt, setup = timed_epochs(model, train)
    steps = -(-len(train) // 4096)
    print(f"{tag:28s}: epoch {dt:6.2f} s = {2 * len(train) / dt / 1e6:6.2f} M samples/s  ({dt / steps * 1e3:6.2f} ms per step of 8,192 samples; loader set-up {setup:.2f} s once per fit)", flush=True)
    tables[tag] = model.net.tables.embed.clone()
    del model
    torch.cuda.empty_cache()
if "device loader, eager" in tables and "device loader, hipGraph" in tables:
    print("device loader: graph == eager tables:", bool(torch.equal(tables["device loader, eager"], tables["device loader, hipGraph"])), flush=True)
