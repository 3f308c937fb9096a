import cProfile, pstats, io, os, sys, time
from pathlib import Path
import numpy as np, pandas as pd, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from librecommender_amd.algorithms import DIN
from librecommender_amd.data import DatasetPure
rng = np.random.default_rng(0)
n, nu, ni = 1_000_000, 100_000, 500_000
df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.zipf(1.15, n) % ni, "label": 1, "time": np.arange(n)})
train, info = DatasetPure.build_trainset(df)
for graph in (False, True):
    model = DIN("ranking", info, embed_size=128, n_epochs=1, lr=1e-3, batch_size=8192, num_neg=1, hidden_units=(128, 64, 32),
                recent_num=50, sampler="random", seed=3, device_sampling=True, graph_step=graph)
    model.fit(train, neg_sampling=True, verbose=0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    model.trainer.run(train, True, 0, True, None, None, 10, 8192, None, 0)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph={graph}: enqueue {t1 - t0:.3f} s, drain {t2 - t1:.3f} s")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l[:160] for l in s.getvalue().splitlines()[:60]))
