"""`fit()` with the fused step replayed as a hipGraph (`graph_step=True`, the default) — regression test of the
device-loader + graph-replay memory fault of round 2 (profiles/r02_fit_bench.md): three epochs with a short last batch,
batches from the host loader and from the device loader, replays on the dedicated stream of
nets/din_fused.py:GraphRunner.  The replayed run must equal the eager run of the same kernels bit for bit
(training/tf_trainer.py:76-101: one `sess.run` per step — the launch mechanism must not change a result)."""
import numpy as np
import pandas as pd
import pytest
import torch

from librecommender_amd.algorithms import DIN, DeepFM
from librecommender_amd.data import DatasetFeat, DatasetPure

pytestmark = pytest.mark.gpu


def feat_frame(n=9000, nu=700, ni=400, nf=3, seed=0):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.zipf(1.3, n) % ni, "label": 1,
                       "time": np.arange(n)})
    ucols, icols = [f"u{c}" for c in range(nf)], [f"i{c}" for c in range(nf)]
    for c in ucols:
        df[c] = rng.integers(0, 30, nu)[df["user"].values]
    for c in icols:
        df[c] = rng.integers(0, 30, ni)[df["item"].values]
    return df, ucols, icols


def _fit(cls, info, train, graph_step, device_sampling, **kw):
    model = cls("ranking", info, embed_size=64, n_epochs=3, lr=1e-3, batch_size=2048, num_neg=1,
                hidden_units=(128, 64, 32), sampler="random", device_sampling=device_sampling, graph_step=graph_step,
                seed=7, **kw)
    import random

    random.seed(1)                      # the host samplers / sequence builders draw from the global generators
    np.random.seed(1)
    torch.manual_seed(1)
    model.fit(train, neg_sampling=True, verbose=0, shuffle=True)
    torch.cuda.synchronize()
    return model


@pytest.mark.parametrize("device_sampling", [False, True])
def test_deepfm_fit_graph_equals_eager(dev, device_sampling):
    df, ucols, icols = feat_frame()
    train, info = DatasetFeat.build_trainset(df, user_col=ucols, item_col=icols, sparse_col=ucols + icols, dense_col=[])
    assert len(train) * 2 % 2048 != 0                                       # a short last batch every epoch
    g = _fit(DeepFM, info, train, True, device_sampling)
    assert g.net.fused_l1 and g.net.hip_tail and g.net._use_graph
    n_graphs = sum("graph" in st for st in g.net._graphs.values())
    assert n_graphs == 2, "one graph per batch shape (full and short batch) expected"
    e = _fit(DeepFM, info, train, False, device_sampling)
    assert not getattr(e.net, "_use_graph", False)
    for name in ("embed", "m", "v", "lin"):
        assert torch.equal(getattr(g.net.tables, name), getattr(e.net.tables, name)), name
    assert torch.equal(g.net.P.flat, e.net.P.flat)
    assert bool(torch.isfinite(g.net.P.flat).all())


@pytest.mark.parametrize("device_sampling", [False, True])
def test_din_fit_graph_equals_eager(dev, device_sampling):
    df, _, _ = feat_frame(n=7000, nu=300, ni=500)
    train, info = DatasetPure.build_trainset(df[["user", "item", "label", "time"]])
    kw = dict(recent_num=12)
    g = _fit(DIN, info, train, True, device_sampling, **kw)
    assert g.net._fstep is not None and g.net.graph_step
    assert sum("graph" in st for st in g.net._fstep.runner.graphs.values()) >= 1
    e = _fit(DIN, info, train, False, device_sampling, **kw)
    for name in ("embed", "m", "v"):
        assert torch.equal(getattr(g.net.tables, name), getattr(e.net.tables, name)), name
    assert torch.equal(g.net.P.flat, e.net.P.flat)
    rec_g = g.recommend_user(info.id2user[0], 7)
    rec_e = e.recommend_user(info.id2user[0], 7)
    np.testing.assert_array_equal(rec_g[info.id2user[0]], rec_e[info.id2user[0]])
