"""Value-level check of the OOV assignment of the TF-family models — `-m gpu`.

After `fit` the reference overwrites the padding rows (`bases/tf_base.py:310-353`,
`assign_tf_variables_oov`): user row `n_users` := mean of rows [0, n_users), item row `n_items` := mean
of rows [0, n_items), and per sparse field the row `sparse_oov[f]` := mean of that field's slice
`[start, oov)` with `start` advancing to `oov + 1` (columns of one multi-sparse field share a slice and
are visited once).  Every user / item / sparse variable is treated alike (embedding and linear
twins).  The expectation below restates that loop in numpy over the fitted tables read back from the
device; nothing of the product's own `assign_oov` is reused.
"""
import numpy as np
import pytest

from librecommender_amd.algorithms import DIN, FM, DeepFM
from librecommender_amd.data import DatasetFeat, split_by_ratio_chrono
from oracle.make_golden import FEAT_KW, MULTI_KW, synthetic_frame

pytestmark = pytest.mark.gpu

PLAIN_KW = dict(sparse_col=["sex", "occupation", "genre1", "genre2", "genre3"],
                user_col=["sex", "occupation"], item_col=["genre1", "genre2", "genre3"])


def reference_oov(var: np.ndarray, kind: str, n_users: int, n_items: int, sparse_oov) -> dict:
    """{row: expected value} per tf_base.py:319-351 for one variable of `kind`."""
    v64 = var.astype(np.float64)
    if kind == "user":
        return {n_users: v64[:n_users].mean(axis=0)}
    if kind == "item":
        return {n_items: v64[:n_items].mean(axis=0)}
    out, start = {}, 0
    for oov in sparse_oov:
        oov = int(oov)
        if start >= oov:
            continue
        out[oov] = v64[start:oov].mean(axis=0)
        start = oov + 1
    return out


def tables_of(model):
    net = model.net
    return net.emb.tables if hasattr(net, "emb") else net.tables


@pytest.mark.parametrize("cls,extra", [(FM, {}), (DeepFM, {"hidden_units": (32, 16)}),
                                       (DIN, {"hidden_units": (32, 16), "recent_num": 6})])
@pytest.mark.parametrize("kw", [PLAIN_KW, FEAT_KW, MULTI_KW], ids=["plain", "feat", "multi"])
def test_oov_rows_are_the_means_of_the_real_rows(dev, cls, extra, kw):
    df = synthetic_frame()
    train, _ = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetFeat.build_trainset(train_data=train, **kw)
    model = cls("ranking", info, embed_size=16, n_epochs=1, lr=1e-2, batch_size=64, num_neg=1, **extra)
    model.fit(train_data, neg_sampling=True, verbose=0)
    t = tables_of(model)
    sparse_oov = list(info.sparse_oov) if info.sparse_oov is not None else []
    checked = 0
    for kind in ("user", "item", "sparse"):
        for which in ("embeds_var", "linear_var"):
            if which == "linear_var" and t.lin is None:
                continue
            var = t.variable(f"{kind}_{which}").detach().cpu().numpy()
            if var.shape[0] == 0:
                continue
            var = var.reshape(var.shape[0], -1)
            want = reference_oov(var, kind, info.n_users, info.n_items, sparse_oov)
            if kind == "sparse":
                assert len(want) >= 1
            for row, val in want.items():
                assert row < var.shape[0]
                # fp32 mean on the device vs fp64 mean here: n <= a few hundred rows of |w| < 1
                np.testing.assert_allclose(var[row], val, rtol=1e-5, atol=1e-6,
                                           err_msg=f"{cls.__name__} {kind}_{which} row {row}")
                assert np.abs(var[row]).max() > 0          # trained tables: the mean is not trivially zero
                checked += 1
    assert checked >= 4
    # the padding user / item are what an unknown id maps to (tf_base.py:283-308): predicting for it
    # must therefore be finite and use those rows
    p = model.predict(user="no such user", item=train.item.iloc[0], cold_start="average")
    assert np.isfinite(p).all()
