"""BASELINE.json cfg 1 on the NAMED dataset (`-m gpu`): pure-CF `TwoTower(embed_size=16)` fit + `recommend_user` on
`examples/sample_data/sample_movielens_rating.dat` (100 k interactions; committed as a data fixture under
tests/golden/) through `DatasetPure` (SURVEY 8d cfg 1: "plumbing only — same API results, metrics within noise").

The HIP model and `TwoTowerOracle` (PyTorch-CPU restatement of algorithms/two_tower.py:113-139,189-410 with TF1 Adam)
start from identical weights and consume the IDENTICAL batches (every batch the trainer hands to the model is also
fed to the oracle).  With the reference's optimiser semantics (`dense_adam=True`: every table row moves every step)
the exported user / item embeddings agree after two epochs, `recommend_user` equals the numpy definition
(recommendation/recommend.py:57-78, ranking.py:10-56) on the exported embeddings, and the evaluation metrics computed
from the two sets of embeddings agree within noise.  The default row-wise Adam run is fitted too (finite metrics, better
than chance)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from librecommender_amd.algorithms import TwoTower
from librecommender_amd.data import DatasetPure, split_by_ratio_chrono
from librecommender_amd.evaluation import evaluate
from oracle import ops_np
from oracle.models_torch import TwoTowerOracle, export_net_weights

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "golden", "sample_movielens_rating.dat")
HID = (32, 16)


@pytest.fixture(scope="module")
def data():
    df = pd.read_csv(DATA, sep="::", engine="python", names=["user", "item", "label", "time"])
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetPure.build_trainset(train)
    eval_data = DatasetPure.build_evalset(evald)
    assert len(df) == 100_000 and info.n_users > 5000 and info.n_items > 3000
    return train, train_data, eval_data, info


def _model(info, **kw):
    return TwoTower("ranking", info, loss_type="softmax", embed_size=16, n_epochs=2, lr=1e-3, batch_size=2048,
                    hidden_units=HID, use_bn=True, seed=42, **kw)


def test_fit_matches_oracle_on_identical_batches(dev, data):
    import random

    train, train_data, eval_data, info = data
    model = _model(info)
    model.build_model()
    model.model_built = True
    model.net.dense_adam = True                       # the reference's optimiser: tf.train.AdamOptimizer moves every row
    o = TwoTowerOracle(export_net_weights(model.net), HID, use_bn=True, lr=1e-3, dtype=torch.float64)
    orig = model.train_on_batch
    seen = []

    def both(b):
        users, items = np.asarray(b.users), np.asarray(b.items)
        corr = torch.from_numpy(np.asarray(model.item_corrections)[items])
        seen.append(float(o.train_step("softmax", torch.from_numpy(users).long(), torch.from_numpy(items).long(), corrections=corr)))
        loss = orig(b)
        seen.append(float(loss))
        return loss

    model.train_on_batch = both
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    model.fit(train_data, neg_sampling=True, verbose=0, shuffle=True)
    ref_l, hip_l = np.array(seen[0::2]), np.array(seen[1::2])
    assert len(ref_l) == 2 * -(-len(train_data) // 2048)
    np.testing.assert_allclose(hip_l, ref_l, rtol=2e-4, atol=2e-4)                 # the same loss trajectory, batch by batch
    # exported embeddings (dyn_embed_base.py:240-269): tower outputs of every known id
    ue = o.user_embeds(torch.arange(info.n_users)).detach().numpy()
    ie = o.item_embeds(torch.arange(info.n_items)).detach().numpy()
    # (fp32 vs fp64 over ~80 Adam steps through two BatchNorms: 1e-2 of the largest entry; the loss trajectory above is
    # the tight check)
    np.testing.assert_allclose(model.user_embeds[: info.n_users].cpu().numpy(), ue, rtol=1e-2, atol=1e-2 * np.abs(ue).max())
    np.testing.assert_allclose(model.item_embeds[: info.n_items].cpu().numpy(), ie, rtol=1e-2, atol=1e-2 * np.abs(ie).max())
    # recommend_user == the numpy definition on the model's own exported embeddings
    users = list(range(0, info.n_users, 97))
    U, I = model.user_embeds.cpu().numpy(), model.item_embeds.cpu().numpy()
    ref_ids, ref_s = ops_np.recommend_from_embedding(U, I[: info.n_items], users, 10, info.n_items, info.user_consumed, True)
    got = model.recommend_user(users, 10, inner_id=True)
    full = U[users] @ I[: info.n_items].T
    for j, u in enumerate(users):
        g_ids = got[u]
        assert not set(g_ids.tolist()) & set(info.user_consumed[u])
        s = np.sort(full[j])[::-1]
        # ids must agree wherever neighbouring scores are separated by more than fp32 rounding of a 16-term dot product
        sc = full[j][ref_ids[j]]
        sep = np.abs(np.diff(np.concatenate([[np.inf], sc, [-np.inf]]))) > 1e-5
        ok = sep[:-1] & sep[1:]
        np.testing.assert_array_equal(g_ids[ok], ref_ids[j][ok])
        assert s[0] >= sc[0] - 1e-6
    # metrics from the two sets of embeddings agree within noise
    res = evaluate(model, eval_data, neg_sampling=True, metrics=["loss", "roc_auc", "precision", "recall", "ndcg"], k=10, seed=1)
    model.user_embeds = torch.cat([torch.from_numpy(ue), torch.from_numpy(ue).mean(0, keepdim=True)]).float().to(dev)
    model.item_embeds = torch.cat([torch.from_numpy(ie), torch.from_numpy(ie).mean(0, keepdim=True)]).float().to(dev)
    res_o = evaluate(model, eval_data, neg_sampling=True, metrics=["loss", "roc_auc", "precision", "recall", "ndcg"], k=10, seed=1)
    for k in res:
        assert abs(res[k] - res_o[k]) <= 2e-3 + 2e-2 * abs(res_o[k]), (k, res[k], res_o[k])
    assert res["roc_auc"] > 0.52                      # two epochs at lr 1e-3: better than chance is all this asks


def test_default_fit_and_api(dev, data):
    """The default product path (row-wise Adam on the touched rows): fit, evaluate, predict, recommend, cold start."""
    train, train_data, eval_data, info = data
    model = _model(info)
    model.fit(train_data, neg_sampling=True, verbose=0, shuffle=True)
    res = evaluate(model, eval_data, neg_sampling=True, metrics=["loss", "roc_auc", "precision"], k=10, seed=1)
    assert np.isfinite(res["loss"]) and res["roc_auc"] > 0.52
    u, i = train.user.iloc[0], train.item.iloc[0]
    assert 0.0 <= float(model.predict(u, i)) <= 1.0
    rec = model.recommend_user(u, 7)[u]
    assert len(rec) == 7 and not set(rec.tolist()) & {info.id2item[x] for x in info.user_consumed[info.user2id[u]]}
    assert len(model.recommend_user("nobody", 5, cold_start="popular")["nobody"]) == 5
