"""Model API on the HIP path (`-m gpu`): the reference's behavioural checkers
(`tests/utils_pred.py:6-27`, `tests/utils_reco.py:14-78`) applied to the drop-in classes, plus the
exception matrices of `tests/models/test_two_tower.py`, on a synthetic movielens-shaped frame.  BASELINE config 1
on the named file (`sample_movielens_rating.dat`) is `tests/test_cfg1_movielens_gpu.py`."""
import numpy as np
import pandas as pd
import pytest

from librecommender_amd.algorithms import TwoTower
from librecommender_amd.data import DatasetFeat, DatasetPure, split_by_ratio_chrono
from librecommender_amd.evaluation import evaluate
from oracle.make_golden import FEAT_KW, synthetic_frame

pytestmark = pytest.mark.gpu


def movielens_like(n=20_000, n_users=600, n_items=900, seed=0):
    rng = np.random.default_rng(seed)
    u = rng.zipf(1.3, n) % n_users
    i = rng.zipf(1.2, n) % n_items
    return pd.DataFrame({"user": u, "item": i, "label": rng.integers(1, 6, n), "time": rng.integers(0, 10**6, n)})


def consumed_raw(data_info, user):
    uid = data_info.user2id[user]
    return {data_info.id2item[i] for i in data_info.user_consumed[uid]}


def check_preds(model, pd_data):
    user, item = pd_data.user.iloc[0], pd_data.item.iloc[0]
    pred = model.predict(user=user, item=item)
    assert 0 <= pred <= 1
    pop = model.predict(user="cold user2", item="cold item2", cold_start="popular")
    assert np.allclose(pop, model.default_pred)
    assert model.predict(user="cold user1", item="cold item2") == model.predict(user="cold user2", item="cold item2")


def check_recommends(model, data_info, pd_data):
    with pytest.raises(ValueError):
        model.recommend_user(user=-99999, n_rec=7, cold_start="sss")
    users = pd_data.user.tolist()
    u1, u2 = users[0], users[1]
    r1 = model.recommend_user(user=u1, n_rec=7)[u1]
    r2 = model.recommend_user(user=u2, n_rec=7)[u2]
    assert len(r1) == len(r2) == 7
    assert not (set(r1.tolist()) & consumed_raw(data_info, u1))
    assert not (set(r2.tolist()) & consumed_raw(data_info, u2))
    assert len(model.recommend_user(user=-1, n_rec=10, cold_start="popular")[-1]) == 10
    for cold in (-99999, -1):
        rec = model.recommend_user(user=cold, n_rec=3)[cold]
        assert len(rec) == 3
        assert np.all(np.isin([data_info.item2id[i] for i in rec], model.default_recs))
    a, b, c = users[2], users[3], users[4]
    rr = model.recommend_user(user=a, n_rec=10, filter_consumed=False, random_rec=True)
    nr = model.recommend_user(user=a, n_rec=10, filter_consumed=False, random_rec=False)
    assert len(rr[a]) == len(nr[a]) == 10
    batch = model.recommend_user(user=[a, b, c, -1], n_rec=3, filter_consumed=True, random_rec=False, cold_start="popular")
    assert len(batch[a]) == len(batch[b]) == len(batch[c]) == 3
    assert np.all(np.isin(batch[-1], data_info.popular_items))


def test_baseline_config1_two_tower_pure(dev):
    """BASELINE.json configs[0]: TwoTower(embed_size=16) on pure data, fit + recommend_user."""
    df = movielens_like()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetPure.build_trainset(train)
    eval_data = DatasetPure.build_evalset(evald)
    model = TwoTower("ranking", info, loss_type="softmax", embed_size=16, n_epochs=2, lr=1e-3,
                     batch_size=2048, use_bn=True, hidden_units=(64, 32))
    model.fit(train_data, neg_sampling=True, verbose=2, shuffle=True, eval_data=eval_data,
              metrics=["roc_auc", "precision", "recall", "ndcg"], eval_user_num=200)
    assert model.user_embeds.shape == (info.n_users + 1, 32)
    assert model.item_embeds.shape == (info.n_items + 1, 32)
    check_preds(model, train)
    check_recommends(model, info, train)
    res = evaluate(model, eval_data, neg_sampling=True, metrics=["roc_auc", "ndcg", "map"], k=10, sample_user_num=100)
    assert 0.0 <= res["roc_auc"] <= 1.0 and 0.0 <= res["ndcg"] <= 1.0
    # recommend_user agrees with a numpy restatement on the exported embeddings
    from oracle import ops_np
    u_ids = list(range(5))
    want, _ = ops_np.recommend_from_embedding(model.user_embeds_np, model.item_embeds_np, u_ids, 10,
                                              info.n_items, info.user_consumed, True)
    got = model.recommend_user(user=u_ids, n_rec=10, inner_id=True)
    for u in u_ids:
        assert len(set(got[u].tolist()) & set(want[u].tolist())) >= 9  # near-ties may swap at the cut


@pytest.mark.parametrize("loss_type,sampler,num_neg,norm,bn", [
    ("cross_entropy", "random", 1, False, True), ("cross_entropy", "unconsumed", 2, True, False),
    ("max_margin", "popular", 2, False, True), ("softmax", "random", 1, True, True),
])
def test_two_tower_with_features(dev, loss_type, sampler, num_neg, norm, bn):
    lr_decay = loss_type == "max_margin"
    df = synthetic_frame()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetFeat.build_trainset(train_data=train, **FEAT_KW)
    eval_data = DatasetFeat.build_testset(evald)
    model = TwoTower("ranking", info, loss_type=loss_type, embed_size=16, norm_embed=norm, n_epochs=1,
                     lr=1e-3, batch_size=64, sampler=sampler, num_neg=num_neg, use_bn=bn,
                     hidden_units=(32, 16), remove_accidental_hits=True, lr_decay=lr_decay)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "precision"])
    check_preds(model, train)
    check_recommends(model, info, train)
    u3 = train.user.tolist()[4]
    dyn = model.recommend_user(user=u3, n_rec=7, user_feats={"sex": "male", "occupation": "b", "age": 23})
    assert len(dyn[u3]) == 7
    with pytest.raises(ValueError):
        model.recommend_user(user=[u3, u3], n_rec=7, user_feats={"sex": "male"})
    with pytest.raises(ValueError):
        model.recommend_user(user=u3, n_rec=7, seq=[1, 2, 3])
    emb = model.dyn_user_embedding(u3, user_feats={"sex": "female"})
    assert emb.shape == (16,)


def test_two_tower_argument_errors(dev):
    df = synthetic_frame()
    train, _ = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetPure.build_trainset(train)
    with pytest.raises(ValueError):
        TwoTower("rating", info)
    with pytest.raises(ValueError):
        TwoTower("ranking", info, loss_type="whatever")
    with pytest.raises(ValueError):
        TwoTower("ranking", info, ssl_pattern="rfm")            # needs item sparse features
    m = TwoTower("ranking", info, loss_type="max_margin", n_epochs=1)
    with pytest.raises(ValueError):
        m.fit(train_data, neg_sampling=False)                    # pairwise loss needs sampling
    with pytest.raises(AssertionError):
        m.fit(train_data, neg_sampling="yes")


def test_two_tower_save_load_roundtrip(dev, tmp_path):
    df = movielens_like(5000, 200, 300)
    train, _ = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetPure.build_trainset(train)
    model = TwoTower("ranking", info, embed_size=16, n_epochs=1, batch_size=512, hidden_units=(32,))
    model.fit(train_data, neg_sampling=True, verbose=0)
    u = train.user.iloc[3]
    before = model.recommend_user(user=u, n_rec=7)[u]
    model.save(str(tmp_path), "tt")
    loaded = TwoTower.load(str(tmp_path), "tt", info)
    np.testing.assert_array_equal(loaded.recommend_user(user=u, n_rec=7)[u], before)
    with pytest.raises(RuntimeError):
        loaded.fit(train_data, neg_sampling=True)


@pytest.mark.parametrize("pattern", ["rfm", "rfm-complementary", "cfm"])
def test_two_tower_ssl_patterns_fit(dev, pattern):
    """`ssl_pattern` through the class: feature masking per batch (feature/ssl.py), mutual-information
    table for `cfm`, extra in-batch softmax term; constructor checks of two_tower.py:173-187."""
    df = synthetic_frame()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetFeat.build_trainset(train_data=train, **FEAT_KW)
    eval_data = DatasetFeat.build_testset(evald)
    model = TwoTower("ranking", info, loss_type="softmax", embed_size=16, n_epochs=2, lr=1e-3, batch_size=64,
                     hidden_units=(32, 16), ssl_pattern=pattern, alpha=0.3)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    if pattern == "cfm":
        assert len(model.sparse_feat_mutual_info) == 1 + len(info.item_sparse_col.name)
    check_recommends(model, info, train)
    with pytest.raises(ValueError):
        TwoTower("ranking", info, loss_type="softmax", ssl_pattern="nope")
    with pytest.raises(ValueError):
        TwoTower("ranking", info, loss_type="cross_entropy", ssl_pattern="rfm")
