"""FM / DeepFM / DIN drop-in classes on the HIP path (`-m gpu`): behavioural checkers of the
reference's model tests on its own synthetic fixtures (plain, dense + multi-sparse data)."""
import numpy as np
import pytest

from librecommender_amd.algorithms import DIN, FM, SIM, DeepFM, Transformer, YouTubeRanking
from librecommender_amd.data import DatasetFeat, split_by_ratio_chrono
from librecommender_amd.nets import DeepFMNet, FeatDeepFMNet, FeatFMNet, FMNet
from oracle.make_golden import FEAT_KW, MULTI_KW, synthetic_frame
from tests.test_api_gpu import check_preds, check_recommends

pytestmark = pytest.mark.gpu

PLAIN_KW = dict(sparse_col=["sex", "occupation", "genre1", "genre2", "genre3"],
                user_col=["sex", "occupation"], item_col=["genre1", "genre2", "genre3"])


def build(kw):
    df = synthetic_frame()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetFeat.build_trainset(train_data=train, **kw)
    return train, train_data, DatasetFeat.build_testset(evald), info


@pytest.mark.parametrize("cls", [FM, DeepFM])
@pytest.mark.parametrize("kw,fused", [(PLAIN_KW, True), (FEAT_KW, False), (MULTI_KW, False)])
def test_fm_family(dev, cls, kw, fused):
    train, train_data, eval_data, info = build(kw)
    extra = {"hidden_units": (32, 16)} if cls is DeepFM else {}
    model = cls("ranking", info, embed_size=16, n_epochs=2, lr=1e-2, batch_size=64, num_neg=2,
                sampler="unconsumed" if fused else "random", lr_decay=not fused, **extra)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "precision", "ndcg"])
    if not fused:   # staircase exponential decay of the TF trainer: lr0 * 0.96 ** (steps // decay_steps)
        assert model.net.lr == pytest.approx(1e-2 * 0.96 ** ((model.net.step - 1) // int(info.data_size / 64)))
    assert isinstance(model.net, (FMNet, DeepFMNet)) == fused
    assert isinstance(model.net, (FeatFMNet, FeatDeepFMNet)) != fused
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[3]
    p = model.predict(user=u, item=train.item.iloc[3], feats={"sex": "male", "genre1": "crime"})
    assert 0 <= p <= 1
    dyn = model.recommend_user(user=u, n_rec=7, user_feats={"sex": "female", "occupation": "c"})
    assert len(dyn[u]) == 7
    with pytest.raises(ValueError):
        model.recommend_user(user=[u, u], n_rec=3, user_feats={"sex": "male"})
    with pytest.raises(ValueError):
        model.recommend_user(user=u, n_rec=3, seq=[1, 2])       # not a sequence model


def test_rating_task_and_errors(dev):
    train, train_data, eval_data, info = build(FEAT_KW)
    model = DeepFM("rating", info, embed_size=16, n_epochs=1, batch_size=64, hidden_units=(16,))
    with pytest.raises(ValueError):
        model.fit(train_data, neg_sampling=True)
    model.fit(train_data, neg_sampling=False, verbose=2, eval_data=eval_data, metrics=["rmse", "mae"])
    p = model.predict(user=train.user.iloc[0], item=train.item.iloc[0])
    assert 1 <= p <= 5
    _, _, _, minfo = build(MULTI_KW)
    with pytest.raises(ValueError):
        FM("ranking", minfo, multi_sparse_combiner="max")


@pytest.mark.parametrize("kw", [PLAIN_KW, FEAT_KW, MULTI_KW])
@pytest.mark.parametrize("pure_items", [True, False])
def test_din(dev, kw, pure_items):
    if pure_items:   # items without side features -> fused attention kernel
        kw = dict(sparse_col=["sex", "occupation"], user_col=["sex", "occupation"], item_col=[])
    train, train_data, eval_data, info = build(kw)
    model = DIN("ranking", info, embed_size=16, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1,
                hidden_units=(32, 16), recent_num=6)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    assert model.net.fused == pure_items
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[5]
    dyn = model.recommend_user(user=u, n_rec=7, seq=[train.item.iloc[0], train.item.iloc[1], -123])
    assert len(dyn[u]) == 7
    cold = model.recommend_user(user="never seen", n_rec=5, seq=[train.item.iloc[2]])
    assert len(cold["never seen"]) == 5
    with pytest.raises(ValueError):
        model.recommend_user(user=[u, u], n_rec=3, seq=[1])
    with pytest.raises(ValueError):
        model.recommend_user(user=u, n_rec=3, seq="abc")


@pytest.mark.parametrize("kw", [PLAIN_KW, MULTI_KW])
@pytest.mark.parametrize("device_sampling", [False, True])
def test_youtube_ranking(dev, kw, device_sampling, tmp_path):
    """Behavioural checks of the reference's `tests/models/test_youtube_ranking.py` shape: fit with evaluation,
    predict, recommend (plain / arbitrary sequence / cold start), save + load."""
    train, train_data, eval_data, info = build(kw)
    model = YouTubeRanking("ranking", info, embed_size=16, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1,
                           hidden_units=(32, 16), recent_num=6, sampler="random", device_sampling=device_sampling)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[5]
    dyn = model.recommend_user(user=u, n_rec=7, seq=[train.item.iloc[0], train.item.iloc[1], -123])
    assert len(dyn[u]) == 7
    cold = model.recommend_user(user="never seen", n_rec=5, seq=[train.item.iloc[2]])
    assert len(cold["never seen"]) == 5
    with pytest.raises(AssertionError):
        YouTubeRanking("rating", info)
    model.save(str(tmp_path), "ytb")
    loaded = YouTubeRanking.load(str(tmp_path), "ytb", info)
    i = train.item.iloc[5]
    np.testing.assert_allclose(loaded.predict(user=u, item=i), model.predict(user=u, item=i), rtol=1e-6)
    np.testing.assert_array_equal(loaded.recommend_user(user=u, n_rec=5)[u], model.recommend_user(user=u, n_rec=5)[u])


@pytest.mark.parametrize("kw,extra", [(PLAIN_KW, dict(num_heads=2)), (FEAT_KW, dict(feat_agg_mode="elementwise", num_tfm_layers=2, use_causal_mask=True)),
                                      (MULTI_KW, dict(positional_embedding="sinusoidal"))])
def test_transformer(dev, kw, extra, tmp_path):
    """Behavioural checks in the shape of the reference's `tests/models/test_transformer.py`."""
    train, train_data, eval_data, info = build(kw)
    model = Transformer("ranking", info, embed_size=16, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1,
                        hidden_units=(32, 16), recent_num=6, **extra)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[5]
    assert len(model.recommend_user(user=u, n_rec=7, seq=[train.item.iloc[0], -123])[u]) == 7
    with pytest.raises(ValueError):
        Transformer("ranking", info, loss_type="bpr")
    with pytest.raises(ValueError):
        Transformer("ranking", info, feat_agg_mode="sum")
    model.save(str(tmp_path), "tfm")
    loaded = Transformer.load(str(tmp_path), "tfm", info)
    i = train.item.iloc[5]
    np.testing.assert_allclose(loaded.predict(user=u, item=i), model.predict(user=u, item=i), rtol=1e-6)


@pytest.mark.parametrize("kw", [PLAIN_KW, FEAT_KW, MULTI_KW])
def test_sim(dev, kw, tmp_path):
    """Behavioural checks in the shape of the reference's `tests/models/test_sim.py`."""
    train, train_data, eval_data, info = build(kw)
    model = SIM("ranking", info, embed_size=16, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1, hidden_units=(32, 16),
                alpha=0.5, beta=1.0, search_topk=3, long_max_len=5, short_max_len=3, num_heads=2)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[5]
    for seq in ([train.item.iloc[0], -123], list(train.item.iloc[:12])):      # shorter / longer than both windows
        assert len(model.recommend_user(user=u, n_rec=7, seq=seq)[u]) == 7
    with pytest.raises(ValueError):
        SIM("ranking", info, loss_type="bpr")
    with pytest.raises(AssertionError):
        SIM("ranking", info, search_topk=200, long_max_len=100)
    model.save(str(tmp_path), "sim")
    loaded = SIM.load(str(tmp_path), "sim", info)
    i = train.item.iloc[5]
    np.testing.assert_allclose(loaded.predict(user=u, item=i), model.predict(user=u, item=i), rtol=1e-6)


def test_save_load_feat_model(dev, tmp_path):
    train, train_data, _, info = build(FEAT_KW)
    model = DeepFM("ranking", info, embed_size=16, n_epochs=1, batch_size=64, hidden_units=(16,))
    model.fit(train_data, neg_sampling=True, verbose=0)
    u, i = train.user.iloc[1], train.item.iloc[1]
    before = model.predict(user=u, item=i)
    model.save(str(tmp_path), "dfm")
    loaded = DeepFM.load(str(tmp_path), "dfm", info)
    np.testing.assert_allclose(loaded.predict(user=u, item=i), before, rtol=1e-6)
    np.testing.assert_array_equal(loaded.recommend_user(user=u, n_rec=5)[u], model.recommend_user(user=u, n_rec=5)[u])


@pytest.mark.parametrize("cls_name,kw", [("FM", {}), ("DeepFM", {"hidden_units": (32, 16)}), ("DeepFM", {"hidden_units": (24,), "use_bn": False}),
                                         ("DeepFM", {}), ("DeepFM", {"hidden_units": (64, 32, 16), "use_bn": False})])   # three relu layers: csrc/pair_mlp.hip
@pytest.mark.parametrize("feat", ["pure", "feat", "multi"])
def test_factorised_catalog_scores_equal_materialised_forward(dev, cls_name, kw, feat):
    """the factorised full-catalog scorer (item side cached, user side per user, MLP
    tail + one dot product per pair) reproduces the model's forward on the materialised B x N
    feature rows — FM / DeepFM, pure ids, plain features, pooled multi-sparse + dense columns."""
    import librecommender_amd.algorithms as A
    from librecommender_amd.data import DatasetFeat, DatasetPure
    from tests.test_api_gpu import movielens_like

    if feat == "pure":
        df = movielens_like(3000, 120, 90)
        train_data, info = DatasetPure.build_trainset(df)
    else:
        df = synthetic_frame()
        train_data, info = DatasetFeat.build_trainset(df, **(FEAT_KW if feat == "feat" else MULTI_KW))
    mkw = dict(kw)
    if feat == "multi":
        mkw["multi_sparse_combiner"] = "sqrtn"
    model = getattr(A, cls_name)("ranking", info, embed_size=16, n_epochs=1, lr=1e-2, batch_size=512, **mkw)
    model.fit(train_data, neg_sampling=True, verbose=0)
    sc = model._catalog_scorer()
    assert sc is not None
    if cls_name == "DeepFM" and len(model.hidden_units) == 3:
        assert sc._fused_tail() is not None                          # the MFMA pair kernel serves these stacks
    uids = [0, 3, min(7, info.n_users - 1), info.n_users]          # incl. the OOV user
    fast = sc.scores(uids).cpu().numpy()
    for r, u in enumerate(uids):
        slow = model._scores_all_items(u, None, None).cpu().numpy()
        np.testing.assert_allclose(fast[r], slow, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(slow).max())))  # fp32 re-association; over-trained toy logits reach 1e3-1e4
    if cls_name == "DeepFM":
        # (round 4) the same scores against the ORACLE on the explicit (user x item) cross product: `recommend_tf_feat`
        # (recommendation/recommend.py:81-105) materialises these rows and runs the TF graph on them; restated in fp64 by
        # `FeatDeepFMOracle.forward` from the trained weights (BatchNorm in inference mode: moving statistics)
        import torch

        from librecommender_amd.bases.feat_base import merge_user_item_feats
        from librecommender_amd.nets import FeatSpec
        from oracle.models_torch import FeatDeepFMOracle, export_fieldnet_weights

        spec = FeatSpec.from_data_info(info, model.multi_sparse_combiner)
        oracle = FeatDeepFMOracle(export_fieldnet_weights(model.net), model.hidden_units, use_bn=model.use_bn, dtype=torch.float64,
                                  plain_cols=spec.plain_cols, fields=list(zip(spec.field_offset, spec.field_len, spec.field_oov)),
                                  combiner=spec.combiner)
        N = info.n_items
        for r, u in enumerate(uids):
            us, its = np.full(N, u), np.arange(N)
            sp, de = merge_user_item_feats(info, us, its)
            want = oracle.forward(torch.from_numpy(us), torch.from_numpy(its),
                                  None if sp is None else torch.from_numpy(sp).long(),
                                  None if de is None else torch.from_numpy(de).double(), training=False).detach().numpy()
            np.testing.assert_allclose(fast[r], want, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(want).max())))
    # recommendations go through the factorised path and respect the consumed filter
    rec = model.recommend_user(user=list(info.id2user[u] for u in (0, 3)), n_rec=7)
    assert all(len(v) == 7 for v in rec.values())


@pytest.mark.parametrize("kw", [FEAT_KW, MULTI_KW])
def test_device_catalog_rows_equal_host_materialised_rows(dev, kw):
    """`recommend_tf_feat` (recommendation/recommend.py:81-105): the (user, item) feature rows assembled on the device
    chunk by chunk give the scores of the rows built on the host by `merge_user_item_feats` + the `user_feats`
    override + the repeated sequence — DIN (attention over (sequence, item) pairs has no factorised scorer)."""
    from librecommender_amd.bases.feat_base import merge_user_item_feats
    from librecommender_amd.feature_override import override_dense, override_sparse

    train, train_data, _, info = build(kw)
    model = DIN("ranking", info, embed_size=16, n_epochs=1, lr=1e-2, batch_size=64, num_neg=1,
                hidden_units=(32, 16), recent_num=6)
    model.fit(train_data, neg_sampling=True, verbose=0)
    model.score_chunk = 37                                        # several ragged chunks over the toy catalogue
    N = info.n_items
    for uid, feats, seq in ((3, None, None), (5, {"sex": "female", "occupation": "c"}, None),
                            (info.n_users, None, [train.item.iloc[0], train.item.iloc[4]])):
        got = model._scores_all_items(uid, feats, seq).cpu().numpy()
        items, users = np.arange(N), np.full(N, uid)
        sparse, dense = merge_user_item_feats(info, users, items)
        if feats is not None:
            sparse = override_sparse(info, sparse, feats) if sparse is not None else None
            dense = override_dense(info, dense, feats) if dense is not None else None
        s1, l1 = model._seq_for(uid, seq)
        want = model._forward(users, items, sparse, dense, np.repeat(s1, N, axis=0), np.repeat(l1, N)).cpu().numpy()
        # same rows through the same kernels; only the GEMM tiling may differ with the chunk size
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())))


def test_catalog_caches_follow_assign_item_features(dev):
    """`DataInfo.assign_item_features` rewrites the per-item rows in place: the cached item side of the factorised
    scorer and the device copy of the rows must follow (scores == the materialised forward afterwards)."""
    import pandas as pd

    train, train_data, _, info = build(FEAT_KW)
    model = DeepFM("ranking", info, embed_size=16, n_epochs=1, batch_size=64, hidden_units=(16,))
    model.fit(train_data, neg_sampling=True, verbose=0)
    before = model._catalog_scorer().scores([2]).cpu().numpy()
    col = info.item_sparse_col.name[0]
    vals = list(info.sparse_unique_vals[col])
    items = [info.id2item[i] for i in range(info.n_items)]
    cur = info.item_sparse_unique[: info.n_items, 0] - info.sparse_offset[info.item_sparse_col.index[0]]
    new = pd.DataFrame({"item": items, col: [vals[(int(c) + 1) % len(vals)] for c in cur]})
    info.assign_item_features(new)
    after = model._catalog_scorer().scores([2]).cpu().numpy()
    slow = model._scores_all_items(2, None, None).cpu().numpy()
    assert np.abs(after - before).max() > 0
    np.testing.assert_allclose(after[0], slow, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(slow).max())))
