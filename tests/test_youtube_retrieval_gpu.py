"""YouTubeRetrieval on the HIP path (`-m gpu`): the graph + sampled losses against the fp64 oracle
restatement from identical weights and an identical candidate set, and the reference's behavioural model checks
(`tests/models/test_youtube_retrieval.py` shape: fit / predict / recommend / dynamic embedding / save + load)."""
import numpy as np
import pytest
import torch

from librecommender_amd.algorithms import YouTubeRetrieval
from librecommender_amd.data import DatasetFeat, DatasetPure, split_by_ratio_chrono
from librecommender_amd.nets import FeatSpec
from librecommender_amd.nets.youtube_nets import YouTubeRetrievalNet
from oracle.make_golden import synthetic_frame
from oracle.models_torch import YouTubeRetrievalOracle, export_retrieval_weights
from tests.test_api_gpu import check_preds, check_recommends, movielens_like

pytestmark = pytest.mark.gpu


def T(x):
    t = torch.from_numpy(np.ascontiguousarray(x))
    return t.long() if t.dtype in (torch.int32, torch.int64) else t


def close(got, ref, name, rtol=1e-4, atol=3e-6):
    np.testing.assert_allclose(got.numpy().reshape(ref.shape), ref.detach().numpy(), rtol=rtol, atol=atol, err_msg=name)


def batch(rng, B, N, L, n_sp, vocab, n_dense):
    items = rng.integers(0, N, B)
    seqs = np.full((B, L), N, dtype=np.int64)
    for b in range(B):
        n = rng.integers(0, L + 1)
        seqs[b, :n] = rng.integers(0, N, n)
    seqs[0] = N                                                    # no history: zero vector
    seqs[1, :2] = seqs[1, 0] if seqs[1, 0] != N else 3             # a repeated item inside one bag
    sparse = (rng.integers(0, vocab, (B, n_sp)) + np.arange(n_sp) * (vocab + 1)) if n_sp else None
    dense = rng.standard_normal((B, n_dense)).astype(np.float32) if n_dense else None
    S = 24
    sampled = rng.permutation(N)[:S]
    sampled[:3] = items[:3]                                        # accidental hits (rows 0..2 see their own label)
    sampled = np.unique(sampled)
    return items, seqs, sparse, dense, sampled


@pytest.mark.parametrize("loss_type,norm,n_sp,n_dense", [("sampled_softmax", False, 0, 0), ("sampled_softmax", True, 2, 1),
                                                         ("nce", False, 2, 1), ("nce", True, 0, 0)])
def test_graph_and_sampled_losses_against_oracle(dev, loss_type, norm, n_sp, n_dense):
    rng = np.random.default_rng(11)
    N, L, K, vocab = 70, 5, 16, 6
    spec = FeatSpec(0, N, n_sp, n_sp * (vocab + 1), n_dense)
    net = YouTubeRetrievalNet(N, spec, K, (32, K), use_bn=True, norm_embed=norm, max_seq_len=L, lr=1e-2, device=dev,
                              dense_adam=True, loss_type=loss_type)
    o = YouTubeRetrievalOracle(export_retrieval_weights(net), N, (32, K), True, norm, loss_type, lr=1e-2,
                               dtype=torch.float64)
    batches = [batch(rng, 40, N, L, n_sp, vocab, n_dense) for _ in range(3)]
    it, sq, sp, de, smp = batches[0]
    ue = net.embed_users(sq, sp, de).cpu().numpy()
    ref = o.user_embeds(T(sq), None if sp is None else T(sp), None if de is None else T(de)).detach().numpy()
    np.testing.assert_allclose(ue, ref, rtol=1e-5, atol=1e-5)
    for it, sq, sp, de, smp in batches:
        l_hip = float(net.train_step(it, sq, sp, de, sampled=torch.as_tensor(smp)))
        l_ref = float(o.train_step(T(it), T(sq), None if sp is None else T(sp), None if de is None else T(de), T(smp)))
        assert abs(l_hip - l_ref) < 2e-5 * max(1.0, abs(l_ref))
    W = export_retrieval_weights(net)
    for name, ref in o.V.v.items():
        close(W[name], ref, name)
    for name, ref in o.V.buffers.items():
        close(W[name], ref, name, atol=1e-6)


def test_candidate_sampler_is_unique_and_uniform(dev):
    net = YouTubeRetrievalNet(50, FeatSpec(0, 50), 16, (16,), device=dev)
    counts = torch.zeros(50)
    for _ in range(400):
        s = net.draw_sampled(10).cpu().long()
        assert len(torch.unique(s)) == 10 and int(s.min()) >= 0 and int(s.max()) < 50
        counts[s] += 1
    assert counts.min() > 40 and counts.max() < 125              # inclusion probability 0.2 per draw of 10


@pytest.mark.parametrize("data", ["pure", "user_feats", "multi_sparse"])
@pytest.mark.parametrize("loss_type,norm", [("sampled_softmax", False), ("nce", True)])
def test_youtube_retrieval_api(dev, data, loss_type, norm, tmp_path):
    if data == "pure":
        df = movielens_like(3000, 120, 90)
        train, evald = split_by_ratio_chrono(df, test_size=0.2)
        train_data, info = DatasetPure.build_trainset(train)
        eval_data = DatasetPure.build_evalset(evald)
    else:
        df = synthetic_frame()
        train, evald = split_by_ratio_chrono(df, test_size=0.2)
        if data == "user_feats":
            kw = dict(sparse_col=["sex", "occupation"], dense_col=["age"], user_col=["sex", "occupation", "age"], item_col=[])
        else:   # a multi-sparse USER field pooled with sqrtn
            kw = dict(sparse_col=["sex"], multi_sparse_col=[["genre1", "genre2", "genre3"]], dense_col=["age"],
                      user_col=["sex", "genre1", "genre2", "genre3", "age"], item_col=[], pad_val=["missing"])
        train_data, info = DatasetFeat.build_trainset(train, **kw)
        eval_data = DatasetFeat.build_evalset(evald)
    model = YouTubeRetrieval("ranking", info, loss_type=loss_type, norm_embed=norm, embed_size=16, n_epochs=2, lr=1e-2,
                             batch_size=64, hidden_units=(32,), num_sampled_per_batch=20, recent_num=6)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "precision", "ndcg"])
    assert model.user_embeds.shape == (info.n_users + 1, 17) and model.item_embeds.shape == (info.n_items + 1, 17)
    check_preds(model, train)
    check_recommends(model, info, train)
    u = train.user.iloc[5]
    dyn = model.recommend_user(user=u, n_rec=7, seq=[train.item.iloc[0], train.item.iloc[1], -123])
    assert len(dyn[u]) == 7
    cold = model.recommend_user(user="never seen", n_rec=5, seq=[train.item.iloc[2]])
    assert len(cold["never seen"]) == 5
    if data != "pure":
        feats = model.recommend_user(user=u, n_rec=4, user_feats={"sex": "female", "age": 33})
        assert len(feats[u]) == 4
    e1 = model.dyn_user_embedding(u)
    np.testing.assert_allclose(e1, model.get_user_embedding(u), rtol=1e-5, atol=1e-6)   # cached window == consumed tail
    assert model.dyn_user_embedding(u, include_bias=True).shape == (17,)
    with pytest.raises(ValueError):
        model.recommend_user(user=[u, u], n_rec=3, seq=[1])
    model.save(str(tmp_path), "ytr")
    loaded = YouTubeRetrieval.load(str(tmp_path), "ytr", info)
    i = train.item.iloc[5]
    np.testing.assert_allclose(loaded.predict(user=u, item=i), model.predict(user=u, item=i), rtol=1e-6)
    np.testing.assert_array_equal(loaded.recommend_user(user=u, n_rec=5)[u], model.recommend_user(user=u, n_rec=5)[u])
    np.testing.assert_allclose(loaded.dyn_user_embedding(u), e1, rtol=1e-6)


def test_youtube_retrieval_errors(dev):
    from oracle.make_golden import FEAT_KW

    df = synthetic_frame()
    _, info_items = DatasetFeat.build_trainset(df, **FEAT_KW)
    with pytest.raises(ValueError):
        YouTubeRetrieval("ranking", info_items)                    # item features are not allowed
    _, info = DatasetPure.build_trainset(movielens_like(500, 30, 40))
    with pytest.raises(AssertionError):
        YouTubeRetrieval("rating", info)
    with pytest.raises(ValueError):
        YouTubeRetrieval("ranking", info, loss_type="bpr")
