"""Generate tests/golden/*.npz by running the REFERENCE's own numpy/torch code on seeded inputs.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference, read-only); the
resulting small fixtures are committed so the GPU box never needs the reference.
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden
Each fixture stores inputs and the reference's outputs; tests compare (i) the oracle restatement
and (ii) the HIP path against them.
"""
from __future__ import annotations

import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"


def _obj(**kw):
    return type("Obj", (), kw)()


def gen_rank(ref):
    """recommendation/ranking.py:10-56 and recommendation/recommend.py:57-78."""
    from libreco.recommendation.ranking import rank_recommendations
    from libreco.recommendation.recommend import recommend_from_embedding

    rng = np.random.default_rng(42)
    cases = {}
    for ci, (B, N, D, k) in enumerate([(5, 50, 8, 7), (16, 3231, 16, 10), (3, 400, 32, 399)]):
        U = rng.standard_normal((B + 4, D)).astype(np.float32)
        I = rng.standard_normal((N + 1, D)).astype(np.float32)  # last row = OOV, excluded
        users = rng.choice(B + 4, size=B, replace=False).tolist()
        consumed = {}
        for u in users:
            c = rng.integers(0, N, size=int(rng.integers(0, 12))).tolist()
            if c:
                consumed[u] = c
        consumed[users[0]] = list(range(N))  # unfilterable user (ranking.py:38)
        model = _obj(task="ranking", n_items=N, user_consumed=consumed)
        for filt in (True, False):
            ids = recommend_from_embedding(model, users, k, U, I, filt, False)
            cases[f"c{ci}_f{int(filt)}_ids"] = np.asarray(ids, dtype=np.int64)
        preds = U[users] @ I[:N].T
        ids, scores = rank_recommendations("ranking", users, preds, k, N, consumed, True, False, True)
        cases[f"c{ci}_scores"] = scores.astype(np.float32)
        cases[f"c{ci}_U"], cases[f"c{ci}_I"] = U, I
        cases[f"c{ci}_users"] = np.asarray(users, dtype=np.int64)
        cases[f"c{ci}_k"] = np.asarray(k)
        cu = np.concatenate([np.asarray([u, len(v)] + list(v), dtype=np.int64) for u, v in consumed.items()])
        cases[f"c{ci}_consumed_flat"] = cu  # [user, n, items...] records
    np.savez_compressed(OUT / "rank_recommendations.npz", **cases)


def gen_negatives(ref):
    """sampling/negatives.py:17-93 — bit-exact target for the host sampler."""
    from libreco.sampling.negatives import (
        neg_probs_from_frequency,
        negatives_from_popular,
        negatives_from_random,
        negatives_from_unconsumed,
    )

    out = {}
    n_items = 60
    rng = np.random.default_rng(7)
    users = rng.integers(0, 20, 40)
    items_pos = rng.integers(0, n_items, 40)
    out["users"], out["items_pos"], out["n_items"] = users, items_pos, np.asarray(n_items)
    for num_neg in (1, 3):
        g = np.random.default_rng(462)  # collators seed: 42 % 3407 * 11 (batch/collators.py:180-187)
        out[f"random_{num_neg}"] = negatives_from_random(g, n_items, items_pos, num_neg)
        g = np.random.default_rng(462)
        out[f"random_items_{num_neg}"] = negatives_from_random(g, n_items, items_pos, num_neg, items=users % n_items)
    item_consumed = {i: rng.integers(0, 20, int(rng.integers(1, 9))).tolist() for i in range(n_items)}
    probs = neg_probs_from_frequency(item_consumed, n_items, 0.75)
    out["popular_probs"] = probs
    out["item_consumed_flat"] = np.concatenate(
        [np.asarray([i, len(v)] + v, dtype=np.int64) for i, v in item_consumed.items()])
    g = np.random.default_rng(462)
    out["popular_2"] = negatives_from_popular(g, n_items, items_pos, 2, probs=probs)
    user_consumed = {u: rng.integers(0, n_items, int(rng.integers(1, 30))).tolist() for u in range(20)}
    out["user_consumed_flat"] = np.concatenate(
        [np.asarray([u, len(v)] + v, dtype=np.int64) for u, v in user_consumed.items()])
    ucs = {u: set(v) for u, v in user_consumed.items()}
    for num_neg in (1, 2):
        random.seed(462)
        out[f"unconsumed_{num_neg}"] = negatives_from_unconsumed(ucs, users, items_pos, n_items, num_neg)
    np.savez_compressed(OUT / "negatives.npz", **out)


def gen_sequences(ref):
    """batch/sequence.py:33-91."""
    from libreco.batch.sequence import get_interacted_seqs, get_recent_seqs

    rng = np.random.default_rng(3)
    n_users, n_items, L = 12, 40, 5
    user_consumed = {u: rng.permutation(n_items)[: int(rng.integers(1, 14))].tolist() for u in range(n_users)}
    flat = np.concatenate([np.asarray([u, len(v)] + v, dtype=np.int64) for u, v in user_consumed.items()])
    users = rng.integers(0, n_users, 30)
    items = np.asarray([user_consumed[u][int(rng.integers(0, len(user_consumed[u])))] for u in users])
    seqs, lens = get_interacted_seqs(users, items, user_consumed, n_items, "recent", L,
                                     {u: set(v) for u, v in user_consumed.items()}, None)
    rs, rl = get_recent_seqs(n_users, user_consumed, n_items, L)
    np.savez_compressed(OUT / "sequences.npz", user_consumed_flat=flat, users=users, items=items,
                        seqs=seqs, lens=lens, recent_seqs=rs, recent_lens=rl,
                        n_users=np.asarray(n_users), n_items=np.asarray(n_items), L=np.asarray(L))


def gen_dual_sequences(ref):
    """batch/sequence.py:95-193 (SIM's long / short windows)."""
    from libreco.batch.sequence import get_dual_seqs, get_recent_dual_seqs

    rng = np.random.default_rng(9)
    n_users, n_items, Lg, S = 14, 60, 7, 3
    user_consumed = {u: rng.permutation(n_items)[: int(rng.integers(1, 25))].tolist() for u in range(n_users)}
    user_consumed[0] = user_consumed[0][:1]                       # a one-item history
    flat = np.concatenate([np.asarray([u, len(v)] + v, dtype=np.int64) for u, v in user_consumed.items()])
    users = rng.integers(0, n_users, 60)
    items = np.asarray([user_consumed[u][int(rng.integers(0, len(user_consumed[u])))] for u in users])
    ls, ll, ss, sl = get_dual_seqs(users, items, user_consumed, n_items, Lg, S, {u: set(v) for u, v in user_consumed.items()})
    rls, rll, rss, rsl = get_recent_dual_seqs(n_users, user_consumed, n_items, Lg, S)
    # the ragged histories YouTubeRetrieval's SparseCollator feeds (sequence.py:6-30), "recent" mode, window 5
    from libreco.batch.sequence import get_sparse_interacted
    sp_idx, sp_val, sp_n = get_sparse_interacted(users, items, user_consumed, "recent", 5, None)
    np.savez_compressed(OUT / "dual_sequences.npz", user_consumed_flat=flat, users=users, items=items, long_seqs=ls,
                        long_lens=ll, short_seqs=ss, short_lens=sl, recent_long=rls, recent_long_lens=rll,
                        recent_short=rss, recent_short_lens=rsl, sparse_rows=sp_idx[:, 0], sparse_values=sp_val,
                        sparse_batch=np.asarray(sp_n), n_users=np.asarray(n_users),
                        n_items=np.asarray(n_items), Lg=np.asarray(Lg), S=np.asarray(S))


def gen_lightgcn(ref):
    """algorithms/torch_modules/lightgcn_module.py:7-96 + torchops/loss.py + torch Adam step."""
    import torch
    from libreco.algorithms.torch_modules.lightgcn_module import LightGCNModel
    from libreco.torchops.loss import bpr_loss, compute_pair_scores

    rng = np.random.default_rng(11)
    n_users, n_items, K, n_layers = 30, 45, 16, 3
    user_consumed = {u: rng.integers(0, n_items, int(rng.integers(1, 10))).tolist() for u in range(n_users)}
    torch.manual_seed(42)
    m = LightGCNModel(n_users, n_items, K, n_layers, 0.0, user_consumed, torch.device("cpu"))
    lap = m.laplacian_matrix.coalesce()
    U0 = m.user_init_embeds.weight.detach().numpy().copy()
    I0 = m.item_init_embeds.weight.detach().numpy().copy()
    ue, ie = m(use_dropout=False)
    users = rng.integers(0, n_users, 20)
    pos = np.asarray([user_consumed[u][0] for u in users])
    neg = rng.integers(0, n_items, 20)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-8, weight_decay=0.0)
    ue2, ie2 = m(use_dropout=True)
    ps, ns = compute_pair_scores(ue2[torch.from_numpy(users)], ie2[torch.from_numpy(pos)], ie2[torch.from_numpy(neg)])
    loss = bpr_loss(ps, ns)
    opt.zero_grad()
    loss.backward()
    gU = m.user_init_embeds.weight.grad.numpy().copy()
    gI = m.item_init_embeds.weight.grad.numpy().copy()
    opt.step()
    np.savez_compressed(
        OUT / "lightgcn.npz",
        user_consumed_flat=np.concatenate([np.asarray([u, len(v)] + v, dtype=np.int64) for u, v in user_consumed.items()]),
        n_users=np.asarray(n_users), n_items=np.asarray(n_items), n_layers=np.asarray(n_layers),
        lap_rows=lap.indices()[0].numpy(), lap_cols=lap.indices()[1].numpy(), lap_vals=lap.values().numpy(),
        U0=U0, I0=I0, user_embeds=ue.detach().numpy(), item_embeds=ie.detach().numpy(),
        users=users, pos=pos, neg=neg, loss=np.asarray(loss.item(), dtype=np.float32), gU=gU, gI=gI,
        U1=m.user_init_embeds.weight.detach().numpy(), I1=m.item_init_embeds.weight.detach().numpy(),
    )


def gen_predict(ref):
    """prediction/predict.py:18-40."""
    from libreco.prediction.predict import normalize_prediction

    rng = np.random.default_rng(5)
    U = rng.standard_normal((20, 16)).astype(np.float32)
    I = rng.standard_normal((30, 16)).astype(np.float32)
    u = rng.integers(0, 20, 64)
    i = rng.integers(0, 30, 64)
    preds = np.sum(U[u] * I[i], axis=1)  # predict.py:39
    model = _obj(task="ranking")
    out = normalize_prediction(preds.copy(), model, "average", 0, [])
    np.savez_compressed(OUT / "predict.npz", U=U, I=I, user=u, item=i, logits=preds, probs=out)


def synthetic_frame():
    """tests/conftest.py:64-84 of the reference (`make_synthetic_data`), regenerated here from the
    same seed so the frame itself need not be stored."""
    import pandas as pd

    size = 200
    np_rng = np.random.default_rng(42)
    genres = ["crime", "drama", "action", "comedy", "missing"]
    return pd.DataFrame({
        "user": np_rng.integers(0, 20, size), "item": np_rng.integers(0, 60, size),
        "label": np_rng.integers(0, 5, size) + 1, "time": np_rng.integers(10000, 20000, size),
        "sex": np_rng.choice(["male", "female"], size), "occupation": np_rng.choice(list("abcdefg"), size),
        "age": np_rng.integers(0, 100, size), "genre1": np_rng.choice(genres, size),
        "genre2": np_rng.choice(genres, size), "genre3": np_rng.choice(genres, size),
        "profit": np_rng.random(size) * 10000,
    })


FEAT_KW = dict(sparse_col=["sex", "occupation", "genre1", "genre2", "genre3"], dense_col=["age", "profit"],
               user_col=["sex", "age", "occupation"], item_col=["genre1", "genre2", "genre3", "profit"])
MULTI_KW = dict(sparse_col=["sex", "occupation"], multi_sparse_col=[["genre1", "genre2", "genre3"]],
                dense_col=["age", "profit"], user_col=["sex", "age", "occupation"],
                item_col=["genre1", "genre2", "genre3", "profit"], pad_val=["missing"])


def _flat(d):
    return np.concatenate([np.asarray([k, len(v)] + list(v), dtype=np.int64) for k, v in d.items()])


def gen_data_layer(ref):
    """data/dataset.py, data/data_info.py, data/split.py, feature/* on the reference's own
    synthetic fixture."""
    from libreco.data import DatasetFeat, DatasetPure, split_by_ratio_chrono

    df = synthetic_frame()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    out = {"train_index": train.index.to_numpy(), "eval_index": evald.index.to_numpy()}
    ts, info = DatasetPure.build_trainset(train)
    ev = DatasetPure.build_evalset(evald)
    out.update(pure_user=ts.user_indices, pure_item=ts.item_indices, pure_label=ts.labels,
               pure_eval_user=ev.user_indices, pure_eval_item=ev.item_indices,
               pure_user_consumed=_flat(info.user_consumed), pure_item_consumed=_flat(info.item_consumed),
               pure_popular=np.asarray(info.popular_items), pure_csr_indptr=ts.sparse_interaction.indptr,
               pure_csr_indices=ts.sparse_interaction.indices, pure_csr_data=ts.sparse_interaction.data)
    ev.build_negatives(info.n_items, 2, seed=42)
    out.update(pure_evalneg_user=ev.user_indices, pure_evalneg_item=ev.item_indices, pure_evalneg_label=ev.labels)
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        ts, info = DatasetFeat.build_trainset(train_data=train, **kw)
        ev = DatasetFeat.build_testset(evald)
        out.update({f"{tag}_sparse": ts.sparse_indices, f"{tag}_dense": ts.dense_values,
                    f"{tag}_offset": info.sparse_offset, f"{tag}_oov": info.sparse_oov,
                    f"{tag}_user_sparse_unique": info.user_sparse_unique,
                    f"{tag}_item_sparse_unique": info.item_sparse_unique,
                    f"{tag}_user_dense_unique": info.user_dense_unique,
                    f"{tag}_item_dense_unique": info.item_dense_unique,
                    f"{tag}_user_sparse_cols": np.asarray(info.user_sparse_col.index),
                    f"{tag}_item_sparse_cols": np.asarray(info.item_sparse_col.index),
                    f"{tag}_user_dense_cols": np.asarray(info.user_dense_col.index),
                    f"{tag}_item_dense_cols": np.asarray(info.item_dense_col.index),
                    f"{tag}_eval_user": ev.user_indices, f"{tag}_eval_item": ev.item_indices})
        if info.multi_sparse_combine_info is not None:
            m = info.multi_sparse_combine_info
            out.update(multi_field_offset=np.asarray(m.field_offset), multi_field_len=np.asarray(m.field_len),
                       multi_feat_oov=np.asarray(m.feat_oov))
    np.savez_compressed(OUT / "data_layer.npz", **out)


def _dump_batch(out, tag, b):
    for f in ("users", "items", "labels", "queries"):
        if hasattr(b, f) and getattr(b, f) is not None:
            out[f"{tag}_{f}"] = np.asarray(getattr(b, f))
    if hasattr(b, "item_pairs"):
        out[f"{tag}_pos"], out[f"{tag}_neg"] = np.asarray(b.item_pairs[0]), np.asarray(b.item_pairs[1])
    for f in ("sparse_indices", "dense_values"):
        v = getattr(b, f, None)
        if v is None:
            continue
        if hasattr(v, "user_feats"):
            for g in ("user_feats", "item_feats"):
                if getattr(v, g) is not None:
                    out[f"{tag}_{f}_{g}"] = np.asarray(getattr(v, g))
        elif hasattr(v, "query_feats"):
            for g in ("query_feats", "item_pos_feats", "item_neg_feats"):
                if getattr(v, g) is not None:
                    out[f"{tag}_{f}_{g}"] = np.asarray(getattr(v, g))
        else:
            out[f"{tag}_{f}"] = np.asarray(v)
    if getattr(b, "seqs", None) is not None:
        out[f"{tag}_seq"], out[f"{tag}_seqlen"] = np.asarray(b.seqs.interacted_seq), np.asarray(b.seqs.interacted_len)


PARTIAL_FEATURE_CONFIGS = {          # the parametrisation of the reference's tests/test_collators.py:62-87
    "none": {"sparse_col": [], "item_col": None},
    "user_only": {"sparse_col": ["sex"], "dense_col": ["age"], "user_col": ["sex", "age"]},
    "item_only": {"sparse_col": ["genre1"], "dense_col": ["profit"], "item_col": ["genre1", "profit"]},
    "user_sparse_item_dense": {"sparse_col": ["sex"], "dense_col": ["profit"], "user_col": ["sex"], "item_col": ["profit"]},
    "both": {"sparse_col": ["sex", "genre1"], "dense_col": ["age", "profit"], "user_col": ["sex", "age"],
             "item_col": ["genre1", "profit"]},
}


def partial_collator_cases(info, stub):
    return [("deepfm", stub("DeepFM", info, num_neg=1)),
            ("din", stub("DIN", info, num_neg=2, sampler="unconsumed", seq_mode="random", max_seq_len=3)),
            ("twotower_ce", stub("TwoTower", info, num_neg=1)),
            ("twotower_softmax", stub("TwoTower", info, loss_type="softmax")),
            ("twotower_bpr", stub("TwoTower", info, loss_type="bpr", num_neg=1, sampler="popular"))]


def gen_collators_partial(ref):
    """batch/collators.py on data sets with features on one side only / no features at all."""
    import types

    from libreco.batch import get_batch_loader
    from libreco.data import DatasetFeat, split_by_ratio_chrono

    train, _ = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    out = {}

    def stub(name, info, **kw):
        m = types.SimpleNamespace(model_name=name, data_info=info, seed=42, task="ranking", sampler="random",
                                  num_neg=1, loss_type="cross_entropy")
        m.__dict__.update(kw)
        return m

    for cfg, kw in PARTIAL_FEATURE_CONFIGS.items():
        ts, info = DatasetFeat.build_trainset(train_data=train.copy(), **kw)
        for tag, m in partial_collator_cases(info, stub):
            loader = get_batch_loader(m, ts, True, batch_size=24, shuffle=True, num_workers=0, seed=42)
            _dump_batch(out, f"{cfg}_{tag}", next(iter(loader)))
    np.savez_compressed(OUT / "collators_partial.npz", **out)


def gen_collators(ref):
    """batch/collators.py + batch/batch_data.py: first batches of a seeded loader."""
    import types

    from libreco.batch import get_batch_loader
    from libreco.data import DatasetFeat, DatasetPure, split_by_ratio_chrono

    df = synthetic_frame()
    train, _ = split_by_ratio_chrono(df, test_size=0.2)
    out = {}

    def model_stub(name, info, **kw):
        m = types.SimpleNamespace(model_name=name, data_info=info, seed=42, task="ranking",
                                  sampler="random", num_neg=1, loss_type="cross_entropy")
        m.__dict__.update(kw)
        return m

    def dump(tag, b):
        _dump_batch(out, tag, b)

    ts, info = DatasetFeat.build_trainset(train_data=train, **FEAT_KW)
    cases = [
        ("deepfm_random", model_stub("DeepFM", info, num_neg=2)),
        ("deepfm_unconsumed", model_stub("DeepFM", info, sampler="unconsumed", num_neg=1)),
        ("deepfm_popular", model_stub("DeepFM", info, sampler="popular", num_neg=3)),
        ("din_random", model_stub("DIN", info, num_neg=1, seq_mode="recent", max_seq_len=4)),
        ("twotower_ce", model_stub("TwoTower", info, num_neg=2)),
        ("twotower_softmax", model_stub("TwoTower", info, loss_type="softmax")),
        ("twotower_maxmargin", model_stub("TwoTower", info, loss_type="max_margin", num_neg=2)),
    ]
    for tag, m in cases:
        loader = get_batch_loader(m, ts, True, batch_size=32, shuffle=True, num_workers=0, seed=42)
        for bi, b in enumerate(loader):
            dump(f"{tag}_b{bi}", b)
            if bi == 1:
                break
    tsp, infop = DatasetPure.build_trainset(train)
    for tag, m in [("lightgcn_bpr", model_stub("LightGCN", infop, loss_type="bpr", num_neg=2)),
                   ("lightgcn_ce", model_stub("LightGCN", infop, num_neg=1))]:
        loader = get_batch_loader(m, tsp, True, batch_size=32, shuffle=True, num_workers=0, seed=42)
        for bi, b in enumerate(loader):
            dump(f"{tag}_b{bi}", b)
            if bi == 1:
                break
    np.savez_compressed(OUT / "collators.npz", **out)


def retrain_frames():
    """Old / new interaction frames for the retrain fixtures: the new one brings new users, new
    items, new categories of a plain sparse column and of the multi-sparse field."""
    df = synthetic_frame()
    old = df.iloc[:120].reset_index(drop=True)
    new = df.iloc[120:].reset_index(drop=True).copy()
    new.loc[new.index[:12], "user"] = np.arange(100, 112)          # unseen users
    new.loc[new.index[10:30], "item"] = np.arange(500, 520)        # unseen items
    new.loc[new.index[5:9], "occupation"] = ["x", "y", "x", "z"]    # unseen categories
    new.loc[new.index[40:44], "genre2"] = ["horror", "western", "horror", "noir"]
    return old, new


def gen_retrain(ref):
    """data/dataset.py:148-196,262-345,548-700 (`merge_trainset/evalset/testset`),
    feature/update.py, data/consumed.py:42-68, data/data_info.py:543-578 (`store_old_info`)."""
    from libreco.data import DatasetFeat, DatasetPure

    old, new = retrain_frames()
    out = {}

    def dump_info(tag, ts, info, ev):
        o = info.old_info
        out.update({f"{tag}_user": ts.user_indices, f"{tag}_item": ts.item_indices, f"{tag}_label": ts.labels,
                    f"{tag}_user_unique": np.asarray(info.user_unique_vals), f"{tag}_item_unique": np.asarray(info.item_unique_vals),
                    f"{tag}_user_consumed": _flat(info.user_consumed), f"{tag}_item_consumed": _flat(info.item_consumed),
                    f"{tag}_old_n": np.asarray([o.n_users, o.n_items]),
                    f"{tag}_old_sparse_len": np.asarray(o.sparse_len, dtype=np.int64),
                    f"{tag}_old_sparse_oov": np.asarray(o.sparse_oov, dtype=np.int64),
                    f"{tag}_old_popular": np.asarray(o.popular_items),
                    f"{tag}_eval_user": ev.user_indices, f"{tag}_eval_item": ev.item_indices})

    for merge in (True, False):
        _, info0 = DatasetPure.build_trainset(old)
        ts, info = DatasetPure.merge_trainset(new, info0, merge_behavior=merge)
        ev = DatasetPure.merge_evalset(old.iloc[:40], info)
        dump_info(f"pure{int(merge)}", ts, info, ev)
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info0 = DatasetFeat.build_trainset(old, **kw)
        ts, info = DatasetFeat.merge_trainset(new, info0, merge_behavior=True)
        ev = DatasetFeat.merge_testset(old.iloc[:40], info)
        dump_info(tag, ts, info, ev)
        out.update({f"{tag}_sparse": ts.sparse_indices, f"{tag}_dense": ts.dense_values,
                    f"{tag}_offset": info.sparse_offset, f"{tag}_oov": info.sparse_oov,
                    f"{tag}_user_sparse_unique": info.user_sparse_unique,
                    f"{tag}_item_sparse_unique": info.item_sparse_unique,
                    f"{tag}_user_dense_unique": info.user_dense_unique,
                    f"{tag}_item_dense_unique": info.item_dense_unique})
        for c, v in (info.sparse_unique_vals or {}).items():
            out[f"{tag}_vocab_{c}"] = np.asarray(v)
        for c, v in (info.multi_sparse_unique_vals or {}).items():
            out[f"{tag}_mvocab_{c}"] = np.asarray(v)
        if info.multi_sparse_combine_info is not None:
            m = info.multi_sparse_combine_info
            out.update({f"{tag}_field_offset": np.asarray(m.field_offset), f"{tag}_field_len": np.asarray(m.field_len),
                        f"{tag}_feat_oov": np.asarray(m.feat_oov)})
    np.savez_compressed(OUT / "retrain.npz", **out)



def gen_metrics(ref):
    """evaluation/metrics.py: listwise + pointwise metrics on random inputs (sklearn / pandas
    based in the reference)."""
    from libreco.evaluation import metrics as M

    rng = np.random.default_rng(11)
    k, n_users, n_items = 10, 25, 60
    out = {"k": np.asarray(k), "n_items": np.asarray(n_items)}
    truths, recos = [], []
    for u in range(n_users):
        truths.append(rng.choice(n_items, size=rng.integers(1, 12), replace=False))
        recos.append(rng.choice(n_items, size=k, replace=False))
    truths[3] = np.setdiff1d(np.arange(n_items), recos[3])[:5]          # no hit at all
    out["truth_flat"] = np.concatenate([np.r_[len(t), t] for t in truths])
    out["reco"] = np.stack(recos)
    yt = {u: truths[u] for u in range(n_users)}
    yr = {u: recos[u] for u in range(n_users)}
    users = list(range(n_users))
    for name, fn in (("precision", M.precision_at_k), ("recall", M.recall_at_k),
                     ("map", M.average_precision_at_k), ("ndcg", M.ndcg_at_k)):
        out[f"{name}_per_user"] = np.asarray([fn(yt[u], yr[u], k) for u in users], dtype=np.float64)
        out[name] = np.asarray(M.listwise_scores(fn, yt, yr, users, k))
    out["coverage"] = np.asarray(M.rec_coverage(yr, users, n_items))
    n = 400
    y = rng.integers(0, 2, n).astype(np.float64)
    p = np.clip(rng.random(n) * 0.6 + y * 0.25, 0, 1)
    uidx = rng.integers(0, 12, n)
    out.update(y_true=y, y_prob=p, user_indices=uidx, rmse=np.asarray(M.rmse(y * 4 + 1, p * 5)),
               balanced_accuracy=np.asarray(M.balanced_accuracy(y, p)), roc_gauc=np.asarray(M.roc_gauc_score(y, p, uidx)),
               pr_auc=np.asarray(M.pr_auc_score(y, p)))
    np.savez_compressed(OUT / "metrics.npz", **out)



def gen_splits(ref):
    """data/split.py: index sets chosen by every split function on the synthetic frame."""
    from libreco.data import random_split, split_by_num, split_by_num_chrono, split_by_ratio, split_by_ratio_chrono

    df = synthetic_frame()
    out = {}
    cases = {
        "random": lambda: random_split(df, test_size=0.2, seed=7),
        "random_multi": lambda: random_split(df, multi_ratios=[0.7, 0.2, 0.1], seed=3, filter_unknown=False),
        "ratio": lambda: split_by_ratio(df, test_size=0.3, shuffle=True, seed=5),
        "ratio_multi_pad": lambda: split_by_ratio(df, multi_ratios=[0.6, 0.2, 0.2], filter_unknown=False,
                                                  pad_unknown=True, pad_val=[777, 888]),
        "ratio_chrono": lambda: split_by_ratio_chrono(df, test_size=0.25),
        "num": lambda: split_by_num(df, test_size=2),
        "num_unordered_shuffled": lambda: split_by_num(df, order=False, shuffle=True, test_size=4, seed=9, filter_unknown=False),
        "num_chrono": lambda: split_by_num_chrono(df, test_size=3),
    }
    for name, fn in cases.items():
        for j, part in enumerate(fn()):
            out[f"{name}_{j}_index"] = part.index.to_numpy()
            out[f"{name}_{j}_user"] = part["user"].to_numpy()
            out[f"{name}_{j}_item"] = part["item"].to_numpy()
    np.savez_compressed(OUT / "splits.npz", **out)



def gen_inference_host(ref):
    """recommendation/cold_start.py, prediction/preprocess.py:15-107 (`get_original_feats`,
    `set_temp_feats`) on a DataInfo built from the synthetic frame."""
    from libreco.data import DatasetFeat
    from libreco.prediction.preprocess import get_original_feats, set_temp_feats
    from libreco.recommendation.cold_start import cold_start_rec

    df = synthetic_frame()
    out = {}
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info = DatasetFeat.build_trainset(df, **kw)
        users, items = np.array([0, 3, 7, info.n_users]), np.array([5, 1, info.n_items, 2])
        _, _, sp, dn = get_original_feats(info, users, items, sparse=True, dense=True)
        out[f"{tag}_orig_sparse"], out[f"{tag}_orig_dense"] = sp, dn
        feats = {"sex": "male", "occupation": "c", "age": 33, "genre2": "crime", "profit": 1.5,
                 "genre1": "never-seen", "not_a_column": 1}
        sp2, dn2 = set_temp_feats(info, sp[:1], dn[:1], feats)
        out[f"{tag}_temp_sparse"], out[f"{tag}_temp_dense"] = sp2, dn2
        default_recs = np.arange(20)[::-1].copy()
        a = cold_start_rec(info, default_recs, "average", ["x", "y"], 6, inner_id=False)
        b = cold_start_rec(info, default_recs, "popular", ["x"], 5, inner_id=True)
        c = cold_start_rec(info, default_recs, "average", ["z"], 4, inner_id=True)
        out[f"{tag}_cold_average"] = np.stack([a["x"], a["y"]])
        out[f"{tag}_cold_popular_inner"] = b["x"]
        out[f"{tag}_cold_average_inner"] = c["z"]
    np.savez_compressed(OUT / "inference_host.npz", **out)



def gen_saved_data_info(ref):
    """data/data_info.py:435-487 — files written by the reference's `DataInfo.save` (tiny), for the
    load-interop test."""
    from libreco.data import DatasetFeat

    df = synthetic_frame()
    out_dir = OUT / "refsave"
    out_dir.mkdir(exist_ok=True)
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info = DatasetFeat.build_trainset(df, **kw)
        info.save(str(out_dir), tag)



def gen_ref_checkpoint(ref):
    """An INFERENCE checkpoint written by the reference's own LightGCN (bases/embed_base.py:267-300)
    after a short CPU fit, plus what that model predicts / recommends — for the load-interop test."""
    from libreco.algorithms.lightgcn import LightGCN
    from libreco.data import DatasetPure

    from libreco.data import split_by_ratio_chrono
    df, ev_df = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    df, ev_df = df[["user", "item", "label"]], ev_df[["user", "item", "label"]]
    train, info = DatasetPure.build_trainset(df)
    model = LightGCN("ranking", info, loss_type="bpr", embed_size=8, n_epochs=2, lr=1e-2, batch_size=64,
                     num_neg=1, device="cpu", seed=42)
    model.fit(train, neg_sampling=True, verbose=0)
    out_dir = OUT / "refckpt"
    out_dir.mkdir(exist_ok=True)
    model.save(str(out_dir), "lgcn", inference_only=True)
    info.save(str(out_dir), "lgcn")
    users = [int(u) for u in info.user_unique_vals[:6]]
    recs = model.recommend_user(users, n_rec=7)
    pu, pi = df["user"].to_numpy()[:30], df["item"].to_numpy()[:30]
    from libreco.evaluation import evaluate
    names = ["loss", "balanced_accuracy", "roc_auc", "pr_auc", "precision", "recall", "map", "ndcg"]
    res = evaluate(model, ev_df, neg_sampling=True, metrics=names, k=5, seed=42)
    res2 = evaluate(model, ev_df, neg_sampling=True, metrics=["roc_auc", "ndcg"], k=5, sample_user_num=7, seed=3)
    eval_vals = np.asarray([res[m] for m in names] + [res2["roc_auc"], res2["ndcg"]], dtype=np.float64)
    np.savez_compressed(out_dir / "expected.npz", eval_vals=eval_vals, users=np.asarray(users),
                        recs=np.stack([recs[u] for u in users]), pred_user=pu, pred_item=pi,
                        preds=np.asarray(model.predict(pu, pi)),
                        cold=np.asarray(model.recommend_user(-12345, n_rec=5, cold_start="popular")[-12345]))



def gen_ssl(ref):
    """feature/ssl.py: self-supervised feature masking of the two-tower model (`get_ssl_features`
    for the three patterns, `get_mutual_info`)."""
    import types

    from libreco.data import DatasetFeat
    from libreco.feature.ssl import get_mutual_info, get_ssl_features

    df = synthetic_frame()
    out = {}
    train, info = DatasetFeat.build_trainset(df, **FEAT_KW)
    mi = get_mutual_info(train, info)
    out["mutual_info"] = np.stack([mi[i] for i in range(len(mi))])
    for pattern in ("rfm", "rfm-complementary", "cfm"):
        _, info = DatasetFeat.build_trainset(df, **FEAT_KW)          # fresh np_rng per pattern
        m = types.SimpleNamespace(data_info=info, n_items=info.n_items, ssl_pattern=pattern, item_dense=True,
                                  sparse_feat_mutual_info=mi, ssl_left_sparse_indices="ls",
                                  ssl_right_sparse_indices="rs", ssl_left_dense_values="ld",
                                  ssl_right_dense_values="rd")
        for call in range(2):
            f = get_ssl_features(m, 12)
            tag = f"{pattern}_{call}"
            out[f"{tag}_left"], out[f"{tag}_right"], out[f"{tag}_dense"] = f["ls"], f["rs"], f["ld"]
    np.savez_compressed(OUT / "ssl.npz", **out)



def gen_ngcf(ref):
    """algorithms/torch_modules/ngcf_module.py:8-146 + torchops/loss.py + torch Adam: Laplacian,
    propagated embeddings, loss, gradient of every parameter and the parameters after two optimiser
    steps (plain, and weight-decay + AMSGrad); then a full `NGCF.fit` for the trained embeddings."""
    import torch
    from libreco.algorithms.torch_modules.ngcf_module import NGCFModel
    from libreco.torchops.loss import bpr_loss, compute_pair_scores

    rng = np.random.default_rng(13)
    n_users, n_items, K, layers = 30, 45, 8, [16, 12]
    user_consumed = {u: rng.integers(0, n_items, int(rng.integers(1, 10))).tolist() for u in range(n_users)}
    users = rng.integers(0, n_users, 20)
    pos = np.asarray([user_consumed[u][0] for u in users])
    neg = rng.integers(0, n_items, 20)
    out = dict(user_consumed_flat=np.concatenate([np.asarray([u, len(v)] + v, dtype=np.int64) for u, v in user_consumed.items()]),
               n_users=np.asarray(n_users), n_items=np.asarray(n_items), layers=np.asarray(layers), users=users, pos=pos, neg=neg)
    for tag, opt_kw in (("plain", dict(weight_decay=0.0, amsgrad=False)), ("wd_ams", dict(weight_decay=0.01, amsgrad=True))):
        torch.manual_seed(42)
        m = NGCFModel(n_users, n_items, K, layers, 0.0, 0.0, user_consumed, torch.device("cpu"))
        if tag == "plain":
            lap = m.laplacian_matrix.coalesce()
            out.update(lap_rows=lap.indices()[0].numpy(), lap_cols=lap.indices()[1].numpy(), lap_vals=lap.values().numpy())
            ue, ie = m(use_dropout=False)
            out.update(user_embeds=ue.detach().numpy(), item_embeds=ie.detach().numpy())
            out["init_embed"] = torch.cat([m.embedding_dict["user_embed"], m.embedding_dict["item_embed"]]).detach().numpy().copy()
            for k, p in m.weight_dict.items():
                out[f"init_{k}"] = p.detach().numpy().copy()
        opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-8, **opt_kw)
        for step in range(2):
            ue2, ie2 = m(use_dropout=True)
            ps, ns = compute_pair_scores(ue2[torch.from_numpy(users)], ie2[torch.from_numpy(pos)], ie2[torch.from_numpy(neg)])
            loss = bpr_loss(ps, ns)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out[f"{tag}_loss"] = np.asarray(loss.item(), dtype=np.float32)
                out[f"{tag}_g_embed"] = torch.cat([m.embedding_dict["user_embed"].grad, m.embedding_dict["item_embed"].grad]).numpy().copy()
                for k, p in m.weight_dict.items():
                    out[f"{tag}_g_{k}"] = p.grad.numpy().copy()
            opt.step()
        out[f"{tag}_embed2"] = torch.cat([m.embedding_dict["user_embed"], m.embedding_dict["item_embed"]]).detach().numpy().copy()
        for k, p in m.weight_dict.items():
            out[f"{tag}_{k}2"] = p.detach().numpy().copy()
    # full fit through the reference's trainer / loader / sampler
    from libreco.algorithms.ngcf import NGCF
    from libreco.data import DatasetPure, split_by_ratio_chrono
    df, _ = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    train, info = DatasetPure.build_trainset(df[["user", "item", "label"]])
    model = NGCF("ranking", info, loss_type="bpr", embed_size=8, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1,
                 hidden_units=(16, 16), device="cpu", seed=42)
    model.fit(train, neg_sampling=True, verbose=0)
    users_raw = [int(u) for u in info.user_unique_vals[:6]]
    recs = model.recommend_user(users_raw, n_rec=7)
    pu, pi = df["user"].to_numpy()[:30], df["item"].to_numpy()[:30]
    out.update(fit_user_embed=model.user_embeds_np, fit_item_embed=model.item_embeds_np, fit_users=np.asarray(users_raw),
               fit_recs=np.stack([recs[u] for u in users_raw]), fit_pred_user=pu, fit_pred_item=pi,
               fit_preds=np.asarray(model.predict(pu, pi)))
    np.savez_compressed(OUT / "ngcf.npz", **out)


def gen_knn(ref):
    """bases/embed_base.py:415-551 (exact mode): neighbours found by the reference on its own
    checkpoint tests/golden/refckpt (written by `gen_ref_checkpoint`)."""
    from libreco.algorithms.lightgcn import LightGCN
    from libreco.data import DataInfo

    d = OUT / "refckpt"
    info = DataInfo.load(str(d), "lgcn")
    model = LightGCN.load(str(d), "lgcn", info, device="cpu") if "device" in LightGCN.load.__code__.co_varnames \
        else LightGCN.load(str(d), "lgcn", info)
    users = [int(u) for u in info.user_unique_vals[:8]]
    items = [int(i) for i in info.item_unique_vals[:8]]
    out = dict(users=np.asarray(users), items=np.asarray(items))
    for sim in ("cosine", "inner-product"):
        model.init_knn(approximate=False, sim_type=sim)
        out[f"{sim}_users"] = np.asarray([model.search_knn_users(u, 5) for u in users])
        out[f"{sim}_items"] = np.asarray([model.search_knn_items(i, 5) for i in items])
    out["user_embedding_shape"] = np.asarray(model.get_user_embedding().shape)
    out["item_vec"] = model.get_item_embedding(items[0])
    np.savez_compressed(OUT / "knn.npz", **out)


def multi_value_frame():
    import pandas as pd

    rng = np.random.default_rng(5)
    pool = ["Crime", " drama", "Action ", "comedy", "Sci Fi"]
    rows = ["|".join(rng.choice(pool, rng.integers(0, 4), replace=False)) + ("|" if rng.random() < 0.3 else "")
            for _ in range(40)]
    tags = [",".join(rng.choice(list("xyz"), rng.integers(1, 3), replace=False)) for _ in range(40)]
    return pd.DataFrame({"user": rng.integers(0, 8, 40), "item": rng.integers(0, 9, 40), "genre": rows,
                         "tag": tags, "other": rng.choice(["p", None], 40)})


def gen_processing(ref):
    """data/processing.py: `process_data` for every normaliser (single frame and train/eval pair) and
    `split_multi_value` with and without `max_len`."""
    from libreco.data import process_data, split_multi_value

    out = {}
    for norm in ("min_max", "standard", "robust", "power"):
        one = synthetic_frame()
        _, cols = process_data(one, dense_col=["age", "profit"], normalizer=norm)
        out[f"{norm}_one_cols"] = np.array(cols)
        for c in cols:
            out[f"{norm}_one_{c}"] = one[c].to_numpy(np.float64)
        tr, ev = synthetic_frame().iloc[:150].copy(), synthetic_frame().iloc[150:].copy()
        _, cols = process_data((tr, ev), dense_col=["age", "label"], normalizer=norm, transformer=("log", "square"))
        out[f"{norm}_pair_cols"] = np.array(cols)
        for tag, fr in (("tr", tr), ("ev", ev)):
            out[f"{norm}_pair_{tag}_columns"] = np.array(list(fr.columns))
            for c in fr.columns:
                if c.startswith(("age", "label")):
                    out[f"{norm}_pair_{tag}_{c}"] = fr[c].to_numpy(np.float64)
    for tag, kw in (("auto", dict(max_len=None, pad_val="missing")), ("capped", dict(max_len=[2, 3], pad_val=["nil", "none"]))):
        frame = multi_value_frame()
        frame["tag"] = frame["tag"].str.replace(",", "|")
        d, multi, ucol, icol = split_multi_value(frame, ["genre", "tag"], "|", user_col=["tag"], item_col=["genre"], **kw)
        out[f"mv_{tag}_columns"] = np.array(list(d.columns))
        out[f"mv_{tag}_multi"] = np.array(["/".join(g) for g in multi])
        out[f"mv_{tag}_user"], out[f"mv_{tag}_item"] = np.array(ucol), np.array(icol)
        for c in d.columns:
            out[f"mv_{tag}_col_{c}"] = d[c].to_numpy().astype(str)
    np.savez_compressed(OUT / "processing.npz", **out)


def serving_stub_model(info, name, seed=3, with_seq=False):
    """What the serializers read off a model: name, DataInfo, numpy embeddings (OOV row last)."""
    rng = np.random.default_rng(seed)
    kw = dict(model_name=name, data_info=info, n_users=info.n_users, n_items=info.n_items,
              user_embeds_np=rng.standard_normal((info.n_users + 1, 6)).astype(np.float32),
              item_embeds_np=rng.standard_normal((info.n_items + 1, 6)).astype(np.float32))
    if with_seq:
        kw["max_seq_len"] = 7
    return _obj(**kw)


def gen_serving(ref):
    """libserving/serialization/{common,embed,online}.py: every JSON file of `save_embed` and the
    feature/mapping files of `save_online` (its TF SavedModel part cannot run here)."""
    import json
    import tempfile
    import types

    from libreco.data import DatasetFeat, DatasetPure
    from oracle.ref_loader import REFERENCE
    for name in ("libserving", "libserving.serialization"):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [str(REFERENCE / name.replace(".", "/"))]
            sys.modules[name] = pkg
    from libserving.serialization import common, embed, online

    out = {}

    def slurp(d, tag):
        for f in sorted(Path(d).iterdir()):
            if f.suffix == ".json":
                out[f"{tag}/{f.name}"] = json.loads(f.read_text())

    _, pure = DatasetPure.build_trainset(synthetic_frame()[["user", "item", "label"]])
    with tempfile.TemporaryDirectory() as d:
        embed.save_embed(d, serving_stub_model(pure, "LightGCN"))
        slurp(d, "embed")
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info = DatasetFeat.build_trainset(synthetic_frame(), **kw)
        model = serving_stub_model(info, "DIN", with_seq=True)
        with tempfile.TemporaryDirectory() as d:
            common.save_model_name(d, model)
            common.save_id_mapping(d, info)
            common.save_user_consumed(d, info)
            common.save_features(d, info, model)
            online.save_user_sparse_mapping(d, info)
            online.save_user_dense_mapping(d, info)
            slurp(d, tag)
    (OUT / "serving.json").write_text(json.dumps(out, sort_keys=True))


API_MODELS = {"FM": "fm", "DeepFM": "deepfm", "DIN": "din", "TwoTower": "two_tower", "LightGCN": "lightgcn", "NGCF": "ngcf",
              "YouTubeRanking": "youtube_ranking", "YouTubeRetrieval": "youtube_retrieval", "Transformer": "transformer",
              "SIM": "sim"}
API_METHODS = ("__init__", "fit", "predict", "recommend_user", "save", "load", "rebuild_model", "get_user_embedding",
               "get_item_embedding", "search_knn_users", "search_knn_items", "init_knn", "dyn_user_embedding")
API_FUNCTIONS = {"data": ("split_by_ratio", "split_by_num", "split_by_ratio_chrono", "split_by_num_chrono", "random_split",
                          "process_data", "split_multi_value"),
                 "evaluation": ("evaluate",),
                 "recommendation": ("rank_recommendations", "recommend_from_embedding", "cold_start_rec",
                                    "popular_recommendations", "construct_rec", "check_dynamic_rec_feats")}
API_DATA_METHODS = {"DatasetPure": ("build_trainset", "build_evalset", "build_testset", "merge_trainset", "merge_evalset",
                                    "merge_testset"),
                    "DatasetFeat": ("build_trainset", "build_evalset", "build_testset", "merge_trainset", "merge_evalset",
                                    "merge_testset"),
                    "DataInfo": ("save", "load", "assign_user_features", "assign_item_features")}


def signature_of(obj):
    """[[name, kind, repr(default) | None], ...] — what `tests/test_api_signatures_cpu.py` compares."""
    import inspect

    return [[n, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
            for n, p in inspect.signature(obj).parameters.items()]


def gen_api_signatures(ref):
    """The seam of SURVEY 8(b) as the reference declares it: constructor / method / function signatures of the model
    classes on and next to the hot path, the data layer and the scoring functions."""
    import importlib
    import json

    out = {}
    for cls, mod in API_MODELS.items():
        C = getattr(importlib.import_module("libreco.algorithms." + mod), cls)
        for m in API_METHODS:
            if hasattr(C, m):
                out[f"algorithms.{cls}.{m}"] = signature_of(getattr(C, m))
    for mod, names in API_FUNCTIONS.items():
        M = importlib.import_module("libreco." + mod)
        for n in names:
            out[f"{mod}.{n}"] = signature_of(getattr(M, n))
    D = importlib.import_module("libreco.data")
    for cls, names in API_DATA_METHODS.items():
        for n in names:
            out[f"data.{cls}.{n}"] = signature_of(getattr(getattr(D, cls), n))
    (OUT / "api_signatures.json").write_text(json.dumps(out, indent=0, sort_keys=True))


def main():
    from oracle import ref_loader

    ref = ref_loader.load()
    OUT.mkdir(parents=True, exist_ok=True)
    for fn in (gen_rank, gen_negatives, gen_sequences, gen_dual_sequences, gen_lightgcn, gen_predict, gen_data_layer, gen_collators, gen_retrain, gen_metrics, gen_splits, gen_inference_host, gen_saved_data_info, gen_ref_checkpoint, gen_ssl, gen_processing, gen_serving, gen_ngcf, gen_knn, gen_collators_partial, gen_api_signatures):
        if len(sys.argv) > 1 and fn.__name__ not in sys.argv[1:]:
            continue
        fn(ref)
        print("wrote fixtures:", fn.__name__)


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
