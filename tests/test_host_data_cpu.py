"""Host-side index work vs the reference's own outputs (fixtures from oracle/make_golden.py):
data encoding, sparse layout, unique-feature matrices, consumed histories, splits, negative
samplers, sequence builders and collators.  Bar: bit-exact."""
import random
import types

import numpy as np
import pandas as pd
import pytest

from librecommender_amd.batch import get_batch_loader, get_interacted_seqs, get_recent_seqs
from librecommender_amd.data import DatasetFeat, DatasetPure, split_by_ratio_chrono
from librecommender_amd.data.consumed import interaction_consumed
from librecommender_amd.sampling import (
    neg_probs_from_frequency,
    negatives_from_popular,
    negatives_from_random,
    negatives_from_unconsumed,
)
from oracle.make_golden import FEAT_KW, MULTI_KW, synthetic_frame
from tests.golden_util import unflatten


@pytest.fixture(scope="module")
def frames():
    df = synthetic_frame()
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    return df, train, evald


def test_consumed_reference_kat():
    uc, ic = interaction_consumed([1, 1, 1, 2, 2, 1, 2, 3, 2, 3], [11, 11, 999, 0, 11, 11, 999, 11, 999, 0])
    assert uc == {1: [11, 999, 11], 2: [0, 11, 999], 3: [11, 0]}      # tests/test_consumed.py:12-25
    assert ic == {11: [1, 2, 1, 3], 999: [1, 2], 0: [2, 3]}


def test_split_and_pure_dataset(frames, golden_dir):
    g = np.load(golden_dir / "data_layer.npz")
    _, train, evald = frames
    np.testing.assert_array_equal(train.index.to_numpy(), g["train_index"])
    np.testing.assert_array_equal(evald.index.to_numpy(), g["eval_index"])
    ts, info = DatasetPure.build_trainset(train)
    np.testing.assert_array_equal(ts.user_indices, g["pure_user"])
    np.testing.assert_array_equal(ts.item_indices, g["pure_item"])
    np.testing.assert_array_equal(ts.labels, g["pure_label"])
    assert info.user_consumed == unflatten(g["pure_user_consumed"])
    assert info.item_consumed == unflatten(g["pure_item_consumed"])
    assert info.popular_items == g["pure_popular"].tolist()
    csr = ts.sparse_interaction
    np.testing.assert_array_equal(csr.indptr, g["pure_csr_indptr"])
    np.testing.assert_array_equal(csr.indices, g["pure_csr_indices"])
    np.testing.assert_array_equal(csr.data, g["pure_csr_data"])
    ev = DatasetPure.build_evalset(evald)
    np.testing.assert_array_equal(ev.user_indices, g["pure_eval_user"])
    np.testing.assert_array_equal(ev.item_indices, g["pure_eval_item"])
    ev.build_negatives(info.n_items, 2, seed=42)
    np.testing.assert_array_equal(ev.user_indices, g["pure_evalneg_user"])
    np.testing.assert_array_equal(ev.item_indices, g["pure_evalneg_item"])
    np.testing.assert_array_equal(ev.labels, g["pure_evalneg_label"])


@pytest.mark.parametrize("tag,kw", [("feat", FEAT_KW), ("multi", MULTI_KW)])
def test_feat_dataset(frames, golden_dir, tag, kw):
    g = np.load(golden_dir / "data_layer.npz")
    _, train, evald = frames
    ts, info = DatasetFeat.build_trainset(train_data=train, **kw)
    np.testing.assert_array_equal(ts.sparse_indices, g[f"{tag}_sparse"])
    np.testing.assert_array_equal(ts.dense_values, g[f"{tag}_dense"])
    np.testing.assert_array_equal(info.sparse_offset, g[f"{tag}_offset"])
    np.testing.assert_array_equal(info.sparse_oov, g[f"{tag}_oov"])
    for name in ("user_sparse_unique", "item_sparse_unique", "user_dense_unique", "item_dense_unique"):
        np.testing.assert_array_equal(getattr(info, name), g[f"{tag}_{name}"], err_msg=name)
    np.testing.assert_array_equal(info.user_sparse_col.index, g[f"{tag}_user_sparse_cols"])
    np.testing.assert_array_equal(info.item_sparse_col.index, g[f"{tag}_item_sparse_cols"])
    np.testing.assert_array_equal(info.user_dense_col.index, g[f"{tag}_user_dense_cols"])
    np.testing.assert_array_equal(info.item_dense_col.index, g[f"{tag}_item_dense_cols"])
    ev = DatasetFeat.build_testset(evald)
    np.testing.assert_array_equal(ev.user_indices, g[f"{tag}_eval_user"])
    np.testing.assert_array_equal(ev.item_indices, g[f"{tag}_eval_item"])
    if tag == "multi":
        m = info.multi_sparse_combine_info
        np.testing.assert_array_equal(m.field_offset, g["multi_field_offset"])
        np.testing.assert_array_equal(m.field_len, g["multi_field_len"])
        np.testing.assert_array_equal(m.feat_oov, g["multi_feat_oov"])


def test_sparse_layout_reference_kat():
    """tests/test_feature.py:148-255 of the reference pins offsets [0,3,7,12,18,23,27] and OOV rows
    [2,6,11,17,22,26,31] for vocab sizes 2,3,4,5,4,3,4."""
    from librecommender_amd.data.vocab import SparseSchema
    sizes = [2, 3, 4, 5, 4, 3, 4]
    s = SparseSchema(sparse_cols=[f"c{i}" for i in range(7)])
    for i, n in enumerate(sizes):
        s.vocab[f"c{i}"] = np.arange(n)
    np.testing.assert_array_equal(s.offsets, [0, 3, 7, 12, 18, 23, 27])
    np.testing.assert_array_equal(s.oov_rows, [2, 6, 11, 17, 22, 26, 31])


def test_dataset_errors(frames):
    _, train, evald = frames
    DatasetPure.train_called = False
    with pytest.raises(RuntimeError):
        DatasetPure.build_evalset(evald)
    with pytest.raises(ValueError):
        DatasetPure.build_trainset(train[["item", "user", "label"]])
    with pytest.raises(ValueError):
        DatasetFeat.build_trainset(train, user_col=["sex"], item_col=[], sparse_col=["sex", "occupation"])
    DatasetPure.build_trainset(train)


def test_negative_samplers_bit_exact(golden_dir):
    g = np.load(golden_dir / "negatives.npz")
    users, pos, n_items = g["users"], g["items_pos"], int(g["n_items"])
    for k in (1, 3):
        np.testing.assert_array_equal(negatives_from_random(np.random.default_rng(462), n_items, pos, k), g[f"random_{k}"])
        np.testing.assert_array_equal(
            negatives_from_random(np.random.default_rng(462), n_items, pos, k, items=users % n_items), g[f"random_items_{k}"])
    ic = unflatten(g["item_consumed_flat"])
    probs = neg_probs_from_frequency(ic, n_items, 0.75)
    np.testing.assert_array_equal(probs, g["popular_probs"])
    np.testing.assert_array_equal(negatives_from_popular(np.random.default_rng(462), n_items, pos, 2, probs=probs), g["popular_2"])
    ucs = {u: set(v) for u, v in unflatten(g["user_consumed_flat"]).items()}
    for k in (1, 2):
        random.seed(462)
        np.testing.assert_array_equal(negatives_from_unconsumed(ucs, users, pos, n_items, k), g[f"unconsumed_{k}"])


def test_sequence_builders_bit_exact(golden_dir):
    g = np.load(golden_dir / "sequences.npz")
    uc = unflatten(g["user_consumed_flat"])
    n_users, n_items, L = int(g["n_users"]), int(g["n_items"]), int(g["L"])
    seqs, lens = get_interacted_seqs(g["users"], g["items"], uc, n_items, "recent", L)
    np.testing.assert_array_equal(seqs, g["seqs"])
    np.testing.assert_array_equal(lens, g["lens"])
    rs, rl = get_recent_seqs(n_users, uc, n_items, L)
    np.testing.assert_array_equal(rs, g["recent_seqs"])
    np.testing.assert_array_equal(rl, g["recent_lens"])


def test_dual_sequence_builders_bit_exact(golden_dir):
    """SIM's long / short windows (batch/sequence.py:95-193) against the reference's own loops."""
    from librecommender_amd.batch.sequence import get_dual_seqs, get_recent_dual_seqs

    g = np.load(golden_dir / "dual_sequences.npz")
    uc = unflatten(g["user_consumed_flat"])
    n_users, n_items, Lg, S = int(g["n_users"]), int(g["n_items"]), int(g["Lg"]), int(g["S"])
    got = get_dual_seqs(g["users"], g["items"], uc, n_items, Lg, S)
    for a, k in zip(got, ("long_seqs", "long_lens", "short_seqs", "short_lens")):
        np.testing.assert_array_equal(a, g[k], err_msg=k)
        assert a.dtype == g[k].dtype
    got = get_recent_dual_seqs(n_users, uc, n_items, Lg, S)
    for a, k in zip(got, ("recent_long", "recent_long_lens", "recent_short", "recent_short_lens")):
        np.testing.assert_array_equal(a, g[k], err_msg=k)
        assert a.dtype == g[k].dtype


def test_padded_windows_equal_reference_sparse_histories(golden_dir):
    """YouTubeRetrieval's batches: the reference feeds ragged (row, item) lists (`get_sparse_interacted`,
    batch/sequence.py:6-30); here the same histories travel as padded windows — non-pad entries per row must be the
    reference's values of that row, in order, and rows without history are all pad."""
    from librecommender_amd.batch.sequence import SequenceBuilder

    g = np.load(golden_dir / "dual_sequences.npz")
    uc = unflatten(g["user_consumed_flat"])
    n_items = int(g["n_items"])
    seqs, _ = SequenceBuilder(uc, n_items, 5, "recent").training_seqs(g["users"], g["items"])
    assert int(g["sparse_batch"]) == len(g["users"])
    rows, vals = g["sparse_rows"], g["sparse_values"]
    for j in range(len(g["users"])):
        got = seqs[j][seqs[j] != n_items]
        np.testing.assert_array_equal(got, vals[rows == j], err_msg=f"row {j}")


def _stub(name, info, **kw):
    m = types.SimpleNamespace(model_name=name, data_info=info, seed=42, task="ranking", sampler="random",
                              num_neg=1, loss_type="cross_entropy", uses_features=name not in ("LightGCN",),
                              uses_sequence=name == "DIN", graph_backend="torch" if name == "LightGCN" else "tf")
    m.__dict__.update(kw)
    return m


def _compare_batch(tag, b, g):
    got = {}
    for f in ("users", "items", "labels", "queries"):
        if getattr(b, f, None) is not None:
            got[f"{tag}_{f}"] = np.asarray(getattr(b, f))
    if hasattr(b, "item_pairs"):
        got[f"{tag}_pos"], got[f"{tag}_neg"] = b.item_pairs
    for f in ("sparse_indices", "dense_values"):
        v = getattr(b, f, None)
        if v is None:
            continue
        if hasattr(v, "user_feats"):
            for h in ("user_feats", "item_feats"):
                if getattr(v, h) is not None:
                    got[f"{tag}_{f}_{h}"] = getattr(v, h)
        elif hasattr(v, "query_feats"):
            for h in ("query_feats", "item_pos_feats", "item_neg_feats"):
                if getattr(v, h) is not None:
                    got[f"{tag}_{f}_{h}"] = getattr(v, h)
        else:
            got[f"{tag}_{f}"] = v
    if getattr(b, "seqs", None) is not None:
        got[f"{tag}_seq"], got[f"{tag}_seqlen"] = b.seqs.interacted_seq, b.seqs.interacted_len
    want = {k: g[k] for k in g.files if k.startswith(tag + "_")}
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        np.testing.assert_array_equal(np.asarray(got[k]), want[k], err_msg=k)


def test_collators_and_loader_bit_exact(frames, golden_dir):
    g = np.load(golden_dir / "collators.npz")
    _, train, _ = frames
    ts, info = DatasetFeat.build_trainset(train_data=train, **FEAT_KW)
    cases = [
        ("deepfm_random", _stub("DeepFM", info, num_neg=2)),
        ("deepfm_unconsumed", _stub("DeepFM", info, sampler="unconsumed", num_neg=1)),
        ("deepfm_popular", _stub("DeepFM", info, sampler="popular", num_neg=3)),
        ("din_random", _stub("DIN", info, num_neg=1, seq_mode="recent", max_seq_len=4)),
        ("twotower_ce", _stub("TwoTower", info, num_neg=2)),
        ("twotower_softmax", _stub("TwoTower", info, loss_type="softmax")),
        ("twotower_maxmargin", _stub("TwoTower", info, loss_type="max_margin", num_neg=2)),
    ]
    for tag, m in cases:
        loader = get_batch_loader(m, ts, True, batch_size=32, shuffle=True, num_workers=0, seed=42)
        for bi, b in enumerate(loader):
            _compare_batch(f"{tag}_b{bi}", b, g)
            if bi == 1:
                break
    tsp, infop = DatasetPure.build_trainset(train)
    for tag, m in [("lightgcn_bpr", _stub("LightGCN", infop, loss_type="bpr", num_neg=2)),
                   ("lightgcn_ce", _stub("LightGCN", infop, num_neg=1))]:
        loader = get_batch_loader(m, tsp, True, batch_size=32, shuffle=True, num_workers=0, seed=42)
        for bi, b in enumerate(loader):
            _compare_batch(f"{tag}_b{bi}", b, g)
            if bi == 1:
                break


def test_retrain_merge_matches_reference(golden_dir):
    """`merge_trainset / merge_evalset / merge_testset` (row f3) against fixtures produced by the
    reference's own data layer (oracle/make_golden.py:gen_retrain): new users, items and feature
    categories are appended, known indices stay put, offsets / OOV rows / unique feature matrices
    and histories are rebuilt exactly; `old_info` carries the reference's quirks."""
    from oracle.make_golden import FEAT_KW, MULTI_KW, retrain_frames
    from librecommender_amd.data import DatasetFeat, DatasetPure

    g = np.load(golden_dir / "retrain.npz", allow_pickle=True)
    old, new = retrain_frames()

    def flat(d):
        return np.concatenate([np.asarray([k, len(v)] + list(v), dtype=np.int64) for k, v in d.items()])

    def check(tag, ts, info, ev):
        for k, v in (("user", ts.user_indices), ("item", ts.item_indices), ("label", ts.labels),
                     ("user_unique", info.user_unique_vals), ("item_unique", info.item_unique_vals),
                     ("user_consumed", flat(info.user_consumed)), ("item_consumed", flat(info.item_consumed)),
                     ("eval_user", ev.user_indices), ("eval_item", ev.item_indices)):
            np.testing.assert_array_equal(np.asarray(v), g[f"{tag}_{k}"], err_msg=f"{tag}_{k}")
        o = info.old_info
        assert [o.n_users, o.n_items] == g[f"{tag}_old_n"].tolist()
        assert list(o.sparse_len) == g[f"{tag}_old_sparse_len"].tolist()
        assert list(o.sparse_oov) == g[f"{tag}_old_sparse_oov"].tolist()
        assert list(o.popular_items) == g[f"{tag}_old_popular"].tolist()

    for merge in (True, False):
        _, info0 = DatasetPure.build_trainset(old)
        ts, info = DatasetPure.merge_trainset(new, info0, merge_behavior=merge)
        check(f"pure{int(merge)}", ts, info, DatasetPure.merge_evalset(old.iloc[:40], info))
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info0 = DatasetFeat.build_trainset(old, **kw)
        ts, info = DatasetFeat.merge_trainset(new, info0)
        check(tag, ts, info, DatasetFeat.merge_testset(old.iloc[:40], info))
        for k, v in (("sparse", ts.sparse_indices), ("dense", ts.dense_values), ("offset", info.sparse_offset),
                     ("oov", info.sparse_oov), ("user_sparse_unique", info.user_sparse_unique),
                     ("item_sparse_unique", info.item_sparse_unique), ("user_dense_unique", info.user_dense_unique),
                     ("item_dense_unique", info.item_dense_unique)):
            np.testing.assert_array_equal(np.asarray(v), g[f"{tag}_{k}"], err_msg=f"{tag}_{k}")
        for c, v in (info.sparse_unique_vals or {}).items():
            np.testing.assert_array_equal(np.asarray(v), g[f"{tag}_vocab_{c}"])
        for c, v in (info.multi_sparse_unique_vals or {}).items():
            np.testing.assert_array_equal(np.asarray(v), g[f"{tag}_mvocab_{c}"])
        if info.multi_sparse_combine_info is not None:
            m = info.multi_sparse_combine_info
            assert list(m.field_offset) == g[f"{tag}_field_offset"].tolist()
            assert list(m.field_len) == g[f"{tag}_field_len"].tolist()
            np.testing.assert_array_equal(np.asarray(m.feat_oov), g[f"{tag}_feat_oov"])
    with pytest.raises(ValueError):
        DatasetFeat.merge_trainset(new.drop(columns=["occupation"]), info0)


def test_rebuild_growth_index_matches_reference_logic(golden_dir):
    """Table growth map of `rebuild_model` (row f3) == tfops/rebuild.py:49-74 applied variable by
    variable, on the real old_info / offsets of the retrain fixtures (incl. the multi-sparse quirk)."""
    from types import SimpleNamespace

    from librecommender_amd.data.retrain import OldInfo
    from librecommender_amd.training.rebuild import table_growth_index
    from oracle import ops_np

    g = np.load(golden_dir / "retrain.npz", allow_pickle=True)
    rng = np.random.default_rng(0)
    for tag in ("feat", "multi"):
        uo, no = g[f"{tag}_old_n"].tolist()
        old = OldInfo(uo, no, g[f"{tag}_old_sparse_len"].tolist(), g[f"{tag}_old_sparse_oov"].tolist(), [])
        s_old = int(max(old.sparse_oov)) + 1
        new_info = SimpleNamespace(n_users=len(g[f"{tag}_user_unique"]), n_items=len(g[f"{tag}_item_unique"]),
                                   sparse_offset=g[f"{tag}_offset"])
        s_new = int(g[f"{tag}_oov"].max()) + 1
        K = 4
        olds = {"user": rng.random((uo + 1, K)), "item": rng.random((no + 1, K)), "sparse": rng.random((s_old, K))}
        news = {"user": rng.random((new_info.n_users + 1, K)), "item": rng.random((new_info.n_items + 1, K)),
                "sparse": rng.random((s_new, K))}
        want = np.concatenate([ops_np.rebuild_assign(news[k], olds[k], k, uo, no, old.sparse_len, old.sparse_oov,
                                                     new_info.sparse_offset) for k in ("user", "item", "sparse")])
        old_cat = np.concatenate([olds[k] for k in ("user", "item", "sparse")])
        got = np.concatenate([news[k] for k in ("user", "item", "sparse")])
        src, dst = table_growth_index(len(old_cat), old, new_info)
        got[dst] = old_cat[src]
        np.testing.assert_array_equal(got, want)


def test_split_functions_match_reference(golden_dir):
    """data/split.py: every split function picks exactly the rows the reference picks
    (tests/golden/splits.npz, generated with the reference on the same synthetic frame)."""
    from librecommender_amd.data import (random_split, split_by_num, split_by_num_chrono, split_by_ratio,
                                         split_by_ratio_chrono)
    from oracle.make_golden import synthetic_frame

    g = np.load(golden_dir / "splits.npz")
    df = synthetic_frame()
    cases = {
        "random": lambda: random_split(df, test_size=0.2, seed=7),
        "random_multi": lambda: random_split(df, multi_ratios=[0.7, 0.2, 0.1], seed=3, filter_unknown=False),
        "ratio": lambda: split_by_ratio(df, test_size=0.3, shuffle=True, seed=5),
        "ratio_multi_pad": lambda: split_by_ratio(df, multi_ratios=[0.6, 0.2, 0.2], filter_unknown=False,
                                                  pad_unknown=True, pad_val=[777, 888]),
        "ratio_chrono": lambda: split_by_ratio_chrono(df, test_size=0.25),
        "num": lambda: split_by_num(df, test_size=2),
        "num_unordered_shuffled": lambda: split_by_num(df, order=False, shuffle=True, test_size=4, seed=9,
                                                       filter_unknown=False),
        "num_chrono": lambda: split_by_num_chrono(df, test_size=3),
    }
    for name, fn in cases.items():
        for j, part in enumerate(fn()):
            np.testing.assert_array_equal(part.index.to_numpy(), g[f"{name}_{j}_index"], err_msg=f"{name}_{j}")
            np.testing.assert_array_equal(part["user"].to_numpy(), g[f"{name}_{j}_user"])
            np.testing.assert_array_equal(part["item"].to_numpy(), g[f"{name}_{j}_item"])
    with pytest.raises(AssertionError):
        split_by_num(df, test_size=0.5)


def test_inference_host_helpers_match_reference(golden_dir):
    """Feature extraction for (user, item) pairs, temporary feature overrides and cold-start picks
    (`prediction/preprocess.py:15-107`, `recommendation/cold_start.py`) against reference outputs —
    incl. OOV ids, unknown category values, non-feature keys and the `np_rng` stream of DataInfo."""
    from librecommender_amd.bases.feat_base import merge_user_item_feats
    from librecommender_amd.data import DatasetFeat
    from librecommender_amd.feature_override import override_dense, override_sparse
    from librecommender_amd.recommendation.cold_start import cold_start_rec
    from oracle.make_golden import FEAT_KW, MULTI_KW, synthetic_frame

    g = np.load(golden_dir / "inference_host.npz", allow_pickle=True)
    df = synthetic_frame()
    for tag, kw in (("feat", FEAT_KW), ("multi", MULTI_KW)):
        _, info = DatasetFeat.build_trainset(df, **kw)
        users, items = np.array([0, 3, 7, info.n_users]), np.array([5, 1, info.n_items, 2])
        sp, dn = merge_user_item_feats(info, users, items)
        np.testing.assert_array_equal(sp, g[f"{tag}_orig_sparse"])
        np.testing.assert_array_equal(dn, g[f"{tag}_orig_dense"])
        feats = {"sex": "male", "occupation": "c", "age": 33, "genre2": "crime", "profit": 1.5,
                 "genre1": "never-seen", "not_a_column": 1}
        np.testing.assert_array_equal(override_sparse(info, sp[:1], feats), g[f"{tag}_temp_sparse"])
        np.testing.assert_array_equal(override_dense(info, dn[:1], feats), g[f"{tag}_temp_dense"])
        default_recs = np.arange(20)[::-1].copy()
        a = cold_start_rec(info, default_recs, "average", ["x", "y"], 6, inner_id=False)
        b = cold_start_rec(info, default_recs, "popular", ["x"], 5, inner_id=True)
        c = cold_start_rec(info, default_recs, "average", ["z"], 4, inner_id=True)
        np.testing.assert_array_equal(np.stack([a["x"], a["y"]]), g[f"{tag}_cold_average"])
        np.testing.assert_array_equal(b["x"], g[f"{tag}_cold_popular_inner"])
        np.testing.assert_array_equal(c["z"], g[f"{tag}_cold_average_inner"])
    with pytest.raises(ValueError):
        cold_start_rec(info, default_recs, "oops", ["x"], 3, inner_id=False)


def _same_info(a, b):
    for k in ("user_unique_vals", "item_unique_vals", "sparse_offset", "sparse_oov", "user_sparse_unique",
              "item_sparse_unique", "user_dense_unique", "item_dense_unique"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None), k
        if x is not None:
            np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg=k)
    assert a.col_name_mapping == b.col_name_mapping
    assert a.user_consumed == b.user_consumed and a.item_consumed == b.item_consumed
    assert a.n_users == b.n_users and a.n_items == b.n_items
    for src in ("sparse_unique_vals", "multi_sparse_unique_vals"):
        x, y = getattr(a, src) or {}, getattr(b, src) or {}
        assert x.keys() == y.keys()
        for c in x:
            np.testing.assert_array_equal(np.asarray(x[c]), np.asarray(y[c]))
    ma, mb = a.multi_sparse_combine_info, b.multi_sparse_combine_info
    assert (ma is None) == (mb is None)
    if ma is not None:
        assert list(ma.field_offset) == list(mb.field_offset) and list(ma.field_len) == list(mb.field_len)
        np.testing.assert_array_equal(np.asarray(ma.feat_oov), np.asarray(mb.feat_oov))
        assert dict(ma.pad_val) == dict(mb.pad_val)
    assert list(a.popular_items) == list(b.popular_items)
    assert a.np_rng.integers(0, 1 << 30) == b.np_rng.integers(0, 1 << 30)      # same seed -> same stream


@pytest.mark.parametrize("tag", ["feat", "multi"])
def test_data_info_save_load_and_reference_files(golden_dir, tmp_path, tag):
    """`DataInfo.save / load` use the reference's on-disk layout (data_info.py:435-541): files
    written by the REFERENCE (tests/golden/refsave/, oracle/make_golden.py:gen_saved_data_info) load
    into an object equal to the one built here from the same frame, and our own files round-trip."""
    from librecommender_amd.data import DataInfo, DatasetFeat
    from oracle.make_golden import FEAT_KW, MULTI_KW, synthetic_frame

    _, built = DatasetFeat.build_trainset(synthetic_frame(), **(FEAT_KW if tag == "feat" else MULTI_KW))
    _same_info(DataInfo.load(str(golden_dir / "refsave"), tag), built)
    _, built = DatasetFeat.build_trainset(synthetic_frame(), **(FEAT_KW if tag == "feat" else MULTI_KW))
    built.save(str(tmp_path), "mine")
    _, again = DatasetFeat.build_trainset(synthetic_frame(), **(FEAT_KW if tag == "feat" else MULTI_KW))
    _same_info(DataInfo.load(str(tmp_path), "mine"), again)
    with pytest.raises(OSError):
        DataInfo.load(str(tmp_path / "nope"), "x")


def test_tf_lr_decay_schedule():
    """`lr_decay=True` of the TF trainer: staircase exponential decay, rate 0.96, one decay step per
    int(data_size / batch_size) optimiser steps (training/tf_trainer.py:111-113, tfops/configs.py:38-45)."""
    from types import SimpleNamespace

    from librecommender_amd.bases.base import Base

    m = SimpleNamespace(lr=0.01, lr_decay=True, batch_size=256, data_info=SimpleNamespace(data_size=1000),
                        net=SimpleNamespace(step=0, lr=0.01))
    for step, want in ((0, 0.01), (2, 0.01), (3, 0.01 * 0.96), (5, 0.01 * 0.96), (6, 0.01 * 0.96 ** 2), (300, 0.01 * 0.96 ** 100)):
        m.net.step = step
        assert abs(Base.current_lr(m) - want) < 1e-15
    m.current_lr = lambda: Base.current_lr(m)
    Base.apply_lr_schedule(m)
    assert m.net.lr == Base.current_lr(m) == 0.01 * 0.96 ** 100
    m.lr_decay = False
    assert Base.current_lr(m) == 0.01


def test_ssl_feature_generation_matches_reference(golden_dir):
    """feature/ssl.py: the masked index views of all three `ssl_pattern`s over two consecutive
    batches (np_rng stream) and the mutual-information column table."""
    from types import SimpleNamespace

    from librecommender_amd.data import DatasetFeat
    from librecommender_amd.feature_ssl import get_mutual_info, get_ssl_features
    from oracle.make_golden import FEAT_KW, synthetic_frame

    g = np.load(golden_dir / "ssl.npz")
    df = synthetic_frame()
    train, info = DatasetFeat.build_trainset(df, **FEAT_KW)
    mi = get_mutual_info(train, info)
    np.testing.assert_array_equal(np.stack([mi[i] for i in range(len(mi))]), g["mutual_info"])
    for pattern in ("rfm", "rfm-complementary", "cfm"):
        _, info = DatasetFeat.build_trainset(df, **FEAT_KW)
        m = SimpleNamespace(data_info=info, n_items=info.n_items, ssl_pattern=pattern, item_dense=True,
                            sparse_feat_mutual_info=mi)
        for call in range(2):
            left, right, dense = get_ssl_features(m, 12)
            np.testing.assert_array_equal(left, g[f"{pattern}_{call}_left"])
            np.testing.assert_array_equal(right, g[f"{pattern}_{call}_right"])
            np.testing.assert_array_equal(dense, g[f"{pattern}_{call}_dense"])


# ---- known-answer vectors held by the reference's own tests/test_feature.py ----------------------
def _kat_frame():
    """tests/test_feature.py:124-145 (`feature_data`); label / age / profit are random there and not
    part of any assertion."""
    import pandas as pd
    return pd.DataFrame({
        "user": [4, 1, 10, 11, 12], "item": [1, 2, 3, 4, 5], "label": [1, 2, 3, 4, 5],
        "sex": ["M", "F", "M", "M", "F"], "occupation": ["c", "a", "a", "b", "a"], "age": [10, 20, 30, 40, 50],
        "actor1": [11, 0, 77, 44, 77], "actor2": [0, 22, 11, 99, 77], "profit": [1.0, 2.0, 3.0, 4.0, 5.0],
        "genre1": ["x", "y", "z", "x", "missing"], "genre2": ["xx", "missing", "xx", "z", "missing"],
        "genre3": ["y", "y", "zz", "x", "missing"]})


def test_reference_kat_sparse_indices():
    """tests/test_feature.py:148-255: vocabularies, offsets, OOV rows and the merged index matrix."""
    from librecommender_amd.data import DatasetFeat
    from librecommender_amd.data.vocab import encode

    sparse_cols = ["sex", "occupation", "actor1", "actor2", "genre1", "genre2", "genre3"]
    ts, info = DatasetFeat.build_trainset(_kat_frame(), sparse_col=sparse_cols, dense_col=["age", "profit"],
                                          user_col=["sex", "age", "occupation", "actor1", "actor2"],
                                          item_col=["genre1", "genre2", "genre3", "profit"])
    u = DatasetFeat.sparse_unique_vals
    assert list(u["sex"]) == ["F", "M"] and list(u["occupation"]) == ["a", "b", "c"]
    assert list(u["genre1"]) == ["missing", "x", "y", "z"] and list(u["genre2"]) == ["missing", "xx", "z"]
    assert list(u["genre3"]) == ["missing", "x", "y", "zz"]
    assert list(u["actor1"]) == [0, 11, 44, 77] and list(u["actor2"]) == [0, 11, 22, 77, 99]
    assert DatasetFeat.multi_sparse_col is None and DatasetFeat.multi_sparse_unique_vals is None
    np.testing.assert_array_equal(info.sparse_offset, [0, 3, 7, 12, 18, 23, 27])
    np.testing.assert_array_equal(info.sparse_oov, [2, 6, 11, 17, 22, 26, 31])
    want = np.array([[1, 0, 1, 1, 0], [5, 3, 3, 4, 3], [8, 7, 10, 9, 10], [12, 14, 13, 16, 15],
                     [19, 20, 21, 19, 18], [24, 23, 24, 25, 23], [29, 29, 30, 28, 27]]).T
    np.testing.assert_array_equal(ts.sparse_indices, want)
    with pytest.raises(KeyError):                                   # :210-213
        encode(_kat_frame()["sex"].to_numpy(), np.array(["M"]), allow_unknown=False)
    np.testing.assert_array_equal(encode(_kat_frame()["sex"].to_numpy(), np.array(["M"]), allow_unknown=True), [0, 1, 0, 0, 1])


def test_reference_kat_multi_sparse_indices():
    """tests/test_feature.py:258-378: multi-sparse fields share a vocabulary, an offset and an OOV
    row; padding values map to the OOV row."""
    from librecommender_amd.data import DatasetFeat

    kw = dict(sparse_col=["sex", "occupation"], multi_sparse_col=[["actor1", "actor2"], ["genre1", "genre2", "genre3"]],
              dense_col=["age", "profit"], user_col=["sex", "age", "occupation", "actor1", "actor2"],
              item_col=["genre1", "genre2", "genre3", "profit"])
    with pytest.raises(ValueError, match="Length of `multi_sparse_col` and `pad_val` doesn't match"):
        DatasetFeat.build_trainset(_kat_frame(), pad_val=["missing", "a", "b"], **kw)
    ts, info = DatasetFeat.build_trainset(_kat_frame(), pad_val=[0, "missing"], **kw)
    m = info.multi_sparse_combine_info
    assert list(m.field_offset) == [2, 4] and list(m.field_len) == [2, 3]
    np.testing.assert_array_equal(m.feat_oov, [12, 18])
    assert dict(m.pad_val) == {"actor1": 0, "genre1": "missing"}
    mu = DatasetFeat.multi_sparse_unique_vals
    assert list(mu["actor1"]) == [11, 22, 44, 77, 99] and list(mu["genre1"]) == ["x", "xx", "y", "z", "zz"]
    assert info.col_name_mapping["multi_sparse"] == {"genre2": "genre1", "genre3": "genre1", "actor2": "actor1"}
    np.testing.assert_array_equal(info.sparse_offset, [0, 3, 7, 7, 13, 13, 13])
    np.testing.assert_array_equal(info.sparse_oov, [2, 6, 12, 12, 18, 18, 18])
    want = np.array([[1, 0, 1, 1, 0], [5, 3, 3, 4, 3], [7, 12, 10, 9, 10], [12, 8, 7, 11, 10],
                     [13, 15, 16, 13, 18], [14, 18, 14, 16, 18], [15, 15, 17, 13, 18]]).T
    np.testing.assert_array_equal(ts.sparse_indices, want)


def test_reference_kat_update_features():
    """tests/test_feature.py:380-536 (`feature_data_pair` + `test_update_features`): retrain merge on
    the reference's literal frames — appended vocabularies, re-based unique feature matrices."""
    import pandas as pd

    from librecommender_amd.data import DatasetFeat

    data = pd.DataFrame({"user": [4, 1, 10], "item": [1, 2, 8], "label": [1, 0, 1], "sex": ["M", "F", "M"],
                         "occupation": ["c", "a", "a"], "age": [1, 2, 3], "actor1": [11, 0, 77], "actor2": [0, 22, 11],
                         "genre1": ["missing", "y", "z"], "genre2": ["x", "missing", "x"], "genre3": ["y", "y", "z"]})
    new = pd.DataFrame({"user": [11, 1], "item": [4, 1], "label": [1, 0], "sex": ["M", "F"], "occupation": ["b", "d"],
                        "age": [4, 5], "actor1": [11, 88], "actor2": [99, 0], "genre1": ["xx", "missing"],
                        "genre2": ["z", "yy"], "genre3": ["missing", "x"]})
    kw = dict(sparse_col=["sex", "occupation"], multi_sparse_col=[["actor1", "actor2"], ["genre1", "genre2", "genre3"]],
              dense_col=["age"], user_col=["sex", "age", "occupation", "actor1", "actor2"],
              item_col=["genre1", "genre2", "genre3"], pad_val=[0, "missing"])
    _, old = DatasetFeat.build_trainset(data, **kw)
    np.testing.assert_array_equal(old.user_sparse_unique, [[0, 3, 9, 7], [1, 4, 6, 9], [1, 3, 8, 6], [2, 5, 9, 9]])
    np.testing.assert_array_equal(old.item_sparse_unique, [[13, 10, 11], [11, 13, 11], [12, 10, 12], [13, 13, 13]])
    with pytest.raises(ValueError, match="Old column .* doesn't exist in new data"):
        DatasetFeat.merge_trainset(new.drop("sex", axis=1), old)
    _, info = DatasetFeat.merge_trainset(new, old)
    assert list(info.sparse_unique_vals["occupation"]) == ["a", "c", "b", "d"]
    assert list(info.multi_sparse_unique_vals["actor1"]) == [11, 22, 77, 88, 99]
    assert list(info.multi_sparse_unique_vals["genre1"]) == ["x", "y", "z", "xx", "yy"]
    np.testing.assert_array_equal(info.user_unique_vals, [1, 4, 10, 11])
    np.testing.assert_array_equal(info.item_unique_vals, [1, 2, 8, 4])
    np.testing.assert_array_equal(info.sparse_offset, [0, 3, 8, 8, 14, 14, 14])
    np.testing.assert_array_equal(info.sparse_oov, [2, 7, 13, 13, 19, 19, 19])
    # DataInfo appends the OOV row; the reference asserts the matrices before that step
    np.testing.assert_array_equal(info.user_sparse_unique[:-1], [[0, 6, 11, 9], [1, 4, 8, 13], [1, 3, 10, 8], [1, 5, 8, 12]])
    np.testing.assert_array_equal(info.user_dense_unique[:-1], [[5], [1], [3], [4]])
    np.testing.assert_array_equal(info.item_sparse_unique[:-1], [[19, 18, 14], [15, 19, 15], [16, 14, 16], [17, 16, 19]])
    assert info.item_dense_unique is None


def _kat_pair():
    import pandas as pd

    from librecommender_amd.data import DatasetFeat
    data = pd.DataFrame({"user": [4, 1, 10], "item": [1, 2, 8], "label": [1, 0, 1], "sex": ["M", "F", "M"],
                         "occupation": ["c", "a", "a"], "age": [1, 2, 3], "actor1": [11, 0, 77], "actor2": [0, 22, 11],
                         "genre1": ["missing", "y", "z"], "genre2": ["x", "missing", "x"], "genre3": ["y", "y", "z"]})
    new = pd.DataFrame({"user": [11, 1], "item": [4, 1], "label": [1, 0], "sex": ["M", "F"], "occupation": ["b", "d"],
                        "age": [4, 5], "actor1": [11, 88], "actor2": [99, 0], "genre1": ["xx", "missing"],
                        "genre2": ["z", "yy"], "genre3": ["missing", "x"]})
    kw = dict(sparse_col=["sex", "occupation"], multi_sparse_col=[["actor1", "actor2"], ["genre1", "genre2", "genre3"]],
              dense_col=["age"], user_col=["sex", "age", "occupation", "actor1", "actor2"],
              item_col=["genre1", "genre2", "genre3"], pad_val=[0, "missing"])
    _, info = DatasetFeat.build_trainset(data, **kw)
    return info, new


def test_reference_kat_assign_and_extract_features():
    """tests/test_feature.py:539-616: `assign_user/item_features`, `get_original_feats`,
    `set_temp_feats` on the reference's literal frames."""
    from librecommender_amd.bases.feat_base import merge_user_item_feats
    from librecommender_amd.data.retrain import store_old_info
    from librecommender_amd.feature_override import override_dense, override_sparse

    info, new = _kat_pair()
    old = store_old_info(info)
    assert old.n_users == 3 and old.n_items == 3
    sp, dn = merge_user_item_feats(info, [2, 3], [0, 1])                               # :575-585
    np.testing.assert_array_equal(sp, [[1, 3, 8, 6, 13, 10, 11], [2, 5, 9, 9, 11, 13, 11]])
    np.testing.assert_array_equal(dn, [[3.0], [2.0]])
    sp0, dn0 = merge_user_item_feats(info, [0], [0])
    np.testing.assert_array_equal(sp0, [[0, 3, 9, 7, 13, 10, 11]])
    np.testing.assert_array_equal(dn0, [[2.0]])
    sp1, dn1 = merge_user_item_feats(info, [2], [1])                                   # :604-616
    feats = {"occupation": "xxx", "age": 10, "actor1": 111, "actor2": 77, "genre2": "x"}
    np.testing.assert_array_equal(sp1, [[1, 3, 8, 6, 11, 13, 11]])
    np.testing.assert_array_equal(override_sparse(info, sp1, feats), [[1, 3, 8, 8, 11, 10, 11]])
    np.testing.assert_array_equal(override_dense(info, dn1, feats), [[10.0]])
    np.testing.assert_array_equal(sp1, [[1, 3, 8, 6, 11, 13, 11]])                     # originals untouched
    new = new.drop("sex", axis=1)                                                      # :539-572
    new.loc[1, "actor1"] = 77
    assert getattr(info, "feat_version", 0) == 0
    info.assign_user_features(new)
    info.assign_item_features(new)
    assert info.feat_version == 2          # device-side copies of the feature rows are keyed on it (bases/feat_base.py)
    np.testing.assert_array_equal(info.user_sparse_unique, [[0, 3, 8, 7], [1, 4, 6, 9], [1, 3, 8, 6], [2, 5, 9, 9]])
    np.testing.assert_array_equal(info.item_sparse_unique, [[13, 10, 10], [11, 13, 11], [12, 10, 12], [13, 13, 13]])


def test_reference_kat_batch_and_catalog_features():
    """tests/test_feature.py:619-686: `features_from_batch`, `_get_original_feats`, `process_embed_feat`."""
    from librecommender_amd.prediction.preprocess import catalog_features, features_from_batch, user_tower_features

    info, new = _kat_pair()
    for missing in ("sex", "actor1"):
        with pytest.raises(ValueError, match="Column .* doesn't exist in data"):
            features_from_batch(info, True, True, new.drop(missing, axis=1))
    sp, dn = features_from_batch(info, True, True, new)
    np.testing.assert_array_equal(sp, [[1, 5, 6, 9, 13, 12, 13], [0, 5, 9, 9, 13, 13, 10]])
    np.testing.assert_array_equal(dn, [[4.0], [5.0]])
    sp, dn = catalog_features(info, 0, 3)
    np.testing.assert_array_equal(sp, [[0, 3, 9, 7, 13, 10, 11], [0, 3, 9, 7, 11, 13, 11], [0, 3, 9, 7, 12, 10, 12]])
    np.testing.assert_array_equal(dn, [[2.0]] * 3)
    sp, dn = catalog_features(info, 3, 3, dense=False)
    np.testing.assert_array_equal(sp, [[2, 5, 9, 9, 13, 10, 11], [2, 5, 9, 9, 11, 13, 11], [2, 5, 9, 9, 12, 10, 12]])
    assert dn is None
    sp, dn = catalog_features(info, 2, 1, sparse=False)
    assert sp is None
    np.testing.assert_array_equal(dn, [[3.0]])
    sp, dn = user_tower_features(info, np.array([1]),
                                 {"sex": "out", "occ": "a", "actor1": "out", "actor2": 77, "age": 11})
    np.testing.assert_array_equal(sp, [[1, 4, 6, 8]])
    np.testing.assert_array_equal(dn, [[11.0]])


def test_predict_data_with_feats_plumbing():
    """`predict_data_with_feats` (prediction/predict.py:95-150): ids -> inner ids with OOV, features
    encoded from the frame per batch, cold-start handling; the net is replaced by a recording stub."""
    import torch

    from librecommender_amd.prediction import predict_data_with_feats

    info, new = _kat_pair()
    seen = []

    class Stub:
        task, data_info, n_users, n_items, default_pred = "ranking", info, info.n_users, info.n_items, 0.0

        def _cached_seq(self, users):
            return None, None

        def _forward(self, users, items, sparse, dense, seqs, lens):
            seen.append((np.asarray(users).copy(), np.asarray(items).copy(), sparse.copy(), dense.copy()))
            return torch.from_numpy(sparse.sum(1).astype(np.float32) * 0.01 + dense[:, 0])

    out = predict_data_with_feats(Stub(), new, batch_size=1, cold_start="average")
    assert len(seen) == 2
    np.testing.assert_array_equal(np.concatenate([s[0] for s in seen]), [3, 0])        # user 11 unknown -> OOV id
    np.testing.assert_array_equal(np.concatenate([s[1] for s in seen]), [3, 0])        # item 4 unknown
    np.testing.assert_array_equal(np.concatenate([s[2] for s in seen]),
                                  [[1, 5, 6, 9, 13, 12, 13], [0, 5, 9, 9, 13, 13, 10]])
    from scipy.special import expit
    np.testing.assert_allclose(out, expit(np.array([59 * 0.01 + 4.0, 59 * 0.01 + 5.0], np.float32)), rtol=1e-6)
    pop = predict_data_with_feats(Stub(), new, cold_start="popular")
    assert pop[0] == 0.0 and pop[1] == out[1]


def test_reference_kat_invalid_and_role_columns():
    """tests/test_feature.py:39-145: column validation errors and the user/item x sparse/dense roles."""
    from librecommender_amd.data import DatasetFeat
    from librecommender_amd.data.data_info import Feature
    rng = np.random.default_rng(0)
    n = 200
    d = pd.DataFrame({"user": rng.integers(0, 20, n), "item": rng.integers(0, 30, n), "label": rng.integers(1, 6, n),
                      "sex": rng.choice(["M", "F"], n), "age": rng.integers(1, 60, n),
                      "occupation": rng.integers(0, 5, n), "genre1": rng.choice(list("abc"), n),
                      "genre2": rng.choice(list("abc"), n), "genre3": rng.choice(list("abc"), n)})
    with pytest.raises(ValueError, match="Got inconsistent columns"):
        DatasetFeat.build_trainset(d, sparse_col=["genre1", "occupation"], dense_col=["age"],
                                   user_col=["age", "sex"], item_col=["genre1"])
    with pytest.raises(ValueError, match="Please make sure length of columns match"):
        DatasetFeat.build_trainset(d, sparse_col=["genre1", "occupation", "age"], dense_col=["age"],
                                   user_col=["age", "occupation"], item_col=["genre1"])
    with pytest.raises(ValueError, match="Please make sure length of columns match"):
        DatasetFeat.build_trainset(d, multi_sparse_col=[["genre1", "genre2", "genre3"]], sparse_col=[],
                                   dense_col=["age"], user_col=[], item_col=["genre1", "genre2", "genre3"])
    d["item_dense_col"] = rng.integers(0, 10000, n)
    _, info = DatasetFeat.build_trainset(
        d, user_col=["age"], item_col=["genre1", "genre2", "genre3", "item_dense_col"], sparse_col=[],
        dense_col=["age", "item_dense_col"], multi_sparse_col=[["genre1", "genre2", "genre3"]], shuffle=False)
    assert info.user_sparse_col == Feature(name=[], index=[])
    assert info.user_dense_col == Feature(name=["age"], index=[0])
    assert info.item_sparse_col == Feature(name=["genre1", "genre2", "genre3"], index=[0, 1, 2])
    assert info.item_dense_col == Feature(name=["item_dense_col"], index=[1])
    assert info.user_col == ["age"] and info.item_col == ["genre1", "genre2", "genre3", "item_dense_col"]
    assert info.sparse_col == Feature(name=["genre1", "genre2", "genre3"], index=[0, 1, 2])
    assert info.dense_col == Feature(name=["age", "item_dense_col"], index=[0, 1])


def test_process_data_and_split_multi_value_match_reference(golden_dir):
    """data/processing.py of the reference, outputs stored by `oracle.make_golden.gen_processing`."""
    from librecommender_amd.data import process_data, split_multi_value
    from oracle.make_golden import multi_value_frame
    g = np.load(golden_dir / "processing.npz")
    with pytest.raises(ValueError):
        process_data(synthetic_frame(), dense_col="age")
    with pytest.raises(ValueError):
        process_data(synthetic_frame(), dense_col=["age"], normalizer="unknown")
    for norm in ("min_max", "standard", "robust", "power"):
        one = synthetic_frame()
        _, cols = process_data(one, dense_col=["age", "profit"], normalizer=norm)
        assert cols == g[f"{norm}_one_cols"].tolist()
        for c in cols:
            np.testing.assert_allclose(one[c].to_numpy(np.float64), g[f"{norm}_one_{c}"], rtol=1e-12, atol=1e-12)
        tr, ev = synthetic_frame().iloc[:150].copy(), synthetic_frame().iloc[150:].copy()
        _, cols = process_data((tr, ev), dense_col=["age", "label"], normalizer=norm, transformer=("log", "square"))
        assert cols == g[f"{norm}_pair_cols"].tolist()
        for tag, fr in (("tr", tr), ("ev", ev)):
            assert list(fr.columns) == g[f"{norm}_pair_{tag}_columns"].tolist()
            for c in fr.columns:
                if c.startswith(("age", "label")):
                    np.testing.assert_allclose(fr[c].to_numpy(np.float64), g[f"{norm}_pair_{tag}_{c}"], rtol=1e-7)
    for tag, kw in (("auto", dict(max_len=None, pad_val="missing")),
                    ("capped", dict(max_len=[2, 3], pad_val=["nil", "none"]))):
        frame = multi_value_frame()
        frame["tag"] = frame["tag"].str.replace(",", "|")
        d, multi, ucol, icol = split_multi_value(frame, ["genre", "tag"], "|", user_col=["tag"], item_col=["genre"], **kw)
        assert list(d.columns) == g[f"mv_{tag}_columns"].tolist()
        assert ["/".join(x) for x in multi] == g[f"mv_{tag}_multi"].tolist()
        assert ucol == g[f"mv_{tag}_user"].tolist() and icol == g[f"mv_{tag}_item"].tolist()
        for c in d.columns:
            np.testing.assert_array_equal(d[c].to_numpy().astype(str), g[f"mv_{tag}_col_{c}"])
    with pytest.raises(AssertionError):
        split_multi_value(multi_value_frame(), ["genre"], "|", max_len=3)
    with pytest.raises(AssertionError):
        split_multi_value(multi_value_frame(), ["genre", "tag"], "|", max_len=[3])


def test_collators_with_partial_feature_sets(golden_dir):
    """The reference's collators on data sets with no features / user-side only / item-side only
    (the parametrisation of its tests/test_collators.py), first batch of a seeded loader, bit-exact."""
    from oracle.make_golden import PARTIAL_FEATURE_CONFIGS, partial_collator_cases
    g = np.load(golden_dir / "collators_partial.npz")
    train, _ = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    for cfg, kw in PARTIAL_FEATURE_CONFIGS.items():
        ts, info = DatasetFeat.build_trainset(train_data=train.copy(), **kw)
        for tag, m in partial_collator_cases(info, _stub):
            loader = get_batch_loader(m, ts, True, batch_size=24, shuffle=True, num_workers=0, seed=42)
            _compare_batch(f"{cfg}_{tag}", next(iter(loader)), g)


def test_consumed_index_csr():
    """`ConsumedIndex.batch_csr`: ascending unique ids of the users that get filtered — the rule of
    ranking.py:38 on the RAW history length — unknown / OOV users and `filter_consumed=False` empty."""
    from librecommender_amd.recommendation import ConsumedIndex
    uc = {0: [5, 3, 5, 9], 1: [], 2: [1], 7: [4]}                  # user 7 is outside n_users=4
    ci = ConsumedIndex(uc, 4)
    ptr, idx, flag = ci.batch_csr([0, 1, 2, 3, 4, 0], n_rec=2, n_items=10, filter_consumed=True, device="cpu")
    assert ptr.tolist() == [0, 3, 3, 4, 4, 4, 7] and idx.tolist() == [3, 5, 9, 1, 3, 5, 9] and flag.tolist() == [1, 0, 1, 0, 0, 1]
    assert ptr.dtype.is_floating_point is False and str(idx.dtype) == "torch.int32" and str(flag.dtype) == "torch.uint8"
    ptr, idx, flag = ci.batch_csr([0, 2], n_rec=7, n_items=10, filter_consumed=True, device="cpu")   # 7 + 4 > 10: user 0 unfiltered
    assert ptr.tolist() == [0, 0, 1] and idx.tolist() == [1] and flag.tolist() == [0, 1]
    ptr, idx, flag = ci.batch_csr([0, 2], n_rec=2, n_items=10, filter_consumed=False, device="cpu")
    assert ptr.tolist() == [0, 0, 0] and flag.tolist() == [0, 0]
    assert ci.consumed(0).tolist() == [3, 5, 9] and ci.consumed(1) is None and ci.consumed(9) is None


def test_reference_import_paths_and_console_helpers():
    """Names that code written against the reference imports from where the reference keeps them
    (`data/data_info.py:540-578`, `utils/misc.py:46-73`), and `time_block` / `time_func` behaviour
    (`tests/test_misc.py` of the reference): nothing printed for a block that raises, the exception propagates."""
    import io
    from contextlib import redirect_stdout

    import pytest

    from librecommender_amd.data import data_info as di
    from librecommender_amd.data import retrain
    from librecommender_amd.utils.misc import colorize, time_block, time_func

    assert di.OldInfo is retrain.OldInfo and di.store_old_info is retrain.store_old_info
    with pytest.raises(AttributeError):
        di.no_such_name

    @time_func
    def work(x):
        return x + 1

    buf = io.StringIO()
    with redirect_stdout(buf):
        assert work(1) == 2
        with time_block("quiet", verbose=0):
            pass
        with time_block("loud"):
            pass
        with pytest.raises(RuntimeError):
            with time_block("failing"):
                raise RuntimeError
    out = buf.getvalue()
    assert "work elapsed" in out and "loud elapsed" in out and "quiet" not in out and "failing" not in out
    assert colorize("x", "red", bold=True, highlight=True).endswith("\x1b[0m")
