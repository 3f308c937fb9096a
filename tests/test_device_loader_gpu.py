"""Device-side pointwise batches: layout and feature semantics of the host
`PointwiseCollator` (batch/collators.py:225-274), and end-to-end training through them."""
import numpy as np
import pytest
import torch

from librecommender_amd.algorithms import FM, DeepFM
from librecommender_amd.batch.device_loader import DevicePointwiseLoader, device_loader_supported
from librecommender_amd.data import DatasetFeat, DatasetPure, split_by_ratio_chrono
from librecommender_amd.evaluation import evaluate
from oracle.make_golden import FEAT_KW, synthetic_frame
from tests.test_api_gpu import movielens_like

pytestmark = pytest.mark.gpu


def test_batch_layout_and_features(dev):
    df = synthetic_frame()
    train_data, info = DatasetFeat.build_trainset(df, **FEAT_KW)
    model = DeepFM("ranking", info, embed_size=16, n_epochs=1, batch_size=64, num_neg=2, sampler="unconsumed",
                   device_sampling=True)
    model.build_model()
    assert device_loader_supported(model, True)
    loader = DevicePointwiseLoader(model, train_data, 20, shuffle=True, seed=3)
    uc0 = info.user_sparse_col.index
    train_rows = {(int(u), int(i), tuple(r[uc0].tolist())) for u, i, r in
                  zip(train_data.user_indices, train_data.item_indices, train_data.sparse_indices)}
    seen = 0
    for b in loader:
        k = 3
        users, items, labels = b.users.cpu().numpy(), b.items.cpu().numpy(), b.labels.cpu().numpy()
        assert len(users) % k == 0
        np.testing.assert_array_equal(labels.reshape(-1, k), np.tile([1.0, 0.0, 0.0], (len(users) // k, 1)))
        np.testing.assert_array_equal(users.reshape(-1, k)[:, 0:1].repeat(k, 1), users.reshape(-1, k))
        it = items.reshape(-1, k)
        assert (it[:, 1:] != it[:, :1]).all() and (it[:, 1] != it[:, 2]).all()
        for u, row in zip(users.reshape(-1, k)[:, 0], it):
            assert row[0] in info.user_consumed[u]                      # the positive is a train interaction
        sp, dn = b.sparse_indices.cpu().numpy(), b.dense_values.cpu().numpy()
        ic, uc = info.item_sparse_col.index, info.user_sparse_col.index
        np.testing.assert_array_equal(sp[:, ic], info.item_sparse_unique[items])      # collators.py:468-477
        # user-side columns: the positive's own training row, repeated (collators.py:241-246)
        usp = sp[:, uc].reshape(-1, k, len(uc))
        np.testing.assert_array_equal(usp, usp[:, :1].repeat(k, 1))
        for u, row, f in zip(users.reshape(-1, k)[:, 0], it, usp[:, 0]):
            assert (int(u), int(row[0]), tuple(f.tolist())) in train_rows
        np.testing.assert_allclose(dn[:, info.item_dense_col.index], info.item_dense_unique[items])
        udn = dn[:, info.user_dense_col.index].reshape(-1, k, len(info.user_dense_col.index))
        np.testing.assert_array_equal(udn, udn[:, :1].repeat(k, 1))
        seen += len(users) // k
    assert seen == len(train_data)                                          # every positive exactly once


@pytest.mark.parametrize("cls,feat", [(DeepFM, False), (FM, True), (DeepFM, True)])
def test_fit_with_device_sampling(dev, cls, feat):
    if feat:
        df = synthetic_frame()
        train, evald = split_by_ratio_chrono(df, test_size=0.2)
        train_data, info = DatasetFeat.build_trainset(train, **FEAT_KW)
        eval_data = DatasetFeat.build_evalset(evald)
    else:
        df = movielens_like(6000, 200, 150)
        train, evald = split_by_ratio_chrono(df, test_size=0.2)
        train_data, info = DatasetPure.build_trainset(train)
        eval_data = DatasetPure.build_evalset(evald)
    model = cls("ranking", info, embed_size=16, n_epochs=3, lr=1e-2, batch_size=256, num_neg=1,
                device_sampling=True)
    model.fit(train_data, neg_sampling=True, verbose=0)
    res = evaluate(model, eval_data, neg_sampling=True, metrics=["loss", "roc_auc"])
    assert np.isfinite(res["loss"]) and 0.0 <= res["roc_auc"] <= 1.0
    rec = model.recommend_user(info.id2user[0], 5)
    assert len(rec[info.id2user[0]]) == 5
    # same seed -> same training run (counter-based sampler, seeded device permutation)
    model2 = cls("ranking", info, embed_size=16, n_epochs=3, lr=1e-2, batch_size=256, num_neg=1,
                 device_sampling=True)
    model2.fit(train_data, neg_sampling=True, verbose=0)
    torch.testing.assert_close(model.net.tables.embed, model2.net.tables.embed, rtol=0, atol=0)
