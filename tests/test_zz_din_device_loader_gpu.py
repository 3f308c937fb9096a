"""DIN trained through the device-side loader (`device_sampling=True`: negatives from
`lr_sample_negatives_i32`, collation and behaviour sequences as device tensor code; row f1).  The
batch semantics are checked on CPU tensors in tests/test_device_sequences_cpu.py.  (Added after this
round's GPU budget was spent: first run is the driver's.)"""
import numpy as np
import pytest

from librecommender_amd.algorithms import DIN
from librecommender_amd.batch.device_loader import DevicePointwiseLoader, device_loader_supported
from librecommender_amd.batch.sequence import SequenceBuilder
from librecommender_amd.data import DatasetFeat, split_by_ratio_chrono
from librecommender_amd.evaluation import evaluate
from oracle.make_golden import FEAT_KW, synthetic_frame

pytestmark = pytest.mark.gpu


def test_din_fit_with_device_sampling(dev):
    train, evald = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    train_data, info = DatasetFeat.build_trainset(train, **FEAT_KW)
    eval_data = DatasetFeat.build_evalset(evald)
    kw = dict(embed_size=16, n_epochs=2, lr=1e-2, batch_size=128, num_neg=1, recent_num=5, device_sampling=True)
    model = DIN("ranking", info, **kw)
    model.build_model()
    assert device_loader_supported(model, True)
    batch = next(iter(DevicePointwiseLoader(model, train_data, 64, shuffle=True, seed=1)))
    users, items = batch.users.cpu().numpy(), batch.items.cpu().numpy()
    host = SequenceBuilder(info.user_consumed, info.n_items, 5, "recent")
    pos = host.positions(users, items)
    import random
    random.seed(0)
    h_seqs, h_lens = host.training_seqs(users, items)
    known = pos >= 0
    np.testing.assert_array_equal(batch.seqs.interacted_seq.cpu().numpy()[known], h_seqs[known])
    np.testing.assert_array_equal(batch.seqs.interacted_len.cpu().numpy()[known], h_lens[known])
    model = DIN("ranking", info, **kw)
    model.fit(train_data, neg_sampling=True, verbose=0)
    res = evaluate(model, eval_data, neg_sampling=True, metrics=["loss", "roc_auc"])
    assert np.isfinite(res["loss"]) and 0.0 <= res["roc_auc"] <= 1.0
    again = DIN("ranking", info, **kw)
    again.fit(train_data, neg_sampling=True, verbose=0)
    pu, pi = train["user"].to_numpy()[:50], train["item"].to_numpy()[:50]
    np.testing.assert_allclose(model.predict(pu, pi), again.predict(pu, pi), rtol=1e-2, atol=1e-3)   # seeded run
