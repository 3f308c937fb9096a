"""NGCF net logic (Laplacian + transpose map, autograd wiring of the sparse product / batch-row
gather, Adam incl. weight decay + AMSGrad, init RNG protocol) on CPU with the oracle kernels
injected, against the REFERENCE module itself (tests/golden/ngcf.npz, oracle.make_golden.gen_ngcf).
The HIP kernels behind the same provider interface have their own GPU parity tests."""
import numpy as np
import pytest
import torch

from librecommender_amd.nets.ngcf_net import NGCFNet, build_ngcf_laplacian_csr
from tests.golden_util import unflatten
from tests.oracle_kernels import OracleKernels


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "ngcf.npz")


def _net(g, **kw):
    return NGCFNet(int(g["n_users"]), int(g["n_items"]), 8, g["layers"].tolist(), 0.0, 0.0,
                   unflatten(g["user_consumed_flat"]), torch.device("cpu"), seed=42, lr=1e-2, epsilon=1e-8,
                   kern=OracleKernels(), **kw)


def test_laplacian_matches_reference(g):
    rp, col, val, tperm = build_ngcf_laplacian_csr(int(g["n_users"]), int(g["n_items"]), unflatten(g["user_consumed_flat"]))
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    order = np.lexsort((g["lap_cols"], g["lap_rows"]))
    np.testing.assert_array_equal(rows, g["lap_rows"][order])
    np.testing.assert_array_equal(col, g["lap_cols"][order])
    np.testing.assert_array_equal(val, g["lap_vals"][order])                      # bit-exact fp32 values
    n = len(rp) - 1
    dense = np.zeros((n, n), np.float32)
    dense[rows, col] = val
    dense_t = np.zeros((n, n), np.float32)
    dense_t[rows, col] = val[tperm]
    np.testing.assert_array_equal(dense_t, dense.T)


def test_init_and_propagation_match_reference(g):
    net = _net(g)
    np.testing.assert_array_equal(net.params["embed"].numpy(), g["init_embed"])
    for k in ("W_self_0", "b_self_0", "W_pair_0", "b_pair_0", "W_self_1", "W_pair_1"):
        np.testing.assert_array_equal(net.params[k].numpy(), g[f"init_{k}"])
    ue, ie = net.embeddings()
    assert ue.shape[1] == net.out_dim == 8 + 16 + 12
    np.testing.assert_allclose(ue.numpy(), g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.numpy(), g["item_embeds"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("wd_ams", dict(reg=0.01, amsgrad=True))])
def test_two_train_steps_match_reference(g, tag, kw):
    net = _net(g, **kw)
    loss, grads = net.train_step("bpr", g["users"], g["pos"], items_neg=g["neg"])
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    for k, gr in grads.items():
        np.testing.assert_allclose(gr.numpy(), g[f"{tag}_g_{k}"], rtol=1e-4, atol=1e-7, err_msg=k)
    net.train_step("bpr", g["users"], g["pos"], items_neg=g["neg"])
    for k, p in net.params.items():
        np.testing.assert_allclose(p.numpy(), g[f"{tag}_{k}2"], rtol=1e-4, atol=2e-6, err_msg=k)


def test_full_fit_matches_reference_fit(g, monkeypatch):
    """`NGCF.fit` (trainer, loader, negative sampler, lr handling, per-batch step) against the
    REFERENCE's own 2-epoch fit on the same data; the net runs on CPU with the oracle kernels
    injected and the device-only epilogue (default recommendations) is skipped."""
    from librecommender_amd.algorithms import NGCF
    from librecommender_amd.algorithms import ngcf as ngcf_mod
    from librecommender_amd.data import DatasetPure, split_by_ratio_chrono
    from oracle.make_golden import synthetic_frame

    df, _ = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    train, info = DatasetPure.build_trainset(df[["user", "item", "label"]])
    with pytest.raises(ValueError):
        NGCF("rating", info)
    with pytest.raises(ValueError):
        NGCF("ranking", info, loss_type="whatever")
    model = NGCF("ranking", info, loss_type="bpr", embed_size=8, n_epochs=2, lr=1e-2, batch_size=64, num_neg=1,
                 hidden_units=(16, 16), seed=42)

    def build_on_cpu():
        model.device = torch.device("cpu")
        model.net = ngcf_mod.NGCFNet(model.n_users, model.n_items, 8, model.hidden_units, 0.0, 0.0, model.user_consumed,
                                     model.device, 42, 1e-2, 1e-8, None, 1.0, kern=OracleKernels())
    monkeypatch.setattr(model, "build_model", build_on_cpu)
    monkeypatch.setattr(model, "after_fit", model.set_embeddings)
    model.fit(train, neg_sampling=True, verbose=0)
    np.testing.assert_allclose(model.user_embeds.numpy(), g["fit_user_embed"][:-1], rtol=1e-3, atol=2e-5)       # fixture has the OOV (mean) row
    np.testing.assert_allclose(model.item_embeds.numpy(), g["fit_item_embed"][:-1], rtol=1e-3, atol=2e-5)
    hp = model._hparams()
    assert hp["hidden_units"] == [16, 16] or hp["hidden_units"] == (16, 16)
    assert set(model.variables_np()) == {f"var::{k}" for k in model.net.params}


def test_rebuild_model_moves_rows_and_optimizer_state(monkeypatch, tmp_path):
    """Retrain flow (`torchops/rebuild.py:13-105`): save full state, merge new data (new users and
    items), `rebuild_model` -> known rows and their Adam moments land at their new positions,
    layer weights and step count are taken over, new rows keep the fresh initialisation."""
    from librecommender_amd.algorithms import NGCF
    from librecommender_amd.algorithms import ngcf as ngcf_mod
    from librecommender_amd.data import DatasetPure
    from oracle.make_golden import retrain_frames

    old_df, new_df = retrain_frames()
    cols = ["user", "item", "label"]
    train, info = DatasetPure.build_trainset(old_df[cols])

    def on_cpu(model):
        def build():
            model.device = torch.device("cpu")
            model.net = ngcf_mod.NGCFNet(model.n_users, model.n_items, 8, model.hidden_units, 0.0, 0.0,
                                         model.user_consumed, model.device, 42, 1e-2, 1e-8, None, 1.0,
                                         amsgrad=True, kern=OracleKernels())
        monkeypatch.setattr(model, "build_model", build)
        monkeypatch.setattr(model, "after_fit", lambda: (model.set_embeddings(), model.assign_embedding_oov()))
        return model

    kw = dict(loss_type="bpr", embed_size=8, n_epochs=1, lr=1e-2, batch_size=64, hidden_units=(16,), amsgrad=True)
    model = on_cpu(NGCF("ranking", info, **kw))
    model.fit(train, neg_sampling=True, verbose=0)
    model.save(str(tmp_path), "ngcf", inference_only=False)
    old_net = model.net
    _, merged = DatasetPure.merge_trainset(new_df[cols], info, merge_behavior=True)
    assert merged.n_users > info.n_users and merged.n_items > info.n_items
    new = on_cpu(NGCF("ranking", merged, **kw))
    new.rebuild_model(str(tmp_path), "ngcf")
    net = new.net
    ou, oi, nu = info.n_users, info.n_items, merged.n_users
    for group_new, group_old in ((net.params, old_net.params), (net.m, old_net.m), (net.v, old_net.v), (net.vmax, old_net.vmax)):
        np.testing.assert_array_equal(group_new["embed"][:ou].numpy(), group_old["embed"][:ou].numpy())
        np.testing.assert_array_equal(group_new["embed"][nu:nu + oi].numpy(), group_old["embed"][ou:].numpy())
        np.testing.assert_array_equal(group_new["W_pair_0"].numpy(), group_old["W_pair_0"].numpy())
    assert net.step == old_net.step > 0
    assert float(net.m["embed"][ou:nu].abs().sum()) == 0.0                # new users: zero moments
    fresh = ngcf_mod.NGCFNet(merged.n_users, merged.n_items, 8, [16], 0.0, 0.0, merged.user_consumed,
                             torch.device("cpu"), 42, kern=OracleKernels())
    np.testing.assert_array_equal(net.params["embed"][ou:nu].numpy(), fresh.params["embed"][ou:nu].numpy())
    new.fit(DatasetPure.merge_trainset(new_df[cols], info, merge_behavior=True)[0], neg_sampling=True, verbose=0)
    assert new.user_embeds.shape == (merged.n_users + 1, 8 + 16) and torch.isfinite(new.user_embeds).all()


def test_dropout_paths_and_all_losses_run():
    """Node / message dropout (the transposed operator is rebuilt from the dropped values), weight
    decay + AMSGrad, every loss incl. several negatives per positive: finite losses, finite output."""
    rng = np.random.default_rng(0)
    uc = {u: rng.integers(0, 30, 5).tolist() for u in range(20)}
    net = NGCFNet(20, 30, 8, [16], 0.3, 0.2, uc, torch.device("cpu"), seed=1, lr=1e-3, reg=0.01, amsgrad=True,
                  kern=OracleKernels())
    before = net.params["W_self_0"].clone()
    for loss in ("max_margin", "bpr", "cross_entropy", "focal"):
        if loss in ("cross_entropy", "focal"):
            value, _ = net.train_step(loss, rng.integers(0, 20, 12), rng.integers(0, 30, 12),
                                      labels=rng.integers(0, 2, 12).astype(np.float32))
        else:
            value, _ = net.train_step(loss, rng.integers(0, 20, 6), rng.integers(0, 30, 6), items_neg=rng.integers(0, 30, 12))
        assert np.isfinite(float(value))
    ue, ie = net.embeddings()
    assert ue.shape == (20, 24) and ie.shape == (30, 24) and torch.isfinite(ue).all() and torch.isfinite(ie).all()
    assert not torch.equal(before, net.params["W_self_0"]) and net.step == 4
    val, val_t = net._edge_values(use_dropout=True)
    n = len(net.rowptr) - 1
    rows = torch.repeat_interleave(torch.arange(n), net.rowptr[1:] - net.rowptr[:-1])
    A = torch.zeros(n, n).index_put_((rows, net.col.long()), val)
    At = torch.zeros(n, n).index_put_((rows, net.col.long()), val_t)
    torch.testing.assert_close(At, A.t())                     # dropped edges stay consistent in L^T
