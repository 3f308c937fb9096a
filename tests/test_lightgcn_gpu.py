"""LightGCN on the HIP path vs fixtures produced by the REFERENCE module itself
(tests/golden/lightgcn.npz: `LightGCNModel` + `bpr_loss` + `torch.optim.Adam`, `-m gpu`)."""
import numpy as np
import pytest
import torch

from librecommender_amd.algorithms import LightGCN
from librecommender_amd.data import DatasetPure, split_by_ratio_chrono
from librecommender_amd.nets.graph_nets import LightGCNNet, build_laplacian_csr, cosine_warm_restart_lr
from tests.golden_util import unflatten
from tests.test_api_gpu import check_preds, check_recommends, movielens_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fuse_adam", [False, True])
def test_reference_module_fixture(dev, golden_dir, fuse_adam):
    """`fuse_adam`: the optimiser step as the epilogue of the last backward product (no gradient table) — the same step."""
    g = np.load(golden_dir / "lightgcn.npz")
    nu, ni, L = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    uc = unflatten(g["user_consumed_flat"])
    net = LightGCNNet(nu, ni, 16, L, 0.0, uc, dev, seed=42, lr=1e-2, epsilon=1e-8)
    net.fuse_adam = fuse_adam
    # same RNG protocol -> same initial embeddings as the reference module
    np.testing.assert_array_equal(net.E[:nu].cpu().numpy(), g["U0"])
    np.testing.assert_array_equal(net.E[nu:].cpu().numpy(), g["I0"])
    # Laplacian == reference COO
    import scipy.sparse as ssp
    ref = ssp.coo_matrix((g["lap_vals"], (g["lap_rows"], g["lap_cols"])), shape=(nu + ni, nu + ni)).tocsr()
    mine = ssp.csr_matrix((net.val.cpu().numpy(), net.col.cpu().numpy(), net.rowptr.cpu().numpy()), shape=(nu + ni, nu + ni))
    assert (abs(ref - mine) > 1e-7).nnz == 0
    ue, ie = net.embeddings()
    np.testing.assert_allclose(ue.cpu().numpy(), g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.cpu().numpy(), g["item_embeds"], rtol=1e-5, atol=1e-6)
    loss, G = net.train_step("bpr", g["users"], g["pos"], items_neg=g["neg"])
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    if fuse_adam and L >= 2:
        assert G is None                                                                  # never materialised
    else:
        np.testing.assert_allclose(G[:nu].cpu().numpy(), g["gU"], rtol=1e-4, atol=1e-7)   # gradients 1e-4
        np.testing.assert_allclose(G[nu:].cpu().numpy(), g["gI"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(net.E[:nu].cpu().numpy(), g["U1"], rtol=1e-4, atol=2e-6)  # one torch-Adam step
    np.testing.assert_allclose(net.E[nu:].cpu().numpy(), g["I1"], rtol=1e-4, atol=2e-6)


def test_transpose_map_and_dropout_backward(dev):
    rng = np.random.default_rng(0)
    nu, ni = 40, 50
    uc = {u: rng.integers(0, ni, 6).tolist() for u in range(nu)}
    rp, ci, va, tp = build_laplacian_csr(nu, ni, uc)
    import scipy.sparse as ssp
    w = rng.random(len(va)).astype(np.float32)           # asymmetric values on the symmetric pattern
    A = ssp.csr_matrix((w, ci, rp), shape=(nu + ni, nu + ni))
    At = ssp.csr_matrix((w[tp], ci, rp), shape=(nu + ni, nu + ni))
    assert abs(A.T - At).max() == 0
    net = LightGCNNet(nu, ni, 16, 2, 0.3, uc, dev, seed=1, lr=1e-2)
    net.fuse_adam = False                                  # (the gradient table is looked at)
    torch.manual_seed(0)
    loss, G = net.train_step("max_margin", rng.integers(0, nu, 8), rng.integers(0, ni, 8), items_neg=rng.integers(0, ni, 16))
    assert torch.isfinite(loss) and torch.isfinite(G).all()
    # the fused optimiser epilogue, edge dropout and AMSGrad included: the same tables as the two-launch form, bit for bit
    pair = []
    for fuse in (False, True):
        n2 = LightGCNNet(nu, ni, 16, 3, 0.3, uc, dev, seed=1, lr=1e-2, amsgrad=True)
        n2.fuse_adam = fuse
        r2 = np.random.default_rng(4)
        for _ in range(3):
            torch.manual_seed(7)
            n2.train_step("bpr", r2.integers(0, nu, 8), r2.integers(0, ni, 8), items_neg=r2.integers(0, ni, 8))
        pair.append((n2.E.clone(), n2.m.clone(), n2.v.clone(), n2.vmax.clone()))
    for a_, b_ in zip(*pair):
        assert torch.equal(a_, b_)


def test_cosine_warm_restart_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.01)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=1, T_mult=2)
    for e in (0.0, 0.3, 0.99, 1.0, 1.7, 2.9, 3.0, 5.5, 7.0):
        sch.step(e)
        assert abs(opt.param_groups[0]["lr"] - cosine_warm_restart_lr(0.01, e)) < 1e-9


@pytest.mark.parametrize("loss_type,num_neg,lr_decay", [("bpr", 1, False), ("max_margin", 2, True), ("cross_entropy", 1, False), ("focal", 2, False)])
def test_lightgcn_api(dev, loss_type, num_neg, lr_decay):
    df = movielens_like(8000, 300, 400)
    train, evald = split_by_ratio_chrono(df, test_size=0.2)
    train_data, info = DatasetPure.build_trainset(train)
    eval_data = DatasetPure.build_evalset(evald)
    model = LightGCN("ranking", info, loss_type=loss_type, embed_size=16, n_epochs=2, lr=1e-2,
                     lr_decay=lr_decay, batch_size=1024, num_neg=num_neg, dropout_rate=0.1 if num_neg == 2 else 0.0)
    model.fit(train_data, neg_sampling=True, verbose=2, eval_data=eval_data, metrics=["roc_auc", "recall"])
    check_preds(model, train)
    check_recommends(model, info, train)
    with pytest.raises(ValueError):
        LightGCN("rating", info)
    with pytest.raises(ValueError):
        LightGCN("ranking", info, loss_type="nce")


def test_load_reference_inference_checkpoint(dev, golden_dir, tmp_path):
    """An inference checkpoint + DataInfo written by the REFERENCE's LightGCN / DataInfo.save
    (tests/golden/refckpt/, oracle/make_golden.py:gen_ref_checkpoint) loads here; predictions and
    recommendations served by `lr_pair_dot_f32` / `lr_score_topk_f32` equal the reference's; our
    own inference checkpoint has the same layout and round-trips."""
    from librecommender_amd.data import DataInfo

    d = golden_dir / "refckpt"
    info = DataInfo.load(str(d), "lgcn")
    model = LightGCN.load(str(d), "lgcn", info)
    exp = np.load(d / "expected.npz")
    np.testing.assert_allclose(model.predict(exp["pred_user"], exp["pred_item"]), exp["preds"], rtol=1e-5, atol=1e-6)
    users = exp["users"].tolist()
    recs = model.recommend_user(users, n_rec=7)
    np.testing.assert_array_equal(np.stack([recs[u] for u in users]), exp["recs"])
    cold = model.recommend_user(-12345, n_rec=5, cold_start="popular")[-12345]
    assert set(cold.tolist()) <= set(info.popular_items)
    # evaluate() on the reference's checkpoint == evaluate() of the reference (same predictions,
    # same eval negatives, same user sample)
    from librecommender_amd.evaluation import evaluate
    from oracle.make_golden import synthetic_frame
    _, ev_df = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    ev_df = ev_df[["user", "item", "label"]]
    names = ["loss", "balanced_accuracy", "roc_auc", "pr_auc", "precision", "recall", "map", "ndcg"]
    res = evaluate(model, ev_df, neg_sampling=True, metrics=names, k=5, seed=42)
    np.testing.assert_allclose([res[m] for m in names], exp["eval_vals"][:8], rtol=1e-5, atol=1e-7)
    res2 = evaluate(model, ev_df, neg_sampling=True, metrics=["roc_auc", "ndcg"], k=5, sample_user_num=7, seed=3)
    np.testing.assert_allclose([res2["roc_auc"], res2["ndcg"]], exp["eval_vals"][8:], rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError):
        model.fit(None, neg_sampling=True)                       # loaded models are inference-only
    model.save(str(tmp_path), "again", inference_only=True)
    import os
    assert sorted(os.listdir(tmp_path)) == ["again.npz", "again_default_recs.npz", "again_hyper_parameters.json"]
    m2 = LightGCN.load(str(tmp_path), "again", info)
    np.testing.assert_array_equal(np.stack([m2.recommend_user(users, n_rec=7)[u] for u in users]), exp["recs"])


def test_full_fit_matches_reference_fit(dev, golden_dir):
    """`LightGCN.fit` here vs the REFERENCE's own fit (2 epochs, BPR, seed 42, CPU) on the same data:
    same initial embeddings, same batch order, same negatives (host loader / samplers are
    bit-exact), HIP kernels for propagation, gather, scatter and Adam -> the trained embeddings of
    tests/golden/refckpt/lgcn.npz within fp32 training noise."""
    from librecommender_amd.data import DatasetPure
    from oracle.make_golden import synthetic_frame

    df, ev_df = split_by_ratio_chrono(synthetic_frame(), test_size=0.2)
    train, info = DatasetPure.build_trainset(df[["user", "item", "label"]])
    model = LightGCN("ranking", info, loss_type="bpr", embed_size=8, n_epochs=2, lr=1e-2, batch_size=64,
                     num_neg=1, seed=42)
    model.fit(train, neg_sampling=True, verbose=0)
    ref = np.load(golden_dir / "refckpt" / "lgcn.npz")
    np.testing.assert_allclose(model.user_embeds_np, ref["user_embed"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(model.item_embeds_np, ref["item_embed"], rtol=1e-3, atol=2e-5)
    exp = np.load(golden_dir / "refckpt" / "expected.npz")
    users = exp["users"].tolist()
    recs = model.recommend_user(users, n_rec=7)
    assert np.mean(np.stack([recs[u] for u in users]) == exp["recs"]) > 0.9       # near-tied scores may swap
    # the whole evaluation pipeline (eval negatives, predict, recommend, sklearn metrics) on held-out data
    from librecommender_amd.evaluation import evaluate
    names = ["loss", "balanced_accuracy", "roc_auc", "pr_auc", "precision", "recall", "map", "ndcg"]
    res = evaluate(model, ev_df[["user", "item", "label"]], neg_sampling=True, metrics=names, k=5, seed=42)
    np.testing.assert_allclose([res[m] for m in names], exp["eval_vals"][:8], rtol=2e-2, atol=2e-3)
