"""Pin the CPU oracle: (i) the reference's own known-answer tests, restated verbatim;
(ii) fixtures produced by running the reference's code (oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import ops_np
from tests.golden_util import unflatten


def test_rank_recommendations_reference_kat():
    """tests/test_rank_reco.py:7-87 of the reference."""
    user_ids = [1, 2]
    preds = np.array([-0.1, -0.01, 0, 0.1, 0.01, 1, -2, 4, 5, 6])
    consumed = {1: [3, 4], 2: [4]}
    with pytest.raises(ValueError):
        ops_np.rank_recommendations(user_ids, preds, 12, 5, consumed)
    ids, _ = ops_np.rank_recommendations(user_ids, preds, 2, 5, consumed)
    np.testing.assert_array_equal(ids, [[2, 1], [3, 2]])
    ids, _ = ops_np.rank_recommendations(user_ids, preds, 4, 5, consumed)  # can't-filter branch
    np.testing.assert_array_equal(ids, [[3, 4, 2, 1], [3, 2, 0, 1]])
    ids, scores = ops_np.rank_recommendations(user_ids, preds.reshape(2, 5), 2, 5, consumed)
    np.testing.assert_array_equal(ids, [[2, 1], [3, 2]])
    assert np.all(np.diff(scores, axis=1) <= 0)


def test_interaction_consumed_reference_kat():
    """tests/test_consumed.py:12-25 and rust/src/utils.rs:41-59: consecutive-duplicate removal."""
    u = [1, 1, 1, 2, 2, 1, 2, 3, 2, 3]
    i = [11, 11, 999, 0, 11, 11, 999, 11, 999, 0]
    uc, ic = ops_np.interaction_consumed(u, i)
    assert uc == {1: [11, 999, 11], 2: [0, 11, 999], 3: [11, 0]}
    assert ic == {11: [1, 2, 1, 3], 999: [1, 2], 0: [2, 3]}


def test_rank_against_reference_outputs(golden_dir):
    g = np.load(golden_dir / "rank_recommendations.npz")
    for ci in range(3):
        U, I = g[f"c{ci}_U"], g[f"c{ci}_I"]
        users = g[f"c{ci}_users"].tolist()
        k = int(g[f"c{ci}_k"])
        consumed = unflatten(g[f"c{ci}_consumed_flat"])
        N = I.shape[0] - 1
        for filt in (1, 0):
            ids, scores = ops_np.recommend_from_embedding(U, I, users, k, N, consumed, bool(filt))
            np.testing.assert_array_equal(ids, g[f"c{ci}_f{filt}_ids"])
        _, scores = ops_np.recommend_from_embedding(U, I, users, k, N, consumed, True)
        from scipy.special import expit
        np.testing.assert_allclose(expit(scores), g[f"c{ci}_scores"], rtol=1e-6)


def test_predict_against_reference_outputs(golden_dir):
    g = np.load(golden_dir / "predict.npz")
    np.testing.assert_array_equal(ops_np.pair_dot(g["U"], g["I"], g["user"], g["item"]), g["logits"])


def test_lightgcn_against_reference_module(golden_dir):
    g = np.load(golden_dir / "lightgcn.npz")
    nu, ni, nl = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    uc = unflatten(g["user_consumed_flat"])
    rp, ci, va = ops_np.lightgcn_laplacian(nu, ni, uc)
    import scipy.sparse as ssp
    ref = ssp.coo_matrix((g["lap_vals"], (g["lap_rows"], g["lap_cols"])), shape=(nu + ni, nu + ni)).tocsr()
    mine = ssp.csr_matrix((va, ci, rp), shape=(nu + ni, nu + ni))
    assert (abs(ref - mine) > 1e-7).nnz == 0
    E0 = np.concatenate([g["U0"], g["I0"]])
    out = ops_np.lightgcn_propagate(rp, ci, va, E0, nl)
    np.testing.assert_allclose(out[:nu], g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[nu:], g["item_embeds"], rtol=1e-5, atol=1e-6)


def test_bag_pool_matches_tf_documented_semantics():
    """tfops/features.py:90-118 worked by hand: OOV rows are zero, sqrtn divides by sqrt(#non-oov)."""
    table = np.arange(12, dtype=np.float32).reshape(6, 2)
    idx = np.array([[0, 1, 5], [5, 5, 5], [2, 5, 4]], np.int32)
    out = ops_np.bag_pool(table, idx, "sqrtn", oov=5)
    np.testing.assert_allclose(out[0], (table[0] + table[1]) / np.sqrt(2))
    np.testing.assert_array_equal(out[1], [0, 0])
    np.testing.assert_allclose(out[2], (table[2] + table[4]) / np.sqrt(2))
    np.testing.assert_allclose(ops_np.bag_pool(table, idx, "mean", 5)[2], (table[2] + table[4]) / 2)
    np.testing.assert_allclose(ops_np.bag_pool(table, idx, "sum", 5)[0], table[0] + table[1])


def test_din_attention_matches_direct_formula():
    """layers/attention.py:28-64 evaluated literally (concat -> dense -> mask -> softmax)."""
    rng = np.random.default_rng(0)
    B, L, K = 3, 4, 8
    q = rng.standard_normal((B, K)); keys = rng.standard_normal((B, L, K))
    W1 = rng.standard_normal((4 * K, 16)); b1 = rng.standard_normal(16)
    W2 = rng.standard_normal((16, 1)); b2 = rng.standard_normal(1)
    lens = np.array([4, 1, 2])
    out, a = ops_np.din_attention(q, keys, lens, W1, b1, W2, b2)
    for b in range(B):
        s = []
        for l in range(L):
            x = np.concatenate([q[b], keys[b, l], q[b] - keys[b, l], q[b] * keys[b, l]])
            h = 1 / (1 + np.exp(-(x @ W1 + b1)))
            s.append((h @ W2 + b2)[0] / np.sqrt(K) if l < lens[b] else -(2 ** 32) + 1)
        s = np.array(s); e = np.exp(s - s.max()); w = e / e.sum()
        np.testing.assert_allclose(a[b], w, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(out[b], w @ keys[b], rtol=1e-12)


def test_segments_and_adam_oracles():
    idx = np.array([3, 1, 3, 7, -1, 1, 3], np.int32)
    pos, rows, start = ops_np.segments(idx, 5)  # 7 and -1 are dropped
    np.testing.assert_array_equal(rows, [1, 3])
    np.testing.assert_array_equal(start, [0, 2, 5])
    np.testing.assert_array_equal(pos, [1, 5, 0, 2, 6])
    w = np.ones(3, np.float32); z = np.zeros(3, np.float32); g = np.array([1, -2, 0.5], np.float32)
    w1, m1, v1 = ops_np.adam_step(w, z, z, g, lr=0.1, step=1, eps=0.0)
    np.testing.assert_allclose(w1, w - 0.1 * np.sign(g), rtol=1e-5)  # first Adam step = lr*sign(g)


def test_evaluation_metrics_against_reference(golden_dir):
    """evaluation/metrics.py of the reference (sklearn / pandas based) vs the host metrics used by
    `fit(..., eval_data, metrics)` / `evaluate` — listwise metrics per user and averaged, coverage,
    rmse, balanced accuracy, grouped AUC, PR-AUC."""
    from librecommender_amd.evaluation import metrics as M

    g = np.load(golden_dir / "metrics.npz")
    k, n_items = int(g["k"]), int(g["n_items"])
    flat, truths, p = g["truth_flat"], [], 0
    while p < len(flat):
        truths.append(flat[p + 1: p + 1 + flat[p]])
        p += 1 + flat[p]
    recos = g["reco"]
    users = list(range(len(truths)))
    yt, yr = dict(enumerate(truths)), dict(enumerate(recos))
    for name, fn in (("precision", M.precision_at_k), ("recall", M.recall_at_k),
                     ("map", M.average_precision_at_k), ("ndcg", M.ndcg_at_k)):
        per_user = np.asarray([fn(yt[u], yr[u], k) for u in users], dtype=np.float64)
        np.testing.assert_allclose(per_user, g[f"{name}_per_user"], rtol=1e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(M.listwise_mean(fn, yt, yr, users, k), float(g[name]), rtol=1e-6)
    assert M.coverage(yr, users, n_items) == float(g["coverage"])
    y, prob, uidx = g["y_true"], g["y_prob"], g["user_indices"]
    np.testing.assert_allclose(M.rmse(y * 4 + 1, prob * 5), float(g["rmse"]), rtol=1e-12)
    np.testing.assert_allclose(M.roc_gauc(y, prob, uidx), float(g["roc_gauc"]), rtol=1e-12)
    from sklearn.metrics import auc, balanced_accuracy_score, precision_recall_curve
    pr, rc, _ = precision_recall_curve(y, prob)
    np.testing.assert_allclose(auc(rc, pr), float(g["pr_auc"]), rtol=1e-12)
    np.testing.assert_allclose(balanced_accuracy_score(y, np.round(prob)), float(g["balanced_accuracy"]), rtol=1e-12)
    # single-class users contribute 0 (the reference's `_safe_roc_auc` intent, metrics.py:44-49)
    y2 = y.copy(); y2[uidx == 5] = 1.0
    assert np.isfinite(M.roc_gauc(y2, prob, uidx))


def test_independent_restatements_agree():
    """The TF graphs are restated twice, independently: op by op in numpy (oracle/ops_np.py) and as
    whole models in PyTorch (oracle/models_torch.py).  They must agree with each other — attention,
    FM pairwise term, TF1 Adam, BatchNorm — since no TensorFlow is available to pin either."""
    import torch

    from oracle import models_torch as MT

    rng = np.random.default_rng(5)
    # DIN attention: DINOracle._attention vs ops_np.din_attention
    B, L, K = 4, 5, 8
    q, keys = rng.standard_normal((B, K)), rng.standard_normal((B, L, K))
    W1, b1 = rng.standard_normal((4 * K, 16)), rng.standard_normal(16)
    W2, b2 = rng.standard_normal((16, 1)), rng.standard_normal(1)
    lens = np.array([5, 1, 3, 2])
    W = {"user_embeds_var": torch.zeros(2, K), "item_embeds_var": torch.zeros(2, K),
         "attention/attention_layer1/kernel": torch.from_numpy(W1), "attention/attention_layer1/bias": torch.from_numpy(b1),
         "attention/attention_layer2/kernel": torch.from_numpy(W2), "attention/attention_layer2/bias": torch.from_numpy(b2),
         "mlp/mlp_layer1/kernel": torch.zeros(3 * K, 1), "mlp/mlp_layer1/bias": torch.zeros(1),
         "out/kernel": torch.zeros(1, 1), "out/bias": torch.zeros(1)}
    o = MT.DINOracle(W, (1,), use_bn=False, max_seq_len=L, dtype=torch.float64)
    att = o._attention(torch.from_numpy(q), torch.from_numpy(keys), torch.from_numpy(lens)).detach().numpy()
    ref, _ = ops_np.din_attention(q, keys, lens, W1, b1, W2, b2)
    np.testing.assert_allclose(att, ref, rtol=1e-12, atol=1e-14)
    # TF1 Adam: TF1Adam.step vs ops_np.adam_step over three steps with changing gradients (fp32 both)
    w = torch.from_numpy(rng.standard_normal((6, 4)).astype(np.float32)).requires_grad_(True)
    w_np, m_np, v_np = w.detach().numpy().copy(), np.zeros((6, 4), np.float32), np.zeros((6, 4), np.float32)
    opt = MT.TF1Adam(lr=1e-2, eps=1e-5)
    for step in range(1, 4):
        g = rng.standard_normal((6, 4)).astype(np.float32)
        w.grad = torch.from_numpy(g.copy())
        opt.step([w])
        w_np, m_np, v_np = ops_np.adam_step(w_np, m_np, v_np, g, 1e-2, step, eps=1e-5, tf_style=True)
        np.testing.assert_allclose(w.detach().numpy(), w_np, rtol=2e-7, atol=1e-9)
    # BatchNorm (training): tf_batch_norm vs the closed form, and its moving-average update
    x = torch.from_numpy(rng.standard_normal((32, 5)))
    gamma, beta = torch.from_numpy(rng.random(5) + 0.5), torch.from_numpy(rng.standard_normal(5))
    mm, mv = torch.zeros(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64)
    y = MT.tf_batch_norm(x, gamma, beta, mm, mv, training=True).numpy()
    mu, var = x.numpy().mean(0), x.numpy().var(0)
    np.testing.assert_allclose(y, (x.numpy() - mu) / np.sqrt(var + 1e-3) * gamma.numpy() + beta.numpy(), rtol=1e-12)
    np.testing.assert_allclose(mm.numpy(), 0.01 * mu, rtol=1e-12)
    np.testing.assert_allclose(mv.numpy(), 0.99 + 0.01 * var, rtol=1e-12)
    # FM pairwise term inside DeepFMOracle == ops_np.fm_pairwise
    e = rng.standard_normal((3, 6, 4))
    pair, fsum = ops_np.fm_pairwise(e)
    np.testing.assert_allclose(pair, 0.5 * (e.sum(1) ** 2 - (e ** 2).sum(1)), rtol=1e-12)


def test_partition_ranking_equals_lexsort_ranking_on_distinct_scores():
    """The O(N) selection used as the recommend leg's CPU baseline returns the ids of the pinned ranking rule."""
    from oracle import ops_np

    rng = np.random.default_rng(3)
    preds = rng.standard_normal((7, 500)).astype(np.float32)
    consumed = {u: sorted(rng.choice(500, 20, replace=False).tolist()) for u in range(7)}
    ids_a, _ = ops_np.rank_recommendations(list(range(7)), preds, 10, 500, consumed, True)
    ids_b = ops_np.rank_recommendations_partition(list(range(7)), preds, 10, 500, consumed, True)
    np.testing.assert_array_equal(ids_a, ids_b)


def test_softmax_ce_oracle_against_torch_autograd():
    """`ops_np.softmax_ce` (closed-form loss and gradients) against torch's own cross_entropy + autograd on the same
    masked, bias-corrected logits — an independent implementation, since no TensorFlow is here to pin it."""
    import torch

    rng = np.random.default_rng(17)
    B, N, D, pos0 = 7, 19, 6, 3
    X, Y = rng.standard_normal((B, D)), rng.standard_normal((N, D))
    bias = rng.standard_normal(N) * 0.3
    col_ids = rng.integers(0, 9, N)
    row_ids = col_ids[pos0:pos0 + B].copy()                       # the positives' own ids (duplicates guaranteed)
    g = rng.random(B) + 0.5
    loss, dX, dY = ops_np.softmax_ce(X, Y, bias, row_ids, col_ids, pos0, g)
    Xt, Yt = torch.tensor(X, requires_grad=True), torch.tensor(Y, requires_grad=True)
    logits = Xt @ Yt.T + torch.tensor(bias)[None, :]
    lab = torch.arange(B) + pos0
    hit = torch.tensor(row_ids)[:, None] == torch.tensor(col_ids)[None, :]
    hit[torch.arange(B), lab] = False
    logits = torch.where(hit, torch.tensor(float(np.finfo(np.float32).min), dtype=torch.float64), logits)
    lt = torch.nn.functional.cross_entropy(logits, lab, reduction="none")
    (lt * torch.tensor(g)).sum().backward()
    np.testing.assert_allclose(loss, lt.detach().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dX, Xt.grad.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(dY, Yt.grad.numpy(), rtol=1e-10, atol=1e-13)
    assert hit.any()                                              # the accidental-hit branch was exercised


def test_deepfm_first_layer_oracles_against_torch_autograd():
    """The fused lookup + first Dense layer restatements (`deepfm_l1_fwd / _wgrad / _dgrad`, `fm_rows_gradient`)
    against autograd of the plain expression: z1 = concat_f(table[idx]) @ Wp + b,  pair = FM term,  lin = lin[idx]."""
    import torch

    rng = np.random.default_rng(23)
    B, F, K, H, V = 6, 4, 8, 5, 11
    frs = np.array([0, 3, 6, 8, 11])
    idx = np.stack([rng.integers(frs[f], frs[f + 1], B) for f in range(F)], axis=1)
    idx[0, 1] = idx[1, 1]                                         # a repeated row inside one field
    table, lin = rng.standard_normal((V, K)), rng.standard_normal(V)
    Wp, bias = rng.standard_normal((F * K, H)), rng.standard_normal(H)
    wp = rng.standard_normal(K)                                   # output-layer weights of the FM pairwise term
    gz, gl = rng.standard_normal((B, H)), rng.standard_normal(B)

    z1, pair, fsum, lin_out = ops_np.deepfm_l1_fwd(table, lin, idx, Wp, bias)
    T = torch.tensor(table, requires_grad=True)
    Wt = torch.tensor(Wp, requires_grad=True)
    e = T[torch.tensor(idx)]                                      # [B,F,K]
    zt = e.reshape(B, F * K) @ Wt + torch.tensor(bias)
    pt = 0.5 * (e.sum(1) ** 2 - (e ** 2).sum(1))
    np.testing.assert_allclose(z1, zt.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(pair, pt.detach().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lin_out, lin[idx], rtol=0)
    # upstream: dL/dz1 = gz, dL/dpair[b,:] = gl[b] * wp  (logit += pair @ wp)
    ((zt * torch.tensor(gz)).sum() + (pt * (torch.tensor(gl)[:, None] * torch.tensor(wp)[None, :])).sum()).backward()
    np.testing.assert_allclose(ops_np.deepfm_l1_wgrad(table, idx, gz), Wt.grad.numpy(), rtol=1e-11, atol=1e-13)

    order, rows, start, slotT = ops_np.segments_fields(idx, frs)
    ge = ops_np.deepfm_l1_dgrad(gz, Wp, K, gl, wp, fsum, slotT)   # [B*F,K] in slot order, WITHOUT the -e*... term
    out = ops_np.fm_rows_gradient(table, ge, order, rows, start, F, gl, wp, None, None)
    grow = out[0] if isinstance(out, tuple) else out
    np.testing.assert_allclose(grow, T.grad.numpy()[rows], rtol=1e-10, atol=1e-12)


def test_tf1_adam_restatement_against_torch_adam_at_zero_epsilon():
    """tf.train.AdamOptimizer puts epsilon outside the bias correction ("epsilon hat"), torch.optim.Adam inside; with
    epsilon = 0 the two updates are the same function of the gradient history.  This pins the moment / bias-correction
    algebra of both restatements (`TF1Adam`, `ops_np.adam_step(tf_style=True)`) to an independent implementation."""
    import torch

    from oracle.models_torch import TF1Adam

    rng = np.random.default_rng(31)
    w0 = rng.standard_normal((5, 3))
    a = torch.tensor(w0, requires_grad=True)
    b = torch.tensor(w0.copy(), requires_grad=True)
    w_np, m_np, v_np = w0.copy(), np.zeros_like(w0), np.zeros_like(w0)
    mine, ref = TF1Adam(lr=3e-3, eps=0.0), torch.optim.Adam([b], lr=3e-3, eps=0.0)
    for step in range(1, 7):
        g = rng.standard_normal((5, 3)) + 0.1                     # no exact zeros: 0 / 0 would appear at eps = 0
        a.grad, b.grad = torch.tensor(g), torch.tensor(g.copy())
        mine.step([a])
        ref.step()
        w_np, m_np, v_np = ops_np.adam_step(w_np, m_np, v_np, g, 3e-3, step, eps=0.0, tf_style=True)
        np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(w_np, b.detach().numpy(), rtol=1e-12, atol=1e-14)


def test_lightgcn_oracle_against_reference_step(golden_dir):
    """`LightGCNOracle` (the CPU baseline of `bench.py --workload lightgcn`) reproduces the reference module's own
    loss, table gradients and first Adam step (fixture generated by oracle/make_golden.py from
    libreco/algorithms/torch_modules/lightgcn_module.py)."""
    import torch
    from oracle.models_torch import LightGCNOracle

    g = np.load(golden_dir / "lightgcn.npz")
    nu, ni, nl = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    uc = unflatten(g["user_consumed_flat"])
    eu = np.concatenate([np.full(len(v), u) for u, v in uc.items()])
    ei = np.concatenate([np.asarray(v) for v in uc.values()])
    o = LightGCNOracle(nu, ni, 16, nl, eu, ei, lr=1e-2, epsilon=1e-8, U0=g["U0"], I0=g["I0"])
    ue, ie = o.propagate()
    np.testing.assert_allclose(ue.detach().numpy(), g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.detach().numpy(), g["item_embeds"], rtol=1e-5, atol=1e-6)
    U, I = o.U, o.I
    ue, ie = o.propagate()
    u, p, n = (torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))
    loss = -torch.nn.functional.logsigmoid((ue[u] * ie[p]).sum(1) - (ue[u] * ie[n]).sum(1)).mean()
    gU, gI = torch.autograd.grad(loss, [U, I])
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(gU.numpy(), g["gU"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(gI.numpy(), g["gI"], rtol=1e-4, atol=1e-7)
    o.train_step(g["users"], g["pos"], g["neg"])
    np.testing.assert_allclose(o.U.detach().numpy(), g["U1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o.I.detach().numpy(), g["I1"], rtol=1e-5, atol=1e-6)
