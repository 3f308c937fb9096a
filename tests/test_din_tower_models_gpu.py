"""DIN and TwoTower graphs on the HIP path vs the PyTorch-CPU oracle restatement of the
reference's TF graphs (`oracle/models_torch.py`), from identical weights (`-m gpu`).

Parity contract as for FM/DeepFM: forward logits / tower embeddings 1e-5 abs against the fp64
oracle; losses 1e-5; with dense_adam=True the TF1 dense-Adam weight trajectory over several steps
(rtol 1e-4, atol 3e-6 on weights that moved by ~lr per step); with the default row-wise Adam the
first step (zero moments) is identical on touched rows and untouched rows stay frozen."""
import numpy as np
import pytest
import torch

from librecommender_amd.nets import FeatDINNet, FeatSpec, TwoTowerNet
from oracle.models_torch import DINOracle, TwoTowerOracle, export_net_weights

pytestmark = pytest.mark.gpu


def T(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    return t.long() if dtype is None and t.dtype in (torch.int32, torch.int64) else t


def close(got, ref, name, rtol=1e-4, atol=3e-6):
    np.testing.assert_allclose(got.numpy().reshape(ref.shape), ref.detach().numpy(), rtol=rtol, atol=atol, err_msg=name)


# ----------------------------------------------------------------------------------------------
# DIN
# ----------------------------------------------------------------------------------------------
def din_batch(rng, B, nu, ni, L, n_sp, vocab, n_dense):
    users = rng.integers(0, nu, B)
    items = rng.integers(0, ni, B)
    lens = rng.integers(1, L + 1, B)
    seqs = np.full((B, L), ni, dtype=np.int64)                    # pad id = n_items (sequence.py:56-58)
    for b in range(B):
        seqs[b, : lens[b]] = rng.integers(0, ni, lens[b])
    lens[0] = 1; seqs[0] = ni                                     # empty history: one attended pad key
    sparse = (rng.integers(0, vocab, (B, n_sp)) + np.arange(n_sp) * (vocab + 1)) if n_sp else None
    dense = rng.standard_normal((B, n_dense)).astype(np.float32) if n_dense else None
    labels = rng.integers(0, 2, B).astype(np.float32)
    return users, items, sparse, dense, seqs, lens, labels


def din_pair(dev, K, hidden, L, n_sp=0, vocab=7, n_dense=0, item_side=False, dense_adam=True, lr=1e-2, seed=0,
             use_tf_attention=False):
    rng = np.random.default_rng(seed)
    nu, ni = 40, 60
    spec = FeatSpec(nu, ni, n_sp, n_sp * (vocab + 1), n_dense)
    kw = {}
    okw = {}
    if item_side:   # last sparse column and last dense column are item features
        isu = rng.integers(0, vocab, (ni + 1, 1)) + (n_sp - 1) * (vocab + 1)
        idu = rng.standard_normal((ni + 1, 1)).astype(np.float32)
        kw = dict(item_sparse_unique=isu, item_dense_unique=idu, item_dense_cols=[n_dense - 1])
        okw = dict(kw)
    net = FeatDINNet(spec, K, hidden, use_bn=True, max_seq_len=L, lr=lr, device=dev, dense_adam=dense_adam,
                     use_tf_attention=use_tf_attention, **kw)
    W = export_net_weights(net)
    o = DINOracle(W, hidden, True, L, lr=lr, dtype=torch.float64, use_tf_attention=use_tf_attention, **okw)
    return rng, net, o, W, (nu, ni, L, n_sp, vocab, n_dense)


def din_call(b):
    users, items, sparse, dense, seqs, lens, labels = b
    return dict(users=users, items=items, sparse=sparse, dense=dense, seqs=seqs, seq_lens=lens)


def din_oracle_args(b):
    users, items, sparse, dense, seqs, lens, labels = b
    return (T(users), T(items), None if sparse is None else T(sparse), None if dense is None else T(dense),
            T(seqs), T(lens))


@pytest.mark.parametrize("K,item_side,n_sp,n_dense", [(16, False, 0, 0), (32, False, 3, 2), (16, True, 3, 2)])
def test_din_forward_and_tf_dense_adam_trajectory(dev, K, item_side, n_sp, n_dense):
    rng, net, o, W, shp = din_pair(dev, K, (32, 16), 6, n_sp, 7, n_dense, item_side)
    assert net.fused == (not item_side)                            # pure-id keys use the fused kernels
    batches = [din_batch(rng, 48, *shp) for _ in range(3)]
    lg = net.forward(**din_call(batches[0])).cpu().numpy()
    np.testing.assert_allclose(lg, o.forward(*din_oracle_args(batches[0])).detach().numpy(), rtol=1e-5, atol=1e-5)
    for b in batches:
        l_hip = float(net.train_step(labels=b[-1], **din_call(b)))
        l_ref = float(o.train_step(*din_oracle_args(b), T(b[-1])))
        assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    for name, ref in o.V.v.items():
        close(W2[name], ref, name)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)


@pytest.mark.parametrize("K,n_sp,n_dense", [(16, 0, 0), (32, 3, 2)])
def test_youtube_ranking_forward_and_tf_dense_adam_trajectory(dev, K, n_sp, n_dense):
    """algorithms/youtube_ranking.py: forward logits and three TF1-dense-Adam steps (every variable incl. the item
    table through the pooled field) against the fp64 restatement; the batches hold an empty history (all pads)."""
    from librecommender_amd.nets import FeatYouTubeRankingNet
    from oracle.models_torch import YouTubeRankingOracle

    rng = np.random.default_rng(4)
    nu, ni, L, vocab = 40, 60, 6, 7
    spec = FeatSpec(nu, ni, n_sp, n_sp * (vocab + 1), n_dense)
    net = FeatYouTubeRankingNet(spec, K, (32, 16), use_bn=True, max_seq_len=L, lr=1e-2, device=dev, dense_adam=True)
    o = YouTubeRankingOracle(export_net_weights(net), ni, (32, 16), True, lr=1e-2, dtype=torch.float64)
    shp = (nu, ni, L, n_sp, vocab, n_dense)
    batches = [din_batch(rng, 48, *shp) for _ in range(3)]
    lg = net.forward(**din_call(batches[0])).cpu().numpy()
    np.testing.assert_allclose(lg, o.forward(*din_oracle_args(batches[0])).detach().numpy(), rtol=1e-5, atol=1e-5)
    for b in batches:
        l_hip = float(net.train_step(labels=b[-1], **din_call(b)))
        l_ref = float(o.train_step(*din_oracle_args(b), T(b[-1])))
        assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    for name, ref in o.V.v.items():
        close(W2[name], ref, name)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)


@pytest.mark.parametrize("K,item_side,mode,heads,layers,pos,causal", [
    (16, False, "concat", 1, 1, "trainable", False), (16, True, "concat", 2, 2, "sinusoidal", True),
    (16, True, "elementwise", 4, 1, "trainable", False)])
def test_transformer_forward_and_tf_dense_adam_trajectory(dev, K, item_side, mode, heads, layers, pos, causal):
    """algorithms/transformer.py: forward logits and three TF1-dense-Adam steps of every variable (tables, positional
    encoding, attention / FFN kernels, norms, MLP) against the fp64 restatement."""
    from librecommender_amd.nets.seq_nets import FeatTransformerNet, sinusoidal_encoding
    from oracle.models_torch import TransformerOracle

    rng = np.random.default_rng(8)
    nu, ni, L, vocab = 40, 60, 6, 7
    n_sp, n_dense = (3, 2) if item_side else (2, 1)
    spec = FeatSpec(nu, ni, n_sp, n_sp * (vocab + 1), n_dense)
    kw = {}
    if item_side:   # last sparse column and last dense column are item features
        isu = rng.integers(0, vocab, (ni + 1, 1)) + (n_sp - 1) * (vocab + 1)
        idu = rng.standard_normal((ni + 1, 1)).astype(np.float32)
        kw = dict(item_sparse_unique=isu, item_dense_unique=idu, item_dense_cols=[n_dense - 1])
    net = FeatTransformerNet(spec, K, (32, 16), use_bn=True, max_seq_len=L, num_heads=heads, num_tfm_layers=layers,
                             positional_embedding=pos, use_causal_mask=causal, feat_agg_mode=mode, lr=1e-2, device=dev,
                             dense_adam=True, **kw)
    o = TransformerOracle(export_net_weights(net), (32, 16), True, L, heads, layers, pos == "trainable",
                          sinusoidal_encoding(L, K), causal, mode, lr=1e-2, dtype=torch.float64, **kw)
    shp = (nu, ni, L, n_sp, vocab, n_dense)
    batches = [din_batch(rng, 48, *shp) for _ in range(3)]
    lg = net.forward(**din_call(batches[0])).cpu().numpy()
    np.testing.assert_allclose(lg, o.forward(*din_oracle_args(batches[0])).detach().numpy(), rtol=1e-5, atol=1e-5)
    for b in batches:
        l_hip = float(net.train_step(labels=b[-1], **din_call(b)))
        l_ref = float(o.train_step(*din_oracle_args(b), T(b[-1])))
        assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    for name, ref in o.V.v.items():
        close(W2[name], ref, name)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)


@pytest.mark.parametrize("item_side,heads,alpha,beta", [(False, 2, 1.0, 1.0), (True, 4, 0.3, 0.8)])
def test_sim_forward_and_tf_dense_adam_trajectory(dev, item_side, heads, alpha, beta):
    """algorithms/sim.py: second-stage (inference) logits and three TF1-dense-Adam steps of every variable (the loss
    mixes both stages) against the fp64 restatement; windows with fewer valid items than `search_topk` included."""
    from librecommender_amd.nets.seq_nets import FeatSIMNet
    from oracle.models_torch import SIMOracle

    rng = np.random.default_rng(13)
    nu, ni, Lg, S, K, vocab, topk = 40, 60, 9, 4, 16, 7, 5
    n_sp, n_dense = (3, 2) if item_side else (2, 1)
    spec = FeatSpec(nu, ni, n_sp, n_sp * (vocab + 1), n_dense)
    kw = {}
    if item_side:
        isu = rng.integers(0, vocab, (ni + 1, 1)) + (n_sp - 1) * (vocab + 1)
        idu = rng.standard_normal((ni + 1, 1)).astype(np.float32)
        kw = dict(item_sparse_unique=isu, item_dense_unique=idu, item_dense_cols=[n_dense - 1])
    net = FeatSIMNet(spec, K, (32, 16), use_bn=True, alpha=alpha, beta=beta, search_topk=topk, long_max_len=Lg,
                     short_max_len=S, num_heads=heads, lr=1e-2, device=dev, dense_adam=True, **kw)
    o = SIMOracle(export_net_weights(net), (32, 16), True, alpha, beta, topk, Lg, S, heads, lr=1e-2,
                  dtype=torch.float64, **kw)

    def sim_batch(B=48):
        users, items = rng.integers(0, nu, B), rng.integers(0, ni, B)
        seqs = np.full((B, Lg + S), ni, dtype=np.int64)
        lens = np.ones((B, 2), dtype=np.int64)
        for b in range(B):
            nl, ns = rng.integers(0, Lg + 1), rng.integers(0, S + 1)
            seqs[b, :nl] = rng.permutation(ni)[:nl]               # distinct items: no ties in the top-k search
            seqs[b, Lg:Lg + ns] = rng.integers(0, ni, ns)
            lens[b] = max(nl, 1), max(ns, 1)
        sparse = rng.integers(0, vocab, (B, n_sp)) + np.arange(n_sp) * (vocab + 1)
        dense = rng.standard_normal((B, n_dense)).astype(np.float32)
        return users, items, sparse, dense, seqs, lens, rng.integers(0, 2, B).astype(np.float32)

    batches = [sim_batch() for _ in range(3)]
    lg = net.forward(**din_call(batches[0])).cpu().numpy()
    np.testing.assert_allclose(lg, o.forward(*din_oracle_args(batches[0])).detach().numpy(), rtol=1e-5, atol=1e-5)
    for b in batches:
        l_hip = float(net.train_step(labels=b[-1], **din_call(b)))
        l_ref = float(o.train_step(*din_oracle_args(b), T(b[-1])))
        assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    for name, ref in o.V.v.items():
        close(W2[name], ref, name)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)


def test_din_tf_attention_variant(dev):
    """`use_tf_attention=True`: keras dot-product attention (layers/attention.py:5-25) instead of the
    DIN attention MLP."""
    rng, net, o, W, shp = din_pair(dev, 16, (32, 16), 6, 2, 7, 1, False, use_tf_attention=True)
    assert not net.fused
    for _ in range(2):
        b = din_batch(rng, 40, *shp)
        l_hip = float(net.train_step(labels=b[-1], **din_call(b)))
        l_ref = float(o.train_step(*din_oracle_args(b), T(b[-1])))
        assert abs(l_hip - l_ref) < 1e-5
    W2 = export_net_weights(net)
    for name in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var", "mlp/mlp_layer1/kernel", "out/kernel"):
        close(W2[name], o.V.v[name], name)
    b = din_batch(rng, 40, *shp)
    np.testing.assert_allclose(net.forward(**din_call(b)).cpu().numpy(),
                               o.forward(*din_oracle_args(b)).detach().numpy(), rtol=1e-4, atol=1e-5)


def test_din_lazy_adam_first_step(dev):
    rng, net, o, W, shp = din_pair(dev, 16, (32, 16), 5, 2, 9, 0, False, dense_adam=False, lr=1e-3, seed=3)
    b = din_batch(rng, 64, *shp)
    net.train_step(labels=b[-1], **din_call(b))
    o.train_step(*din_oracle_args(b), T(b[-1]))
    W2 = export_net_weights(net)
    users, items, sparse, _, seqs, lens, _ = b
    valid = np.arange(seqs.shape[1])[None, :] < lens[:, None]
    touched = {"user_embeds_var": np.unique(users), "item_embeds_var": np.unique(np.concatenate([items, seqs[valid]])),
               "sparse_embeds_var": np.unique(sparse)}
    for name, rows in touched.items():
        np.testing.assert_allclose(W2[name].numpy()[rows], o.V.v[name].detach().numpy()[rows], rtol=1e-4, atol=2e-6, err_msg=name)
        rest = np.setdiff1d(np.arange(W[name].shape[0]), rows)
        np.testing.assert_array_equal(W2[name].numpy()[rest], W[name].numpy()[rest])
    for name in ("attention/attention_layer1/kernel", "attention/attention_layer2/bias", "mlp/mlp_layer1/kernel", "out/kernel"):
        close(W2[name], o.V.v[name], name, atol=2e-6)


# ----------------------------------------------------------------------------------------------
# TwoTower
# ----------------------------------------------------------------------------------------------
def tower_pair(dev, K, hidden, n_us, n_is, n_ud, n_id, vocab=6, lr=1e-2, seed=0, **kw):
    rng = np.random.default_rng(seed)
    nu, ni = 50, 45
    n_sp = n_us + n_is
    ud_cols, id_cols = list(range(n_ud)), list(range(n_ud, n_ud + n_id))
    net = TwoTowerNet(nu, ni, n_sp * (vocab + 1), n_us, n_is, ud_cols, id_cols, n_ud + n_id, embed_size=K,
                      hidden_units=hidden, lr=lr, device=dev, **kw)
    W = export_net_weights(net)
    okw = {k: v for k, v in kw.items() if k in ("use_bn", "norm_embed", "margin", "temperature", "use_correction",
                                                "remove_accidental_hits")}
    o = TwoTowerOracle(W, hidden, user_dense_cols=ud_cols, item_dense_cols=id_cols, lr=lr, dtype=torch.float64, **okw)
    return rng, net, o, W, (nu, ni, n_us, n_is, n_ud, n_id, vocab)


def tower_batch(rng, B, nu, ni, n_us, n_is, n_ud, n_id, vocab, neg=False):
    d = dict(users=rng.integers(0, nu, B), items=rng.integers(0, ni, B))
    if n_us:
        d["user_sparse"] = rng.integers(0, vocab, (B, n_us)) + np.arange(n_us) * (vocab + 1)
    if n_is:
        d["item_sparse"] = rng.integers(0, vocab, (B, n_is)) + (n_us + np.arange(n_is)) * (vocab + 1)
    if n_ud:
        d["user_dense"] = rng.standard_normal((B, n_ud)).astype(np.float32)
    if n_id:
        d["item_dense"] = rng.standard_normal((B, n_id)).astype(np.float32)
    if neg:
        d["items_neg"] = rng.integers(0, ni, B)
        if n_is:
            d["item_sparse_neg"] = rng.integers(0, vocab, (B, n_is)) + (n_us + np.arange(n_is)) * (vocab + 1)
        if n_id:
            d["item_dense_neg"] = rng.standard_normal((B, n_id)).astype(np.float32)
    return d


def to_oracle(d):
    return {k: T(v) for k, v in d.items()}


@pytest.mark.parametrize("loss_type,feat,kw", [
    ("cross_entropy", (0, 0, 0, 0), {}),                                           # BASELINE config 1: pure ids
    ("cross_entropy", (2, 1, 1, 1), {"norm_embed": True}),
    ("max_margin", (1, 2, 0, 1), {"margin": 0.7}),
    ("softmax", (1, 1, 1, 0), {"temperature": 0.5, "remove_accidental_hits": True}),
    ("softmax", (0, 0, 0, 0), {"temperature": 0.0, "norm_embed": True}),          # learned temperature
])
def test_two_tower_losses_and_trajectory(dev, loss_type, feat, kw):
    rng, net, o, W, shp = tower_pair(dev, 16, (32, 16), *feat, dense_adam=True, **kw)
    nu, ni = shp[0], shp[1]
    counts = rng.integers(1, 50, ni)
    corr = (counts / counts.sum()).astype(np.float32)
    for step in range(3):
        b = tower_batch(rng, 40, *shp, neg=loss_type == "max_margin")
        extra = {}
        if loss_type == "cross_entropy":
            extra["labels"] = rng.integers(0, 2, 40).astype(np.float32)
        if loss_type == "softmax":
            extra["corrections"] = corr[b["items"]]                # batch/tf_feed_dicts.py:121-122
        l_hip = float(net.train_step(loss_type, **b, **extra))
        l_ref = float(o.train_step(loss_type, **to_oracle(b), **to_oracle(extra)))
        assert abs(l_hip - l_ref) < 2e-5, (step, l_hip, l_ref)
    W2 = export_net_weights(net)
    # Adam normalises the gradient: an element whose gradient is ~1e-7 (softmax tails, masked
    # logits) turns an fp32 rounding difference into a visible fraction of one lr-sized step.
    # 5e-5 = 0.5 % of one step at lr = 1e-2.
    for name, ref in o.V.v.items():
        # (the item tower's last bias and the BatchNorm beta in front of it have NO gradient under the in-batch softmax — the
        # columns of d loss / d Y sum to zero — so their fp64 trajectory stays at 1e-14 while any fp32 evaluation feeds Adam pure
        # rounding noise, which it normalises towards +-lr per step whatever the arithmetic: those parameters are bounded by
        # 1.5 % of one step instead of 0.5 %)
        dead = loss_type == "softmax" and float(ref.detach().abs().max()) < 1e-9
        close(W2[name], ref, name, atol=1.5e-4 if dead else 5e-5)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)
    b = tower_batch(rng, 30, *shp)
    ue = net.embed_users(b["users"], b.get("user_sparse"), b.get("user_dense")).cpu()
    ie = net.embed_items(b["items"], b.get("item_sparse"), b.get("item_dense")).cpu()
    ob = to_oracle(b)
    oue = o.user_embeds(ob["users"], ob.get("user_sparse"), ob.get("user_dense")).detach()
    oie = o.item_embeds(ob["items"], ob.get("item_sparse"), ob.get("item_dense")).detach()
    np.testing.assert_allclose(ue.numpy(), oue.numpy(), rtol=1e-3, atol=1e-4)   # downstream of the weights above
    if loss_type == "softmax" and not kw.get("norm_embed"):
        # the item tower's output bias is not identified by the in-batch softmax (a constant added to every item embedding moves
        # every logit of a row alike): its gradient-free random walk (see above) shifts every item embedding by the same vector
        last = f"item_tower/item_tower_layer{len((32, 16))}/bias"
        ie = ie - (W2[last] - o.V.v[last].detach().float())
    np.testing.assert_allclose(ie.numpy(), oie.numpy(), rtol=1e-3, atol=1e-4)


def test_two_tower_first_step_tables(dev):
    """One step from zero moments: row-wise Adam == TF1 dense Adam on touched rows; others frozen."""
    rng, net, o, W, shp = tower_pair(dev, 16, (32, 16), 2, 1, 1, 1, lr=1e-3, seed=5)
    b = tower_batch(rng, 64, *shp)
    labels = rng.integers(0, 2, 64).astype(np.float32)
    fwd_u = net.embed_users(b["users"], b["user_sparse"], b["user_dense"]).cpu().numpy()
    ob = to_oracle(b)
    np.testing.assert_allclose(fwd_u, o.user_embeds(ob["users"], ob["user_sparse"], ob["user_dense"]).detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    net.train_step("cross_entropy", labels=labels, **b)
    o.train_step("cross_entropy", labels=T(labels), **ob)
    W2 = export_net_weights(net)
    touched = {"user_embeds_var": np.unique(b["users"]), "item_embeds_var": np.unique(b["items"]),
               "sparse_embeds_var": np.unique(np.concatenate([b["user_sparse"].ravel(), b["item_sparse"].ravel()]))}
    for name, rows in touched.items():
        np.testing.assert_allclose(W2[name].numpy()[rows], o.V.v[name].detach().numpy()[rows], rtol=1e-4, atol=2e-6, err_msg=name)
        rest = np.setdiff1d(np.arange(W[name].shape[0]), rows)
        np.testing.assert_array_equal(W2[name].numpy()[rest], W[name].numpy()[rest])
    for name in ("user_tower/user_tower_layer1/kernel", "item_tower/bn_in/gamma", "embedding/dense_embeds_var"):
        close(W2[name], o.V.v[name], name, atol=2e-6)


@pytest.mark.parametrize("feat,kw", [((1, 2, 0, 1), {"temperature": 0.7}), ((0, 3, 0, 0), {"temperature": 0.0, "norm_embed": True})])
def test_two_tower_ssl_term(dev, feat, kw):
    """`ssl_pattern` (two_tower.py:295-353, tfops/loss.py:38-47): two masked views of
    [item id | item sparse features] through the item tower, in-batch softmax between them, added with
    weight alpha — the zero "default" row of the ssl table is an out-of-range id on the HIP path."""
    rng, net, o, W, shp = tower_pair(dev, 16, (32, 16), *feat, dense_adam=True, **kw)
    nu, ni, n_us, n_is, n_ud, n_id, vocab = shp
    counts = rng.integers(1, 50, ni)
    corr = (counts / counts.sum()).astype(np.float32)
    for step in range(3):
        b = tower_batch(rng, 32, *shp)
        sit = rng.integers(0, ni, 32)
        sfe = rng.integers(0, vocab, (32, n_is)) + (n_us + np.arange(n_is)) * (vocab + 1)
        idx = np.hstack([sit[:, None] + 1, sfe + ni + 1])
        left, right = idx.copy(), idx.copy()
        perm = rng.permutation(idx.shape[1])
        left[:, perm[: idx.shape[1] // 2]] = 0
        right[:, perm[idx.shape[1] // 2:]] = 0
        extra = {"corrections": corr[b["items"]], "ssl_left": left, "ssl_right": right, "alpha": 0.3}
        if n_id:
            extra["ssl_dense"] = rng.standard_normal((32, n_id)).astype(np.float32)
        l_hip = float(net.train_step("softmax", **b, **extra))
        l_ref = float(o.train_step("softmax", **to_oracle(b), **{k: (T(v) if not np.isscalar(v) else v) for k, v in extra.items()}))
        assert abs(l_hip - l_ref) < 2e-5 * max(1.0, abs(l_ref)), (step, l_hip, l_ref)   # fp32 relative
    W2 = export_net_weights(net)
    for name, ref in o.V.v.items():
        close(W2[name], ref, name, atol=5e-5)
    for name, ref in o.V.buffers.items():
        close(W2[name], ref, name, atol=1e-6)
