"""numpy restatement of the reference's hot-path arithmetic — TEST INFRASTRUCTURE ONLY.

Paths are relative to the reference checkout (massquantity/LibRecommender v1.5.2).  Where the
arithmetic lives in TensorFlow (un-vendored, ``tensorflow>=1.15,<2.16`` per requirements.txt:5)
the function restates TF's documented op semantics at the cited call site: PARITY UNPINNED for
those (no reference test pins them); everything numpy/torch-based is pinned by fixtures made
with the reference itself (oracle/make_golden.py).
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------
# (a1) embedding_lookup — layers/embedding.py:4-23 -> tf.nn.embedding_lookup.  [UNPINNED: TF]
# Out-of-range ids: TF-GPU returns zeros (TF-CPU raises); the HIP path follows TF-GPU.
# ----------------------------------------------------------------------------------------
def embedding_lookup(table: np.ndarray, idx: np.ndarray) -> np.ndarray:
    table = np.asarray(table)
    idx = np.asarray(idx)
    t2 = table.reshape(table.shape[0], -1)
    ok = (idx >= 0) & (idx < t2.shape[0])
    out = t2[np.where(ok, idx, 0)]
    out = np.where(ok[..., None], out, 0).astype(table.dtype)
    return out


# ----------------------------------------------------------------------------------------
# (a2) multi_sparse_alone — tfops/features.py:90-118; seq_embeds_pooling —
# layers/embedding.py:54-85.  OOV row is zeroed, rows summed, divided by count / sqrt(count) of
# non-OOV entries with tf.div_no_nan.  [UNPINNED: TF]
# ----------------------------------------------------------------------------------------
def bag_pool(table: np.ndarray, idx: np.ndarray, combiner: str, oov: int) -> np.ndarray:
    V = table.shape[0]
    live = (idx != oov) & (idx >= 0) & (idx < V)
    rows = table[np.where(live, idx, 0)] * live[..., None]
    res = rows.sum(axis=1, dtype=table.dtype)
    if combiner in ("mean", "sqrtn"):
        cnt = live.sum(axis=1).astype(table.dtype)
        d = np.sqrt(cnt) if combiner == "sqrtn" else cnt
        with np.errstate(divide="ignore", invalid="ignore"):
            res = np.where(d[:, None] > 0, res / d[:, None], 0).astype(table.dtype)
    return res


def bag_pool_bwd(gout: np.ndarray, idx: np.ndarray, V: int, combiner: str, oov: int) -> np.ndarray:
    """d(out)->per-entry gradient [nbags*bag_len, K] (autodiff of bag_pool)."""
    live = (idx != oov) & (idx >= 0) & (idx < V)
    cnt = live.sum(axis=1).astype(gout.dtype)
    if combiner == "sum":
        scale = np.ones_like(cnt)
    else:
        d = np.sqrt(cnt) if combiner == "sqrtn" else cnt
        with np.errstate(divide="ignore"):
            scale = np.where(d > 0, 1.0 / d, 0).astype(gout.dtype)
    g = gout[:, None, :] * scale[:, None, None] * live[..., None]
    return g.reshape(-1, gout.shape[1]).astype(gout.dtype)


# ----------------------------------------------------------------------------------------
# segments: what TF's IndexedSlices path does before Adam (training/tf_trainer.py:120-121 ->
# unique + unsorted_segment_sum).  Integer work, bit-exact target.
# ----------------------------------------------------------------------------------------
def segments(idx: np.ndarray, V: int):
    idx = np.asarray(idx).reshape(-1)
    ok = (idx >= 0) & (idx < V)
    order = np.argsort(np.where(ok, idx, V), kind="stable")
    order = order[: int(ok.sum())]
    sorted_idx = idx[order]
    rows, start = np.unique(sorted_idx, return_index=True)
    start = np.append(start, len(sorted_idx)).astype(np.int32)
    return order.astype(np.int32), rows.astype(np.int32), start


def segment_sum(grad: np.ndarray, pos: np.ndarray, start: np.ndarray) -> np.ndarray:
    g2 = grad.reshape(-1, grad.shape[-1])
    out = np.zeros((len(start) - 1, g2.shape[1]), dtype=grad.dtype)
    for s in range(len(start) - 1):
        acc = np.zeros(g2.shape[1], dtype=grad.dtype)
        for p in pos[start[s]:start[s + 1]]:  # ascending position order, like the kernel
            acc = acc + g2[p]
        out[s] = acc
    return out


def scatter_add_dense(V: int, idx: np.ndarray, grad: np.ndarray) -> np.ndarray:
    """Dense gradient of embedding_lookup: np.add.at (torch nn.Embedding semantics)."""
    g2 = grad.reshape(-1, grad.shape[-1])
    out = np.zeros((V, g2.shape[1]), dtype=np.float64)
    ok = (idx.reshape(-1) >= 0) & (idx.reshape(-1) < V)
    np.add.at(out, idx.reshape(-1)[ok], g2[ok].astype(np.float64))
    return out


# ----------------------------------------------------------------------------------------
# (a11) Adam.  TF1: tf.train.AdamOptimizer(lr, epsilon) at training/tf_trainer.py:120 —
#   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
#   w -= lr_t * m / (sqrt(v) + eps)          (dense over every row)          [UNPINNED: TF]
# torch: torch.optim.Adam at training/torch_trainer.py:63-69 —
#   g += wd*w; m,v as above; w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)   [pinned by
#   tests/golden via torch.optim.Adam itself]
# ----------------------------------------------------------------------------------------
def adam_step(w, m, v, g, lr, step, beta1=0.9, beta2=0.999, eps=1e-5, weight_decay=0.0,
              tf_style=True):
    dt = w.dtype
    f = dt.type
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if tf_style:
        # tf.train.AdamOptimizer._apply_sparse_shared: (1 - beta) is formed in the variable's
        # dtype: float32(1) - float32(0.999) = 0.00100004673 (not 0.001)
        omb1, omb2 = f(1) - f(beta1), f(1) - f(beta2)
        m = m * f(beta1) + g * omb1
        v = v * f(beta2) + (g * g) * omb2
        lr_t = f(lr * np.sqrt(bc2) / bc1)
        w = w - lr_t * (m / (np.sqrt(v) + f(eps)))
    else:
        # torch.optim.adam._single_tensor_adam: Python-double (1 - beta) cast to the dtype
        g = g + f(weight_decay) * w
        m = m + (g - m) * f(1.0 - beta1)                 # exp_avg.lerp_(grad, 1 - beta1)
        v = v * f(beta2) + (f(1.0 - beta2) * g) * g      # mul_(beta2).addcmul_(g, g, 1 - beta2)
        denom = np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)
        w = w - f(lr / bc1) * (m / denom)
    return w.astype(dt), m.astype(dt), v.astype(dt)


# ----------------------------------------------------------------------------------------
# Field-partitioned segments (lr_segments_build_fields): `segments` restricted to ids that lie in
# their own field's row range, plus the inverse map slotT[f, b].  Integer work, bit-exact target.
# ----------------------------------------------------------------------------------------
def segments_fields(idx: np.ndarray, field_row_start: np.ndarray):
    idx = np.asarray(idx)
    B, F = idx.shape
    frs = np.asarray(field_row_start, dtype=np.int64)
    ok = (idx >= frs[None, :-1]) & (idx < frs[None, 1:])
    masked = np.where(ok, idx, -1)
    order, rows, start = segments(masked, int(frs[-1]))
    slot = np.full(B * F, -1, dtype=np.int32)
    slot[order] = np.arange(len(order), dtype=np.int32)
    return order, rows, start, slot.reshape(B, F).T.copy()


# ----------------------------------------------------------------------------------------
# DeepFM first layer fused with the lookup (lr_deepfm_l1_*): deep_embed = concat of the gathered
# rows (algorithms/deepfm.py:236-247) times the (BatchNorm-folded) first kernel of dense_nn
# (layers/dense.py:30-41).  fp64 restatements; ids outside [0,V) give a zero row.  [UNPINNED: TF]
# ----------------------------------------------------------------------------------------
def deepfm_l1_fwd(table, lin, idx, Wp, bias):
    e = embedding_lookup(table.astype(np.float64), idx)                    # [B,F,K]
    B, F, K = e.shape
    z1 = e.reshape(B, F * K) @ Wp.astype(np.float64) + (0.0 if bias is None else bias.astype(np.float64))
    pair, fsum = fm_pairwise(e)
    lin_out = None if lin is None else embedding_lookup(lin.reshape(-1, 1).astype(np.float64), idx)[..., 0]
    return z1, pair, fsum, lin_out


def deepfm_l1_wgrad(table, idx, gz):
    e = embedding_lookup(table.astype(np.float64), idx)
    B, F, K = e.shape
    return e.reshape(B, F * K).T @ gz.astype(np.float64)                   # gather^T @ gz


def deepfm_l1_dgrad(gz, Wp, K, gl, wp, fsum, slotT):
    """ge[slot] = gz[b] @ Wp_f^T + gl[b] * wp * fsum[b] for every kept position (b,f)."""
    F, B = slotT.shape
    G = (gz.astype(np.float64) @ Wp.astype(np.float64).T).reshape(B, F, K)
    if gl is not None:
        G = G + (gl.astype(np.float64)[:, None] * wp.astype(np.float64)[None, :] * fsum.astype(np.float64))[:, None, :]
    ge = np.zeros((B * F, K), dtype=np.float64)
    keep = slotT.T >= 0                                                    # [B,F]
    ge[slotT.T[keep]] = G[keep]
    return ge


def fm_rows_gradient(table, ge, pos, rows, start, F, gl, wp, bn_a, bn_c):
    """Per distinct row: g = sum_p ge[p] - n*a_f - w*(n*c_f + wp*sum_p gl[b(p)]); also sum_p gl[b(p)]."""
    K = table.shape[1]
    g = np.zeros((len(rows), K), dtype=np.float64)
    sgl = np.zeros(len(rows), dtype=np.float64)
    for s in range(len(rows)):
        p0, p1 = int(start[s]), int(start[s + 1])
        n = p1 - p0
        f = int(pos[p0]) % F
        acc = ge[p0:p1].astype(np.float64).sum(axis=0)
        if gl is not None:
            sgl[s] = gl[pos[p0:p1] // F].astype(np.float64).sum()
        w = table[rows[s]].astype(np.float64)
        cw = np.zeros(K) if wp is None else sgl[s] * wp.astype(np.float64)
        if bn_a is not None:
            acc = acc - n * bn_a.reshape(F, K)[f].astype(np.float64)
            cw = cw + n * bn_c.reshape(F, K)[f].astype(np.float64)
        g[s] = acc - w * cw
    return g, sgl


# ----------------------------------------------------------------------------------------
# (a4) FM pairwise term — algorithms/fm.py:158-161, deepfm.py:160-163.  [UNPINNED: TF]
# ----------------------------------------------------------------------------------------
def fm_pairwise(e: np.ndarray):
    s = e.sum(axis=1)
    q = (e * e).sum(axis=1)
    return 0.5 * (s * s - q), s


def fm_pairwise_bwd(e: np.ndarray, gpair: np.ndarray) -> np.ndarray:
    s = e.sum(axis=1)
    return gpair[:, None, :] * (s[:, None, :] - e)


# ----------------------------------------------------------------------------------------
# (a7) din_attention — layers/attention.py:28-64 (dense_nn(.., (16,1), use_bn=False,
# activation=sigmoid): layers/dense.py:12-49; last layer has no activation).  [UNPINNED: TF]
# ----------------------------------------------------------------------------------------
def din_attention(q, keys, lens, W1, b1, W2, b2):
    B, L, K = keys.shape
    qt = np.broadcast_to(q[:, None, :], keys.shape)
    cross = np.concatenate([qt, keys, qt - keys, qt * keys], axis=2)     # attention.py:47-49
    z = cross @ W1 + b1
    hcol = 1.0 / (1.0 + np.exp(-z))                                     # sigmoid, layer 1
    s = (hcol @ W2.reshape(-1, 1)).reshape(B, L) + np.asarray(b2).reshape(())  # layer 2, no act
    s = s * (1.0 / np.sqrt(np.asarray(K, dtype=s.dtype)))               # attention.py:58
    mask = np.arange(L)[None, :] < lens[:, None]                        # tf.sequence_mask
    s = np.where(mask, s, np.asarray(-(2 ** 32) + 1, dtype=s.dtype))    # attention.py:59-60
    s = s - s.max(axis=1, keepdims=True)
    a = np.exp(s)
    a = a / a.sum(axis=1, keepdims=True)                                # tf.nn.softmax
    out = (a[:, None, :] @ keys).reshape(B, K)                          # attention.py:64
    return out.astype(keys.dtype), a.astype(keys.dtype)


# ----------------------------------------------------------------------------------------
# (a16/a17) recommend_from_embedding — recommendation/recommend.py:57-78 and
# rank_recommendations — recommendation/ranking.py:10-56.  PINNED: tests/test_rank_reco.py:7-87
# and fixtures generated from the reference function itself.
# Tie order is unspecified in the reference (argpartition/argsort); this restatement — like the
# HIP kernel — resolves ties by (score desc, id asc).
# ----------------------------------------------------------------------------------------
def rank_recommendations(user_ids, preds, n_rec, n_items, user_consumed, filter_consumed=True):
    if n_rec > n_items:
        raise ValueError(f"`n_rec` {n_rec} exceeds num of items {n_items}")   # ranking.py:21-22
    preds = np.asarray(preds)
    if preds.ndim == 1:
        assert len(preds) % n_items == 0
        preds = preds.reshape(-1, n_items)                                     # ranking.py:23-26
    ids_out = np.full((len(preds), n_rec), -1, dtype=np.int64)
    sc_out = np.full((len(preds), n_rec), -np.inf, dtype=preds.dtype)
    for i, u in enumerate(user_ids):
        p = preds[i]
        ids = np.arange(n_items)
        consumed = user_consumed.get(u, []) if hasattr(user_consumed, "get") else user_consumed[u]
        consumed = list(consumed)
        if can_filter(consumed, n_rec, n_items, filter_consumed):
            mask = np.isin(ids, consumed, invert=True)                         # ranking.py:59-61
            ids, p = ids[mask], p[mask]
        order = np.lexsort((ids, -p.astype(np.float64)))                       # score desc, id asc
        order = order[:n_rec]
        ids_out[i, : len(order)] = ids[order]
        sc_out[i, : len(order)] = p[order]
    return ids_out, sc_out


def rank_recommendations_partition(user_ids, preds, n_rec, n_items, user_consumed, filter_consumed=True):
    """The reference's own selection algorithm — `np.argpartition` of the n_rec best, then one `argsort` of those
    (ranking.py:40-56, `partition_select` :76-78) — i.e. its COST (O(N) per user, where `rank_recommendations`
    above pays a full lexsort for a defined tie order).  Used as the "port" CPU baseline of the recommend leg;
    identical ids wherever the scores are distinct."""
    preds = np.asarray(preds)
    out = np.empty((len(preds), n_rec), dtype=np.int64)
    base = np.arange(n_items)
    for i, u in enumerate(user_ids):
        ids, p = base, preds[i]
        consumed = user_consumed.get(u, []) if hasattr(user_consumed, "get") else user_consumed[u]
        if can_filter(list(consumed), n_rec, n_items, filter_consumed):
            mask = np.isin(ids, consumed, assume_unique=True, invert=True)     # ranking.py:59-62
            ids, p = ids[mask], p[mask]
        part = np.argpartition(p, -n_rec)[-n_rec:]                             # ranking.py:76-78
        order = np.argsort(p[part])[::-1]                                      # ranking.py:47-48
        out[i] = ids[part][order]
    return out


def can_filter(consumed, n_rec, n_items, filter_consumed=True) -> bool:
    """ranking.py:38 — `filter_consumed and consumed and n_rec + len(consumed) <= n_items`."""
    return bool(filter_consumed and len(consumed) > 0 and n_rec + len(consumed) <= n_items)


def recommend_from_embedding(user_embeds, item_embeds, user_ids, n_rec, n_items, user_consumed,
                             filter_consumed=True):
    ue = user_embeds[np.asarray(user_ids)]                                     # recommend.py:66
    preds = ue @ item_embeds[:n_items].T                                       # recommend.py:67-68
    return rank_recommendations(user_ids, preds, n_rec, n_items, user_consumed, filter_consumed)


# ----------------------------------------------------------------------------------------
# (a19) predict_from_embedding — prediction/predict.py:36-40 (+ expit for ranking :18-23).
# ----------------------------------------------------------------------------------------
def pair_dot(U, I, user, item):
    return np.sum(U[user] * I[item], axis=1)


# ----------------------------------------------------------------------------------------
# (a12) LightGCN propagation — algorithms/torch_modules/lightgcn_module.py:36-88.  PINNED by
# fixtures produced with the reference module.
# ----------------------------------------------------------------------------------------
def lightgcn_laplacian(n_users: int, n_items: int, user_consumed):
    """CSR of D^-1/2 A D^-1/2 for the bipartite graph (lightgcn_module.py:36-61)."""
    import scipy.sparse as ssp

    rows, cols = [], []
    for u in range(n_users):
        items = np.unique(np.asarray(user_consumed[u], dtype=np.int64))   # R[u, items] = 1 (binary)
        rows.append(np.full(len(items), u, dtype=np.int64))
        cols.append(items)
    r = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    c = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    n = n_users + n_items
    R = ssp.csr_matrix((np.ones(len(r), np.float32), (r, c + n_users)), shape=(n, n))
    A = (R + R.T).tocsr()
    deg = np.asarray(A.sum(axis=1)).reshape(-1)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5, dtype=np.float64)
    dinv[np.isinf(dinv)] = 0.0
    D = ssp.diags(dinv.astype(np.float32))
    L = (D @ A @ D).tocsr()
    L.sort_indices()
    return L.indptr.astype(np.int64), L.indices.astype(np.int32), L.data.astype(np.float32)


def spmm_csr(rowptr, col, val, X):
    import scipy.sparse as ssp

    n = len(rowptr) - 1
    A = ssp.csr_matrix((val, col, rowptr), shape=(n, X.shape[0]))
    return np.asarray(A @ X, dtype=X.dtype)


def lightgcn_propagate(rowptr, col, val, E0, n_layers):
    """mean(E^0..E^L), E^{l+1} = A E^l (lightgcn_module.py:66-88, no dropout)."""
    acc = E0.astype(np.float32).copy()
    cur = E0.astype(np.float32)
    for _ in range(n_layers):
        cur = spmm_csr(rowptr, col, val, cur)
        acc += cur
    return acc / np.float32(n_layers + 1)


# ----------------------------------------------------------------------------------------
# data/consumed.py:7-17 with the canonical (Rust, rust/src/utils.rs:8-35) semantics:
# consecutive-duplicate removal.  PINNED: tests/test_consumed.py:12-25, rust/src/utils.rs:41-59.
# ----------------------------------------------------------------------------------------
def interaction_consumed(user_indices, item_indices):
    uc, ic = {}, {}
    for u, i in zip(list(user_indices), list(item_indices)):
        uc.setdefault(int(u), []).append(int(i))
        ic.setdefault(int(i), []).append(int(u))

    def dedup(v):
        out = [v[0]]
        for x in v[1:]:
            if x != out[-1]:
                out.append(x)
        return out

    return {k: dedup(v) for k, v in uc.items()}, {k: dedup(v) for k, v in ic.items()}


# ----------------------------------------------------------------------------------------
# Device negative sampler (librecommender_amd/csrc/sampling.hip) — restatement of ITS algorithm.
# The acceptance rules are the reference's (sampling/negatives.py:17-31 "random": != positive;
# :55-82 "unconsumed": also != earlier negatives of that positive and, for the first 10 of 20
# tries, not in the user's consumed set); the random numbers come from a counter-based generator
# because the reference's numpy / Python RNG streams cannot be reproduced on a device.
# PINNED only against itself (bit-exact kernel == this function); distributional properties and
# the acceptance rules are tested separately.
# ----------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _mix64(z: int) -> int:
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def sample_negatives_counter(users, items_pos, num_neg, n_items, user_consumed=None, seed=0):
    items_pos = np.asarray(items_pos)
    out = np.empty(len(items_pos) * num_neg, dtype=np.int32)
    for p, pos in enumerate(items_pos.tolist()):
        cons = set(user_consumed.get(int(users[p]), ())) if user_consumed is not None else ()
        for j in range(num_neg):
            ctr = (p * num_neg + j) * 32
            cand = 0
            for t in range(20):
                z = _mix64((seed + 0x9E3779B97F4A7C15 * (ctr + t + 1)) & _M64)
                cand = ((z >> 32) * n_items) >> 32
                bad = cand == pos or cand in out[p * num_neg: p * num_neg + j].tolist()
                if not bad and t < 10 and cand in cons:
                    bad = True
                if not bad:
                    break
            out[p * num_neg + j] = cand
    return out


# ----------------------------------------------------------------------------------------
# tfops/rebuild.py:49-74 (and torchops/rebuild.py:55-75,108-122): variables of a saved model are
# scattered into the larger variables of the rebuilt one.  Index work -> bit-exact.
# ----------------------------------------------------------------------------------------
def rebuild_assign(new_var, old_var, kind, old_n_users, old_n_items, old_sparse_len, old_sparse_oov,
                   new_sparse_offset):
    out = new_var.copy()
    if kind == "user":
        out[:old_n_users] = old_var[:old_n_users]                      # "remove oov values"
    elif kind == "item":
        out[:old_n_items] = old_var[:old_n_items]
    elif kind == "sparse":
        old = np.delete(old_var, old_sparse_oov, axis=0)
        indices = []
        for offset, size in zip(new_sparse_offset, old_sparse_len):
            if size != -1:
                indices.extend(range(offset, offset + size))
        out[indices] = old
    return out


# ----------------------------------------------------------------------------------------
# in-batch softmax cross-entropy — tfops/loss.py:71-75 over adjust_logits (two_tower.py:458-479)
# ----------------------------------------------------------------------------------------
def softmax_ce(X, Y, col_bias=None, row_ids=None, col_ids=None, pos0=0, g=None):
    """fp64 restatement: logits = X @ Y.T (two_tower.py:466 divides by the temperature before: the caller
    passes X / T), `logits -= logQ` (:467-470, col_bias = -logQ), accidental hits -> float32.min (:472-479),
    labels = the diagonal (loss.py:74), `sparse_softmax_cross_entropy_with_logits` (:75).
    Returns (loss[B], dX, dY) with the gradients of sum_i g[i] loss[i] (g = 1 by default)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    B, N = X.shape[0], Y.shape[0]
    logits = X @ Y.T
    if col_bias is not None:
        logits = logits + np.asarray(col_bias, dtype=np.float64)[None, :]
    lab = pos0 + np.arange(B)
    if row_ids is not None:
        equal = np.asarray(row_ids)[:, None] == np.asarray(col_ids)[None, :]
        equal[np.arange(B), lab] = False                         # equal_items - label_diag
        logits = np.where(equal, float(np.finfo(np.float32).min), logits)
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    ssum = e.sum(axis=1, keepdims=True)
    loss = (m + np.log(ssum))[:, 0] - logits[np.arange(B), lab]
    P = e / ssum
    G = P.copy()
    G[np.arange(B), lab] -= 1.0
    if g is not None:
        G *= np.asarray(g, dtype=np.float64)[:, None]
    return loss, G @ Y, G.T @ X
