"""Self-supervised feature masking of the two-tower model (`libreco/feature/ssl.py`): per batch a
fresh draw of items, their [item id | item sparse features] index row in the "ssl table"
`[zero row | item_embeds_var | sparse_embeds_var]` (two_tower.py:295-304), and two views of it with
complementary / random / mutual-information-correlated columns masked to the zero row.  Host index
work on `data_info.np_rng`; pinned bit-exactly by tests/golden/ssl.npz."""
from __future__ import annotations

import numpy as np


def get_ssl_features(model, batch_size):
    """-> (left_sparse [B, 1+Fi], right_sparse [B, 1+Fi], item_dense [B, Fd] or None)."""
    info = model.data_info
    rng, n_items = info.np_rng, model.n_items
    items = rng.choice(n_items, size=batch_size, replace=not (batch_size < n_items))
    feats = info.item_sparse_unique[items]
    idx = np.hstack([np.expand_dims(items + 1, 1), feats + n_items + 1])      # 0 = the zero row
    n_col = idx.shape[1]
    mid = n_col // 2
    if model.ssl_pattern.startswith("cfm"):
        seed_col = rng.integers(n_col)
        left_cols = model.sparse_feat_mutual_info[seed_col]
        right_cols = np.setdiff1d(range(n_col), left_cols)
    elif model.ssl_pattern.endswith("complementary"):
        left_cols, right_cols = np.split(rng.permutation(n_col), [mid])
    else:
        left_cols = rng.permutation(n_col)[:mid]
        right_cols = rng.permutation(n_col)[:mid]
    left, right = idx.copy(), idx.copy()
    left[:, left_cols] = 0
    right[:, right_cols] = 0
    dense = info.item_dense_unique[items] if getattr(model, "item_dense", False) else None
    return left, right, dense


def get_mutual_info(data, data_info):
    """For every column of [item id | item sparse features]: the n//2 columns it shares most mutual
    information with (ssl.py:43-61)."""
    from sklearn.metrics import mutual_info_score

    cols = np.hstack([np.expand_dims(data.item_indices, 1), data.sparse_indices[:, data_info.item_sparse_col.index]])
    n = cols.shape[1]
    mi = np.zeros((n, n))
    np.fill_diagonal(mi, -1)
    for i in range(n):
        for j in range(i + 1, n):
            mi[i][j] = mi[j][i] = mutual_info_score(cols[:, i], cols[:, j])
    top = np.argsort(mi, axis=1)[:, -(n // 2):]
    return {i: top[i] for i in range(n)}
