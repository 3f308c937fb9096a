"""Feature rows for prediction from a user-supplied frame (`libreco/prediction/preprocess.py:134-168`)
and per-user catalog feature blocks (`recommendation/preprocess.py:175-212`)."""
from __future__ import annotations

import numpy as np

from ..data.vocab import encode


def _column_vocab(data_info, col):
    multi = (data_info.col_name_mapping or {}).get("multi_sparse", {})
    main = multi.get(col, col)
    if data_info.multi_sparse_unique_vals and main in data_info.multi_sparse_unique_vals:
        return data_info.multi_sparse_unique_vals[main]
    return data_info.sparse_unique_vals[main]


def features_from_batch(data_info, sparse, dense, data):
    """Encode the feature columns of `data` with the training vocabularies: unknown categories take
    the field's OOV row; a feature column missing from `data` is an error."""
    sparse_indices = dense_values = None
    if sparse:
        fields = data_info.col_name_mapping["sparse_col"]
        sparse_indices = np.zeros((len(data), len(fields)), np.int32)
        for col, f in fields.items():
            if col not in data.columns:
                raise ValueError(f"Column `{col}` doesn't exist in data")
            vocab = _column_vocab(data_info, col)
            idx = encode(data[col].to_numpy(), vocab, allow_unknown=True)
            sparse_indices[:, f] = np.where(idx < len(vocab), idx + data_info.sparse_offset[f],
                                            data_info.sparse_oov[f])
    if dense:
        cols = list(data_info.col_name_mapping["dense_col"])
        for col in cols:
            if col not in data.columns:
                raise ValueError(f"Column `{col}` doesn't exist in data")
        dense_values = data[cols].to_numpy(dtype=np.float32)
    return sparse_indices, dense_values


def catalog_features(data_info, user, n_items, sparse=True, dense=True):
    """Feature rows of (user, item) for items 0..n_items-1, in original column order."""
    from ..bases.feat_base import merge_user_item_feats
    sp, de = merge_user_item_feats(data_info, np.full(n_items, user), np.arange(n_items))
    return (sp if sparse else None), (de if dense else None)


def user_tower_features(data_info, user_id, user_feats=None):
    """User-side sparse rows / dense values for the embed models' dynamic user vector
    (`recommendation/preprocess.py:88-108`); `user_id=None` -> every known user (OOV row dropped)."""
    from ..feature_override import override_dense, override_sparse
    d = data_info
    sp = de = None
    if user_id is None:
        sp = None if d.user_sparse_unique is None else d.user_sparse_unique[:-1]
        de = None if d.user_dense_unique is None else d.user_dense_unique[:-1]
        return sp, de
    if d.user_sparse_unique is not None:
        sp = d.user_sparse_unique[user_id]
        if user_feats is not None:
            sp = override_sparse(d, sp, user_feats, d.user_sparse_col.name)
    if d.user_dense_unique is not None:
        de = d.user_dense_unique[user_id]
        if user_feats is not None:
            de = override_dense(d, de, user_feats, d.user_dense_col.name)
    return sp, de
