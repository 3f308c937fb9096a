"""Temporary feature overrides for single-user inference (`user_feats=` / `feats=` arguments;
`libreco/prediction/preprocess.py:121-168`, `recommendation/preprocess.py:215-256`): values unknown
to the training vocabulary and columns that are not features are silently ignored."""
from __future__ import annotations

import numpy as np


def _vocab_for(data_info, col):
    m = data_info.col_name_mapping
    main = m.get("multi_sparse", {}).get(col, col) if m else col
    return data_info.sparse_idx_mapping.get(main) if data_info.sparse_idx_mapping else None


def override_sparse(data_info, row: np.ndarray, feat_dict, cols=None) -> np.ndarray:
    """`row` is [1, n_cols]; `cols` = names of its columns (None -> all sparse columns)."""
    row = row.copy()
    names = list(cols) if cols is not None else data_info.sparse_col.name
    all_pos = data_info.col_name_mapping.get("sparse_col", {}) if data_info.col_name_mapping else {}
    for col, val in feat_dict.items():
        if col not in names or col not in all_pos:
            continue
        vocab = _vocab_for(data_info, col)
        if vocab is not None and val in vocab:
            row[:, names.index(col)] = vocab[val] + data_info.sparse_offset[all_pos[col]]
    return row


def override_dense(data_info, row: np.ndarray, feat_dict, cols=None) -> np.ndarray:
    row = row.copy()
    names = list(cols) if cols is not None else data_info.dense_col.name
    for col, val in feat_dict.items():
        if col in names:
            row[:, names.index(col)] = val
    return row
