"""Retraining data merge: new interaction data is appended to what a previous
`DataInfo` knows — ids, category vocabularies, feature matrices and histories grow, indices of
everything already known stay put (`libreco/data/dataset.py:148-196,262-345,548-700`,
`feature/update.py`, `data/consumed.py:42-68`, `data/data_info.py:543-578`).  Pure host index work;
pinned bit-exactly by fixtures produced with the reference's own implementation
(tests/golden/retrain.npz)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List

import numpy as np

from .consumed import interaction_consumed, merge_consumed
from .vocab import SparseSchema, encode


@dataclass
class OldInfo:
    """What a rebuilt model needs to know about the previous run (data_info.py:542-548)."""
    n_users: int
    n_items: int
    sparse_len: List[int]
    sparse_oov: List[int]
    popular_items: List[Any]


def store_old_info(info) -> OldInfo:
    """data_info.py:551-578 — note its quirk: the trailing columns of a multi-sparse field get
    length -1 and NO oov entry."""
    sparse_len, sparse_oov = [], []
    su, mu = info.sparse_unique_vals, info.multi_sparse_unique_vals
    multi_map = (info.col_name_mapping or {}).get("multi_sparse", {})
    for i, col in enumerate(info.sparse_col.name):
        if su is not None and col in su:
            sparse_len.append(len(su[col]))
            sparse_oov.append(int(info.sparse_oov[i]))
        elif mu is not None and col in mu:
            sparse_len.append(len(mu[col]))
            sparse_oov.append(int(info.sparse_oov[i]))
        elif col in multi_map:
            sparse_len.append(-1)
    return OldInfo(info.n_users, info.n_items, sparse_len, sparse_oov, list(info.popular_items))


def update_unique_vals(data_vals, old_unique, pad_val=None):
    """Known values keep their index, unseen ones are appended in sorted order (update.py:8-15)."""
    diff = np.setdiff1d(np.asarray(data_vals), np.asarray(old_unique))
    if pad_val is not None:
        diff = diff[diff != pad_val]
    return np.append(old_unique, diff) if len(diff) else old_unique


def merged_schema(data, info) -> SparseSchema:
    """Column layout of `info` with vocabularies extended by `data` (update.py:26-66)."""
    names = info.sparse_col.name
    multi_map = (info.col_name_mapping or {}).get("multi_sparse", {})
    old_s, old_m = info.sparse_unique_vals or {}, info.multi_sparse_unique_vals or {}
    for col in names:
        if col not in data.columns:
            raise ValueError(f"Old column `{col}` doesn't exist in new data")
    plain = [c for c in names if c in old_s]
    fields = []
    for c in names:
        if c in old_m:
            fields.append([c])
        elif c in multi_map:
            fields[[f[0] for f in fields].index(multi_map[c])].append(c)
    schema = SparseSchema(sparse_cols=plain, multi_fields=fields)
    for c in plain:
        schema.vocab[c] = update_unique_vals(np.unique(data[c]), old_s[c])
    pads = info.multi_sparse_combine_info.pad_val if fields else {}
    for f in fields:
        vals = []
        for c in f:
            vals.extend(np.unique(data[c]))
        schema.multi_vocab[f[0]] = update_unique_vals(vals, old_m[f[0]], pads[f[0]])
        schema.pad_val[f[0]] = pads[f[0]]
    return schema


def _grow_sparse(info, old, cols, new_offset, new_oov, new_num):
    """get_sparse_feats (update.py:131-145): re-base known rows on the new offsets, move old OOV
    markers to the new OOV rows, add all-OOV rows for new ids."""
    if old is None:
        return None
    old = old[:-1]
    cols = list(cols)
    out = old + (np.asarray(new_offset)[cols] - np.asarray(info.sparse_offset)[cols])
    for j, c in enumerate(cols):
        out[old[:, j] == info.sparse_oov[c], j] = new_oov[c]
    if new_num > len(old):
        out = np.vstack([out, np.full((new_num - len(old), old.shape[1]), np.asarray(new_oov)[cols], old.dtype)])
    return out


def _grow_dense(old, new_num):
    if old is None:
        return None
    out = old[:-1]
    if new_num > len(out):
        out = np.vstack([out, np.zeros((new_num - len(out), old.shape[1]), old.dtype)])
    return out


def update_unique_feats(data, info, unique_ids, schema: SparseSchema, is_user: bool):
    """Feature matrices of all users / items after the merge: last occurrence in the new data wins
    (update.py:68-128, 179-228)."""
    key = "user" if is_user else "item"
    data = data.drop_duplicates(subset=[key], keep="last")
    sp_info = info.user_sparse_col if is_user else info.item_sparse_col
    ds_info = info.user_dense_col if is_user else info.item_dense_col
    offsets, oovs = schema.offsets, schema.oov_rows
    sp = _grow_sparse(info, info.user_sparse_unique if is_user else info.item_sparse_unique,
                      sp_info.index, offsets, oovs, len(unique_ids))
    ds = _grow_dense(info.user_dense_unique if is_user else info.item_dense_unique, len(unique_ids))
    rows = encode(data[key].to_numpy(), unique_ids, allow_unknown=False)
    multi_map = (info.col_name_mapping or {}).get("multi_sparse", {})
    if sp is not None:
        for j, (col, ci) in enumerate(zip(sp_info.name, sp_info.index)):
            if col in multi_map:
                vocab = schema.multi_vocab[multi_map[col]]
            elif col in schema.multi_vocab:
                vocab = schema.multi_vocab[col]
            else:
                vocab = schema.vocab[col]
            idx = encode(data[col].to_numpy(), vocab, allow_unknown=True)
            known = idx < len(vocab)                     # padding values are skipped
            sp[rows[known], j] = offsets[ci] + idx[known]
    if ds is not None:
        for j, col in enumerate(ds_info.name):
            ds[rows, j] = data[col].to_numpy(np.float32)
    return sp, ds


def update_consumed(user_indices, item_indices, n_users, n_items, info, merge_behavior):
    uc, ic = interaction_consumed(user_indices, item_indices)
    return (merge_consumed(uc, n_users, info.user_consumed, merge_behavior),
            merge_consumed(ic, n_items, info.item_consumed, merge_behavior))
