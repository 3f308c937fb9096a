"""Model side of retraining: a freshly built, larger model takes over the
variables of a saved one (`libreco/tfops/rebuild.py:12-139`, `torchops/rebuild.py:13-148`).

Users and items keep their inner ids (new ones are appended), so their rows are copied 1:1; the
shared sparse table is re-based column by column: the saved rows minus the old OOV rows go to
`[new_offset[c], new_offset[c] + old_len[c])` of every column `c` (multi-sparse trailing columns,
`old_len == -1`, share their field's rows).  Rows of new ids / categories and all OOV rows keep the
new model's fresh initialisation; Adam moments follow the same map (zero elsewhere)."""
from __future__ import annotations

import numpy as np


def sparse_growth_index(old_rows: int, old_info, new_sparse_offset):
    """(src rows of the saved sparse variable, dst rows of the new one) — rebuild.py:63-73."""
    keep = np.setdiff1d(np.arange(old_rows), np.asarray(old_info.sparse_oov, dtype=np.int64))
    dst = []
    for off, size in zip(new_sparse_offset, old_info.sparse_len):
        if size != -1:
            dst.extend(range(int(off), int(off) + int(size)))
    dst = np.asarray(dst, dtype=np.int64)
    if len(dst) != len(keep):
        raise ValueError("saved sparse variable does not match `old_info` (rows without OOV "
                         f"{len(keep)} vs sum of old column sizes {len(dst)})")
    return keep, dst


def table_growth_index(old_rows: int, old_info, new_info, item_oov_row: bool = True):
    """(src, dst) global rows for the concatenated `[user(+oov) | item(+oov) | sparse]` layout of
    `layers.FieldTables`."""
    uo, no = int(old_info.n_users), int(old_info.n_items)
    extra = 1 if item_oov_row else 0
    i_off_old, s_off_old = uo + 1, uo + 1 + no + extra
    i_off_new = new_info.n_users + 1
    s_off_new = i_off_new + new_info.n_items + extra
    src = [np.arange(uo), i_off_old + np.arange(no)]
    dst = [np.arange(uo), i_off_new + np.arange(no)]
    s_rows = old_rows - s_off_old
    if s_rows > 0:
        k, d = sparse_growth_index(s_rows, old_info, new_info.sparse_offset)
        src.append(s_off_old + k)
        dst.append(s_off_new + d)
    return np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64)
