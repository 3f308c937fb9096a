"""User/item interaction histories (`libreco/data/consumed.py:7-17`).

Canonical semantics are those of the reference's Rust helper (`rust/src/utils.rs:8-35`, pinned by
`tests/test_consumed.py:12-25`): histories keep interaction order and drop only CONSECUTIVE
repeats.  Implemented with one stable argsort per side instead of a Python loop per interaction.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def _grouped_histories(keys: np.ndarray, vals: np.ndarray) -> Dict[int, List[int]]:
    if len(keys) == 0:
        return {}
    order = np.argsort(keys, kind="stable")          # keeps interaction order inside a key
    k, v = keys[order], vals[order]
    keep = np.ones(len(k), dtype=bool)
    keep[1:] = (k[1:] != k[:-1]) | (v[1:] != v[:-1])  # drop consecutive repeats within a key
    k, v = k[keep], v[keep]
    bounds = np.flatnonzero(np.r_[True, k[1:] != k[:-1], True])
    out = {int(k[a]): v[a:b].tolist() for a, b in zip(bounds[:-1], bounds[1:])}
    uniq, first_idx = np.unique(keys, return_index=True)   # dict order = first appearance
    return {int(key): out[int(key)] for key in uniq[np.argsort(first_idx)]}


def interaction_consumed(user_indices, item_indices) -> Tuple[Dict[int, List[int]], Dict[int, List[int]]]:
    u = np.asarray(user_indices, dtype=np.int64)
    i = np.asarray(item_indices, dtype=np.int64)
    return _grouped_histories(u, i), _grouped_histories(i, u)


def merge_consumed(new: Dict[int, List[int]], n: int, old: Dict[int, List[int]], merge: bool):
    """Retrain helper (`data/consumed.py:42-68`): old history followed by new, or new-else-old."""
    out = {}
    for k in range(n):
        if k not in new and k not in old:
            raise AssertionError(f"id {k} has no history in old or new data")
        if merge and k in new and k in old:
            out[k] = old[k] + new[k]
        else:
            out[k] = new[k] if k in new else old[k]
    return out


def _merge_dedup(new_consumed, num, old_consumed):
    """The reference's name for `merge_consumed(..., merge=True)` (`data/consumed.py:55-63`)."""
    return merge_consumed(new_consumed, num, old_consumed, True)


def _fill_empty(consumed, num, old_consumed):
    """The reference's name for `merge_consumed(..., merge=False)` (`data/consumed.py:66-68`)."""
    return merge_consumed(consumed, num, old_consumed, False)
