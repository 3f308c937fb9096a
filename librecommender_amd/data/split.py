"""Train/eval splitting helpers used by the examples and tests
(`libreco/data/split.py:8-210,298-341`).  Host-side pandas work, off the hot path."""
from __future__ import annotations

import math

import numpy as np


def _ratios(test_size, multi_ratios):
    if not test_size and not multi_ratios:
        raise ValueError("must provide either 'test_size' or 'multi_ratios'")
    if test_size is not None:
        assert isinstance(test_size, float), "test_size must be float value"
        assert 0.0 < test_size < 1.0, "test_size must be in (0.0, 1.0)"
        return [1 - test_size, test_size]
    if not isinstance(multi_ratios, (list, tuple)):
        raise ValueError("multi_ratios should be list or tuple")
    assert len(multi_ratios) > 1, "multi_ratios must at least have two elements"
    assert all(r > 0.0 for r in multi_ratios), "ratios should be positive values"
    total = math.fsum(multi_ratios)
    return [r / total for r in multi_ratios] if total != 1.0 else list(multi_ratios)


def _drop_unknown(parts):
    train = parts[0]
    users, items = set(train["user"].tolist()), set(train["item"].tolist())
    out = [train]
    for p in parts[1:]:
        known = p["user"].isin(users).to_numpy() & p["item"].isin(items).to_numpy()
        out.append(p[known])
    return out


def _pad_unknown(parts, pad_val):
    u_pad, i_pad = pad_val if isinstance(pad_val, (list, tuple)) else (pad_val, pad_val)
    train = parts[0]
    users, items = set(train.user.tolist()), set(train.item.tolist())
    out = [train]
    for p in parts[1:]:
        q = p.copy()
        q.loc[~p.user.isin(users), "user"] = u_pad
        q.loc[~p.item.isin(items), "item"] = i_pad
        out.append(q)
    return out


def split_by_ratio(data, order=True, shuffle=False, test_size=None, multi_ratios=None,
                   filter_unknown=True, pad_unknown=False, pad_val=None, seed=42):
    """Per-user split: each user's rows (in data order) are cut at the cumulative ratios; users
    with <= 3 rows stay entirely in the first part (`split.py:120-208`)."""
    assert "user" in data.columns, "data must contains user column"
    ratios = _ratios(test_size, multi_ratios)
    cuts = np.cumsum(ratios)[:-1]
    codes = np.unique(data.user.to_numpy(), return_inverse=True)[1]
    order_idx = np.argsort(codes, kind="stable" if order else "quicksort")
    counts = np.bincount(codes)
    buckets = [[] for _ in ratios]
    pos = 0
    for c in counts.tolist():
        rows = order_idx[pos:pos + c]
        pos += c
        if c <= 3:
            buckets[0].append(rows)
            continue
        edges = [0] + [round(x * c) for x in cuts.tolist()] + [c]
        for b in range(len(ratios)):
            buckets[b].append(rows[edges[b]:edges[b + 1]])
    index_lists = [np.concatenate(b) if b else np.zeros(0, dtype=np.int64) for b in buckets]
    if shuffle:
        rng = np.random.default_rng(seed)
        index_lists = [rng.permutation(ix) for ix in index_lists]
    parts = [data.iloc[ix] for ix in index_lists]
    if filter_unknown:
        parts = _drop_unknown(parts)
    elif pad_unknown and pad_val is not None:
        parts = _pad_unknown(parts, pad_val)
    return parts


def split_by_ratio_chrono(data, order=True, shuffle=False, test_size=None, multi_ratios=None, seed=42):
    """Sort by `time`, then `split_by_ratio` (`split.py:298-341`)."""
    assert "user" in data.columns and "time" in data.columns, "data must contains user and time column"
    data = data.sort_values(by=["time"]).reset_index(drop=True)
    return split_by_ratio(data, order, shuffle, test_size, multi_ratios, seed=seed)


def random_split(data, shuffle=True, test_size=None, multi_ratios=None, filter_unknown=True,
                 pad_unknown=False, pad_val=None, seed=42):
    """Row-wise random split (`split.py:8-78`, sklearn.train_test_split underneath)."""
    from sklearn.model_selection import train_test_split

    ratios = _ratios(test_size, multi_ratios)
    rest, parts = data.copy(), []
    for _ in range(len(ratios) - 1):
        size = ratios.pop(-1)
        total = math.fsum(ratios)
        ratios = [r / total for r in ratios]
        rest, part = train_test_split(rest, test_size=size, shuffle=shuffle, random_state=seed)
        parts.insert(0, part)
    parts.insert(0, rest)
    if filter_unknown:
        parts = _drop_unknown(parts)
    elif pad_unknown and pad_val is not None:
        parts = _pad_unknown(parts, pad_val)
    return parts


def split_by_num(data, order=True, shuffle=False, test_size=1, filter_unknown=True, pad_unknown=False,
                 pad_val=None, seed=42):
    """Per-user split holding out each user's last `test_size` rows; users with <= 3 rows stay in
    train, users with <= test_size rows give up only their last one (`split.py:211-295`)."""
    assert "user" in data.columns, "data must contains user column"
    assert isinstance(test_size, int), "test_size must be int value"
    assert 0 < test_size < len(data), "test_size must be in (0, len(data))"
    codes = np.unique(data.user.to_numpy(), return_inverse=True)[1]
    order_idx = np.argsort(codes, kind="stable" if order else "quicksort")
    train, test, pos = [], [], 0
    for c in np.bincount(codes).tolist():
        rows = order_idx[pos:pos + c]
        pos += c
        if c <= 3:
            train.append(rows)
        elif c <= test_size:
            train.append(rows[:-1]); test.append(rows[-1:])
        else:
            train.append(rows[: c - test_size]); test.append(rows[-test_size:])
    index_lists = [np.concatenate(b) if b else np.zeros(0, dtype=np.int64) for b in (train, test)]
    if shuffle:
        rng = np.random.default_rng(seed)
        index_lists = [rng.permutation(ix) for ix in index_lists]
    parts = [data.iloc[ix] for ix in index_lists]
    if filter_unknown:
        parts = _drop_unknown(parts)
    elif pad_unknown and pad_val is not None:
        parts = _pad_unknown(parts, pad_val)
    return parts


def split_by_num_chrono(data, order=True, shuffle=False, test_size=1, seed=42):
    """Sort by `time`, then `split_by_num` (`split.py:344-382`)."""
    assert "user" in data.columns and "time" in data.columns, "data must contains user and time column"
    data = data.sort_values(by=["time"]).reset_index(drop=True)
    return split_by_num(data, order, shuffle, test_size, seed=seed)
