"""Behaviour-sequence builders for sequence models (`libreco/batch/sequence.py:33-91`)."""
from __future__ import annotations

import random

import numpy as np

from .. import _hostlib


class SequenceBuilder:
    """Histories as one CSR + a sorted (user, item) -> first position table, so a batch is a few
    vectorised gathers instead of a per-row `list.index` loop (`batch/sequence.py:33-72`)."""

    def __init__(self, user_consumed, n_items: int, max_seq_len: int, mode: str = "recent"):
        self.pad = n_items
        self.L = max_seq_len
        self.mode = mode
        n_u = (max(user_consumed) + 1) if len(user_consumed) else 0
        counts = np.zeros(n_u, dtype=np.int64)
        for u, h in user_consumed.items():
            counts[u] = len(h)
        self.ptr = np.concatenate([[0], np.cumsum(counts)])
        self.hist = np.zeros(int(self.ptr[-1]) + 1, dtype=np.int64)        # +1: a safe slot for masked reads
        for u, h in user_consumed.items():
            self.hist[self.ptr[u]:self.ptr[u + 1]] = h
        self.counts = counts
        owner = np.repeat(np.arange(n_u, dtype=np.int64), counts)
        where = np.arange(int(self.ptr[-1]), dtype=np.int64) - self.ptr[owner]
        self.stride = int(max(n_items, self.hist.max(initial=0)) + 1)
        key = owner * self.stride + self.hist[:-1]
        order = np.argsort(key, kind="stable")                               # first occurrence first
        sk = key[order]
        first = np.ones(len(sk), dtype=bool)
        first[1:] = sk[1:] != sk[:-1]
        self.keys, self.first_pos = np.ascontiguousarray(sk[first]), np.ascontiguousarray(where[order][first])
        self.kptr = np.concatenate([[0], np.cumsum(np.bincount(self.keys // self.stride, minlength=n_u))]).astype(np.int64)

    def positions(self, users, items):
        """Index of `item` in the user's history (first occurrence), -1 when it is not there."""
        got = _hostlib.pair_positions(self.keys, self.first_pos, self.kptr, self.stride, users, items) if len(self.keys) else None
        if got is not None:                                    # one binary search per row inside the user's own keys
            return got
        key = users.astype(np.int64) * self.stride + items.astype(np.int64)
        order = np.argsort(key, kind="stable")                 # ascending queries walk the table cache-friendly
        j = np.empty(len(key), dtype=np.int64)
        j[order] = np.searchsorted(self.keys, key[order])
        j_safe = np.minimum(j, max(len(self.keys) - 1, 0))
        found = (j < len(self.keys)) & (self.keys[j_safe] == key) if len(self.keys) else np.zeros(len(key), bool)
        return np.where(found, self.first_pos[j_safe] if len(self.keys) else -1, -1)

    def _positions_or_random(self, users, items):
        """Position of `item` in the user's history; a negative item takes `random.randrange(len(history))`, drawn
        in batch order from Python's generator (sequence.py:49-55, :108-113)."""
        pos = self.positions(users, items)
        missing = np.flatnonzero(pos < 0)
        if len(missing):
            widths = self.counts[users[missing]]
            drawn = _hostlib.randrange_stream(widths)          # C loop on the generator's own state
            pos[missing] = drawn if drawn is not None else [random.randrange(0, n) for n in widths.tolist()]
        return pos

    def _windows(self, users, start, n, width):
        """[len(users), width] left-aligned copies of hist[ptr[u] + start : ... + n], padded."""
        base = self.ptr[users] + start
        out = _hostlib.seq_windows(self.hist, base, n, width, self.pad)       # one C pass; the numpy form below defines it
        if out is not None:
            return out
        t = np.arange(width, dtype=np.int64)[None, :]
        valid = t < n[:, None]
        src = np.where(valid, base[:, None] + t, len(self.hist) - 1)
        return np.where(valid, self.hist[src], self.pad).astype(np.int32)

    def training_dual_seqs(self, users, items, long_max_len, short_max_len):
        """SIM's (long, short) windows (`get_dual_seqs`, sequence.py:95-147): the `short_max_len` items right before
        the item's position, and up to `long_max_len` items before those; lengths >= 1 (an empty window is one pad)."""
        users, items = np.asarray(users), np.asarray(items)
        Lg, S = int(long_max_len), int(short_max_len)
        pos = self._positions_or_random(users, items)
        short_n = np.minimum(pos, S)
        long_n = np.clip(pos - S, 0, Lg)
        short = self._windows(users, np.maximum(pos - S, 0), short_n, S)
        long = self._windows(users, np.maximum(pos - S - Lg, 0), long_n, Lg)
        return long, np.maximum(long_n, 1).astype(np.int32), short, np.maximum(short_n, 1).astype(np.int32)

    def training_seqs(self, users, items, np_rng=None):
        """Left-aligned window of the <= L items consumed BEFORE `item`; a negative item takes a
        random position `random.randrange(len(history))`, drawn in batch order (sequence.py:49-55);
        length >= 1 even with empty history (the single key is then the pad id, quirk 5 of SURVEY §8)."""
        users, items = np.asarray(users), np.asarray(items)
        L = self.L
        pos = self._positions_or_random(users, items)
        start = np.maximum(pos - L, 0)
        length = np.minimum(pos, L)
        seqs = self._windows(users, start, length, L)
        if self.mode != "recent":                                           # random windows of long histories
            for j in np.flatnonzero(pos >= L):
                u = users[j]
                seqs[j] = np_rng.choice(self.hist[self.ptr[u]:self.ptr[u + 1]].tolist(), L, replace=False)
        return seqs, np.maximum(length, 1).astype(np.int32)


def get_interacted_seqs(user_indices, item_indices, user_consumed, pad_index, mode, max_seq_len,
                        user_consumed_set=None, np_rng=None):
    """Functional form with the reference's signature."""
    return SequenceBuilder(user_consumed, pad_index, max_seq_len, mode).training_seqs(
        np.asarray(user_indices), np.asarray(item_indices), np_rng)


def get_recent_seqs(n_users, user_consumed, pad_index, max_seq_len):
    """Most recent <= L items of every user + one all-pad OOV row of length 1 (sequence.py:75-91)."""
    L = int(max_seq_len)
    counts = np.zeros(n_users, dtype=np.int64)
    for u in range(n_users):
        counts[u] = len(user_consumed[u])
    ptr = np.concatenate([[0], np.cumsum(counts)])
    hist = np.full(int(ptr[-1]) + 1, pad_index, dtype=np.int64)
    for u in range(n_users):
        hist[ptr[u]:ptr[u + 1]] = user_consumed[u]
    take = np.minimum(counts, L)
    t = np.arange(L, dtype=np.int64)[None, :]
    valid = t < take[:, None]
    src = np.where(valid, (ptr[1:] - take)[:, None] + t, len(hist) - 1)
    seqs = np.full((n_users + 1, L), pad_index, dtype=np.int32)
    seqs[:n_users] = np.where(valid, hist[src], pad_index)
    lens = np.ones(n_users + 1, dtype=np.int32)
    lens[:n_users] = take
    return seqs, lens


def get_dual_seqs(user_indices, item_indices, user_consumed, pad_index, long_max_len, short_max_len,
                  user_consumed_set=None):
    """Functional form with the reference's signature (sequence.py:95-147)."""
    return SequenceBuilder(user_consumed, pad_index, 1).training_dual_seqs(
        np.asarray(user_indices), np.asarray(item_indices), long_max_len, short_max_len)


def get_recent_dual_seqs(n_users, user_consumed, pad_index, long_max_len, short_max_len):
    """Every user's most recent (long, short) windows + one all-pad OOV row of lengths 1 (sequence.py:150-193):
    the windows `get_dual_seqs` would build at position len(history)."""
    b = SequenceBuilder({u: user_consumed[u] for u in range(n_users)}, pad_index, 1)
    users = np.arange(n_users, dtype=np.int64)
    Lg, S = int(long_max_len), int(short_max_len)
    pos = b.counts[:n_users].copy()
    short_n, long_n = np.minimum(pos, S), np.clip(pos - S, 0, Lg)
    short = b._windows(users, np.maximum(pos - S, 0), short_n, S)
    long = b._windows(users, np.maximum(pos - S - Lg, 0), long_n, Lg)
    pad_row = lambda w: np.full((1, w), pad_index, dtype=np.int32)  # noqa: E731
    one = np.ones(1, dtype=np.int32)
    return (np.vstack([long, pad_row(Lg)]), np.concatenate([np.maximum(long_n, 1).astype(np.int32), one]),
            np.vstack([short, pad_row(S)]), np.concatenate([np.maximum(short_n, 1).astype(np.int32), one]))
