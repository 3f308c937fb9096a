"""Behaviour-sequence builders for sequence models (`libreco/batch/sequence.py:33-91`)."""
from __future__ import annotations

import random

import numpy as np


class SequenceBuilder:
    """Per-user first-occurrence index so that `list.index` (O(len)) is paid once per user."""

    def __init__(self, user_consumed, n_items: int, max_seq_len: int, mode: str = "recent"):
        self.user_consumed = user_consumed
        self.pad = n_items
        self.L = max_seq_len
        self.mode = mode
        self._first = {}

    def _positions(self, u):
        m = self._first.get(u)
        if m is None:
            m = {}
            for p, it in enumerate(self.user_consumed[u]):
                m.setdefault(it, p)
            self._first[u] = m
        return m

    def training_seqs(self, users, items, np_rng=None):
        """Left-aligned window of the <= L items consumed BEFORE `item`; a negative item takes a
        random position `random.randrange(len(history))` (sequence.py:49-55); length >= 1 even
        with empty history (the single key is then the pad id, quirk 5 of SURVEY §8)."""
        B = len(users)
        seqs = np.full((B, self.L), self.pad, dtype=np.int32)
        lens = np.empty(B, dtype=np.int32)
        for j, (u, i) in enumerate(zip(users.tolist() if hasattr(users, "tolist") else users,
                                       items.tolist() if hasattr(items, "tolist") else items)):
            hist = self.user_consumed[u]
            pos = self._positions(u).get(i)
            if pos is None:
                pos = random.randrange(0, len(hist))
            if pos == 0:
                lens[j] = 1
            elif pos < self.L:
                seqs[j, :pos] = hist[:pos]
                lens[j] = pos
            else:
                if self.mode == "recent":
                    seqs[j] = hist[pos - self.L:pos]
                else:
                    seqs[j] = np_rng.choice(hist, self.L, replace=False)
                lens[j] = self.L
        return seqs, lens


def get_interacted_seqs(user_indices, item_indices, user_consumed, pad_index, mode, max_seq_len,
                        user_consumed_set=None, np_rng=None):
    """Functional form with the reference's signature."""
    return SequenceBuilder(user_consumed, pad_index, max_seq_len, mode).training_seqs(
        np.asarray(user_indices), np.asarray(item_indices), np_rng)


def get_recent_seqs(n_users, user_consumed, pad_index, max_seq_len):
    """Most recent <= L items of every user + one all-pad OOV row of length 1 (sequence.py:75-91)."""
    seqs = np.full((n_users + 1, max_seq_len), pad_index, dtype=np.int32)
    lens = np.ones(n_users + 1, dtype=np.int32)
    for u in range(n_users):
        hist = user_consumed[u]
        n = min(len(hist), max_seq_len)
        if n:
            seqs[u, :n] = hist[-n:] if len(hist) >= max_seq_len else hist
        lens[u] = n
    lens[n_users] = 1
    return seqs, lens
