"""Host negative samplers (`libreco/sampling/negatives.py:17-93`).

Bit-exact target: given the same `numpy.random.Generator` / `random` state these return the same
arrays as the reference (fixtures in tests/golden/negatives.npz), so they necessarily draw from
the generators in the same order: one bulk `choice`, then bounded re-draws of the colliding slots.
"""
from __future__ import annotations

import math
import random
from typing import Optional

import numpy as np


def _collisions(neg: np.ndarray, pos: np.ndarray, also: Optional[np.ndarray]) -> np.ndarray:
    bad = neg == pos
    if also is not None and len(also) > 0:
        bad = bad | (neg == also)
    return np.flatnonzero(bad)


def negatives_from_random(np_rng, n_items, items_pos, num_neg, items=None, tolerance=10):
    """Uniform negatives; without replacement across the batch while the batch is smaller than
    the catalog (negatives.py:22-23); up to `tolerance` re-draws of slots equal to the positive."""
    pos = np.repeat(items_pos, num_neg) if num_neg > 1 else np.asarray(items_pos)
    also = np.repeat(items, num_neg) if (num_neg > 1 and items is not None) else items
    neg = np_rng.choice(n_items, size=len(pos), replace=not (len(pos) < n_items))
    for _ in range(tolerance):
        bad = _collisions(neg, pos, also)
        if len(bad) == 0:
            break
        neg[bad] = np_rng.choice(n_items, size=len(bad), replace=True)
    return neg


def negatives_from_popular(np_rng, n_items, items_pos, num_neg, items=None, probs=None):
    """Popularity-weighted negatives, one re-draw of colliding slots (negatives.py:34-43)."""
    pos = np.repeat(items_pos, num_neg) if num_neg > 1 else np.asarray(items_pos)
    also = np.repeat(items, num_neg) if (num_neg > 1 and items is not None) else items
    neg = np_rng.choice(n_items, size=len(pos), replace=True, p=probs)
    bad = _collisions(neg, pos, also)
    if len(bad) > 0:
        neg[bad] = np_rng.choice(n_items, size=len(bad), replace=True, p=probs)
    return neg


def negatives_from_out_batch(np_rng, n_items, items_pos, items, num_neg):
    """Negatives from items absent from the batch (negatives.py:46-52)."""
    n = len(items_pos) * num_neg
    pool = list(set(range(n_items)) - set(items_pos) - set(items))
    if not pool:
        return np_rng.choice(n_items, size=n, replace=True)
    return np_rng.choice(pool, size=n, replace=not (n < len(pool)))


class _ConsumedCSR:
    """Consumed sets (list or dict indexed by user) as ascending item runs, for the C loop."""

    def __init__(self, consumed_sets):
        users = range(len(consumed_sets)) if not isinstance(consumed_sets, dict) else consumed_sets.keys()
        n = (max(users) + 1) if len(consumed_sets) else 0
        counts = np.zeros(n, dtype=np.int64)
        for u in users:
            counts[u] = len(consumed_sets[u])
        self.ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.items = np.zeros(max(int(self.ptr[-1]), 1), dtype=np.int64)
        for u in users:
            self.items[self.ptr[u]:self.ptr[u + 1]] = sorted(consumed_sets[u])
        self.known = counts > 0 if isinstance(consumed_sets, dict) else np.ones(n, dtype=bool)
        self.is_dict = isinstance(consumed_sets, dict)
        self.keys = set(consumed_sets) if self.is_dict else None


_csr_cache = []          # [(consumed_sets object, fingerprint, _ConsumedCSR)], identity-keyed, most recent first


def _consumed_size(consumed_sets):
    """O(1) fingerprint (this runs once per batch): entry count + the lengths of a few probe entries."""
    n = len(consumed_sets)
    if n == 0:
        return (0,)
    if isinstance(consumed_sets, dict):
        it = iter(consumed_sets.values())
        return n, len(next(it)), len(consumed_sets[next(reversed(consumed_sets))])
    return n, len(consumed_sets[0]), len(consumed_sets[n // 2]), len(consumed_sets[-1])


def _consumed_csr(consumed_sets):
    """CSR of the consumed sets, cached per object (two most recent); an entry whose fingerprint
    changed (sets grown in place between fits) is rebuilt."""
    size = _consumed_size(consumed_sets)
    for n, (obj, sz, csr) in enumerate(_csr_cache):
        if obj is consumed_sets:
            if sz == size:
                return csr
            del _csr_cache[n]
            break
    csr = _ConsumedCSR(consumed_sets)
    _csr_cache.insert(0, (consumed_sets, size, csr))
    del _csr_cache[2:]
    return csr


def negatives_from_unconsumed(user_consumed_set, users, items, n_items, num_neg, tolerance=10):
    """Per (user, positive): draw `floor(n_items * random.random())` until it is neither the
    positive, an earlier negative of the pair, nor consumed (<= tolerance tries), then relax the
    consumed condition (<= tolerance tries) — negatives.py:55-82.  Uses Python's `random`.

    The loop below is the definition; when `lib/liblibreco_host.so` is built the same loop runs in C
    on the generator's own state (hostsrc/host_loops.c) — identical draws, identical final state."""
    from .. import _hostlib
    if _hostlib.load() is not None and len(users) > 0:
        csr = _consumed_csr(user_consumed_set)
        u = np.asarray(users, dtype=np.int64)
        covered = u.min() >= 0 and u.max() < len(csr.ptr) - 1 and \
            (not csr.is_dict or all(int(x) in csr.keys for x in np.unique(u).tolist()))
        if covered:
            return _hostlib.negatives_unconsumed(csr.ptr, csr.items, u, np.asarray(items, dtype=np.int64),
                                                 n_items, num_neg, tolerance)
    return _negatives_from_unconsumed_py(user_consumed_set, users, items, n_items, num_neg, tolerance)


def _negatives_from_unconsumed_py(user_consumed_set, users, items, n_items, num_neg, tolerance=10):
    rnd, floor = random.random, math.floor
    out = []
    for u, i in zip(users, items):
        seen = user_consumed_set[u]
        mine = []
        for _ in range(num_neg):
            n = floor(n_items * rnd())
            ok = False
            for _ in range(tolerance):
                if n != i and n not in mine and n not in seen:
                    ok = True
                    break
                n = floor(n_items * rnd())
            if not ok:
                for _ in range(tolerance):
                    if n != i and n not in mine:
                        break
                    n = floor(n_items * rnd())
            mine.append(n)
        out.extend(mine)
    return np.array(out)


def neg_probs_from_frequency(item_consumed, n_items, temperature):
    """Sampling probabilities  freq^temperature / sum  with freq = #distinct users of an item
    (negatives.py:85-93)."""
    freq = np.array([len(set(item_consumed[i])) for i in range(n_items)], dtype=np.float64)
    if temperature != 1.0:
        freq = np.array([pow(f, temperature) for f in freq.tolist()])
    return freq / np.sum(freq)
