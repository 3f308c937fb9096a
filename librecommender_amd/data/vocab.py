"""Value -> index encoding and the layout of the shared sparse-feature table.

Mirrors the *results* of `libreco/feature/sparse.py:12-211` and `feature/multi_sparse.py`:
every sparse column owns `len(vocab) + 1` consecutive rows of one table (the last one is its
OOV row); the columns of a multi-sparse field share one vocabulary, offset and OOV row.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np


def encode(values, vocab: np.ndarray, allow_unknown: bool) -> np.ndarray:
    """Index of each value in `vocab`; unknown values -> len(vocab) (the OOV slot).  `vocab` is
    sorted after `build_trainset` and "old values, then appended new ones" after a retrain merge
    (feature/update.py:8-15), so the lookup goes through a sorter (feature/sparse.py:39-57)."""
    values = np.asarray(values)
    vocab = np.asarray(vocab)
    if len(vocab) == 0:
        return np.zeros(len(values), dtype=np.int64)
    if values.dtype != vocab.dtype and (values.dtype.kind in "OUS" or vocab.dtype.kind in "OUS"):
        values, vocab = values.astype(object), vocab.astype(object)
    sorter = np.argsort(vocab, kind="stable")
    pos = np.minimum(np.searchsorted(vocab, values, sorter=sorter), len(vocab) - 1)
    idx = sorter[pos]
    found = vocab[idx] == values
    if not allow_unknown and not found.all():
        raise KeyError("value missing from the vocabulary of its (training) column")
    return np.where(found, idx, len(vocab)).astype(np.int64)


@dataclass
class SparseSchema:
    """Columns, vocabularies and row layout of all sparse + multi-sparse features."""

    sparse_cols: List[str] = field(default_factory=list)
    multi_fields: List[List[str]] = field(default_factory=list)   # each field = list of columns
    vocab: Dict[str, np.ndarray] = field(default_factory=dict)     # sparse col -> sorted uniques
    multi_vocab: Dict[str, np.ndarray] = field(default_factory=dict)  # field[0] -> sorted uniques
    pad_val: Dict[str, object] = field(default_factory=dict)       # field[0] -> padding value

    @property
    def all_cols(self) -> List[str]:
        return list(self.sparse_cols) + [c for f in self.multi_fields for c in f]

    def _sizes(self):
        s = [len(self.vocab[c]) + 1 for c in self.sparse_cols]
        m = [len(self.multi_vocab[f[0]]) + 1 for f in self.multi_fields]
        return s, m

    @property
    def offsets(self) -> Optional[np.ndarray]:
        """First table row of every column (multi-sparse columns repeat their field's offset)."""
        if not self.all_cols:
            return None
        s, m = self._sizes()
        starts = np.concatenate([[0], np.cumsum(s + m)])[:-1]
        out = list(starts[: len(s)])
        for f, st in zip(self.multi_fields, starts[len(s):]):
            out.extend([st] * len(f))
        return np.asarray(out, dtype=np.int64)

    @property
    def oov_rows(self) -> Optional[np.ndarray]:
        """OOV row of every column."""
        if not self.all_cols:
            return None
        s, m = self._sizes()
        ends = np.cumsum(s + m) - 1
        out = list(ends[: len(s)])
        for f, e in zip(self.multi_fields, ends[len(s):]):
            out.extend([e] * len(f))
        return np.asarray(out, dtype=np.int64)

    @property
    def field_oov_rows(self) -> np.ndarray:
        s, m = self._sizes()
        return (np.cumsum(s + m) - 1)[len(s):]

    @property
    def n_rows(self) -> int:
        s, m = self._sizes()
        return int(sum(s) + sum(m))

    def encode_frame(self, df, is_train: bool) -> Optional[np.ndarray]:
        """int32 [n, n_cols] global sparse rows (tfops/features.py indexes one shared table)."""
        cols = self.all_cols
        if not cols:
            return None
        out = np.empty((len(df), len(cols)), dtype=np.int32)
        j = 0
        for c in self.sparse_cols:
            out[:, j] = encode(df[c].to_numpy(), self.vocab[c], allow_unknown=not is_train)
            j += 1
        for f in self.multi_fields:
            for c in f:   # padding values are not in the vocabulary -> the field's OOV slot
                out[:, j] = encode(df[c].to_numpy(), self.multi_vocab[f[0]], allow_unknown=True)
                j += 1
        return out + self.offsets.astype(np.int32)


def last_seen_rows(ids: np.ndarray, matrix: np.ndarray, cols: Sequence[int], n_ids: int) -> np.ndarray:
    """Feature row of the LAST occurrence of every id 0..n_ids-1 (feature/unique.py:4-58: stable
    sort + last-of-run).  `matrix` is [n, n_features]; returns [n_ids, len(cols)]."""
    sub = matrix[:, list(cols)] if matrix.ndim == 2 else matrix.reshape(-1, 1)
    order = np.argsort(ids, kind="stable")
    sorted_ids = ids[order]
    last = np.r_[sorted_ids[1:] != sorted_ids[:-1], True]
    picked = order[last]
    if len(picked) != n_ids:
        raise AssertionError("every id must appear at least once in the training data")
    return sub[picked]
