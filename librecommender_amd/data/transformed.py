"""Index-encoded datasets (`libreco/data/transformed.py`)."""
from __future__ import annotations

import random

import numpy as np
from scipy.sparse import csr_matrix

from .consumed import interaction_consumed


class TransformedSet:
    """Training data in inner ids (`data/transformed.py:13-109`)."""

    def __init__(self, user_indices=None, item_indices=None, labels=None, sparse_indices=None,
                 dense_values=None):
        self._user_indices = user_indices
        self._item_indices = item_indices
        self._labels = labels
        self._sparse_indices = sparse_indices
        self._dense_values = dense_values
        self._sparse_interaction = self._build_csr()

    def _build_csr(self):
        """user x item label matrix; for repeated pairs the LAST label wins."""
        u = np.asarray(self._user_indices, dtype=np.int64)
        i = np.asarray(self._item_indices, dtype=np.int64)
        if len(u) == 0:
            return csr_matrix((0, 0), dtype=np.float32)
        width = int(i.max()) + 1
        key = u * width + i
        # keep the last occurrence of each (user, item)
        _, first_in_reversed = np.unique(key[::-1], return_index=True)
        keep = np.sort(len(key) - 1 - first_in_reversed)
        return csr_matrix((np.asarray(self._labels)[keep], (u[keep], i[keep])), dtype=np.float32)

    def __len__(self):
        return len(self._labels)

    def __getitem__(self, index):
        return self._user_indices[index], self._item_indices[index], self._labels[index]

    user_indices = property(lambda self: self._user_indices)
    item_indices = property(lambda self: self._item_indices)
    labels = property(lambda self: self._labels)
    sparse_indices = property(lambda self: self._sparse_indices)
    dense_values = property(lambda self: self._dense_values)
    sparse_interaction = property(lambda self: self._sparse_interaction)


class TransformedEvalSet:
    """Evaluation / test data in inner ids (`data/transformed.py:112-176`)."""

    def __init__(self, user_indices, item_indices, labels):
        self.user_indices = np.asarray(user_indices)
        self.item_indices = np.asarray(item_indices)
        self.labels = np.asarray(labels)
        self.has_sampled = False
        self.positive_consumed = self._positives()

    def _positives(self):
        all_dummy = bool(np.all(self.labels == 0))   # data without a label column
        keep = np.ones(len(self.labels), bool) if all_dummy else self.labels != 0
        out = {}
        for u, i in zip(self.user_indices[keep].tolist(), self.item_indices[keep].tolist()):
            out.setdefault(u, set()).add(i)
        return {u: sorted(s) for u, s in out.items()}

    def build_negatives(self, n_items, num_neg, seed):
        """Eval-time negative sampling: `negatives_from_unconsumed` after `random.seed(seed)`
        (`data/transformed.py:137-169`), positives and negatives interleaved 1,0,0,..."""
        from ..sampling import negatives_from_unconsumed

        random.seed(seed)
        self.has_sampled = True
        consumed, _ = interaction_consumed(self.user_indices, self.item_indices)
        consumed_set = {u: set(v) for u, v in consumed.items()}
        negs = negatives_from_unconsumed(consumed_set, self.user_indices, self.item_indices, n_items, num_neg)
        self.user_indices = np.repeat(self.user_indices, num_neg + 1)
        self.item_indices = np.repeat(self.item_indices, num_neg + 1)
        self.labels = np.zeros_like(self.item_indices, dtype=np.float32)
        self.labels[:: num_neg + 1] = 1.0
        for j in range(num_neg):
            self.item_indices[j + 1:: num_neg + 1] = negs[j::num_neg]

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        return self.user_indices[index], self.item_indices[index], self.labels[index]
