"""Batch wire format between collators and trainers (`libreco/batch/batch_unit.py:13-185`).
Batches stay numpy on the host; the nets move what they consume to the HIP device
(`utils/device.py:to_device`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class SeqFeats:
    interacted_seq: np.ndarray   # int32 [B, L]
    interacted_len: np.ndarray   # int32 [B]

    def repeat(self, num):
        self.interacted_seq = np.repeat(self.interacted_seq, num, axis=0)
        self.interacted_len = np.repeat(self.interacted_len, num)
        return self


@dataclass
class PairFeats:
    user_feats: Optional[np.ndarray]
    item_feats: Optional[np.ndarray]


@dataclass
class TripleFeats:
    query_feats: Optional[np.ndarray]
    item_pos_feats: Optional[np.ndarray]
    item_neg_feats: Optional[np.ndarray]


@dataclass
class PointwiseBatch:
    users: np.ndarray
    items: np.ndarray
    labels: np.ndarray
    sparse_indices: Optional[np.ndarray] = None
    dense_values: Optional[np.ndarray] = None
    seqs: Optional[SeqFeats] = None


@dataclass
class PointwiseSepFeatBatch:
    """TwoTower: user / item features kept apart (`PairFeats`)."""
    users: np.ndarray
    items: np.ndarray
    labels: np.ndarray
    sparse_indices: Optional[PairFeats] = None
    dense_values: Optional[PairFeats] = None
    seqs: Optional[SeqFeats] = None


@dataclass
class PairwiseBatch:
    queries: np.ndarray
    item_pairs: Tuple[np.ndarray, np.ndarray]
    sparse_indices: Optional[TripleFeats] = None
    dense_values: Optional[TripleFeats] = None
    seqs: Optional[SeqFeats] = None
