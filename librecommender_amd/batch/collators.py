"""Batch collators: negative sampling + feature/sequence assembly on the host
(`libreco/batch/collators.py:33-319`).  This is the parity ("bit-exact index work") path; the
RNG protocol — `seed % 3407 * 11` seeding `random`, torch and a numpy Generator on first use — is
the reference's (`collators.py:180-187`)."""
from __future__ import annotations

import random
from typing import Optional

import numpy as np
import torch

from ..sampling import (
    neg_probs_from_frequency,
    negatives_from_popular,
    negatives_from_random,
    negatives_from_unconsumed,
)
from .batch_unit import (
    PairFeats,
    PairwiseBatch,
    PointwiseBatch,
    PointwiseSepFeatBatch,
    SeqFeats,
    TripleFeats,
)
from .. import _hostlib
from .sequence import SequenceBuilder


def _col_index(cols):
    """Column selector: a slice when the columns are one consecutive ascending run (a view instead of a fancy-index
    copy), the list otherwise."""
    cols = list(cols)
    if cols and all(b - a == 1 for a, b in zip(cols[:-1], cols[1:])):
        return slice(cols[0], cols[-1] + 1)
    return cols


def _take_cols(mat, cols):
    return mat[:, _col_index(cols)] if cols else None


def restore_column_order(user_part, item_part, user_cols, item_cols):
    """Concatenate user and item feature blocks back into the original column order
    (`merge_columns`, collators.py:480-490)."""
    if len(user_part) != len(item_part):
        raise ValueError(f"length of user_features and length of item_features don't match, "
                         f"got {len(user_part)} and {len(item_part)}")
    order = np.argsort(np.asarray(list(user_cols) + list(item_cols)))
    merged = np.concatenate([user_part, item_part], axis=1)
    if np.array_equal(order, np.arange(len(order))):      # user columns first, in order: nothing to permute
        return merged
    return merged[:, order]


class BaseCollator:
    """No sampling: forwards the batch (rating task, `neg_sampling=False`, in-batch softmax)."""

    def __init__(self, model, data_info, separate_features=False, temperature=0.75):
        d = data_info
        self.n_users, self.n_items = d.n_users, d.n_items
        self.user_consumed, self.item_consumed = d.user_consumed, d.item_consumed
        self.cols = {"sparse": (d.user_sparse_col.index, d.item_sparse_col.index),
                     "dense": (d.user_dense_col.index, d.item_dense_col.index)}
        self.item_unique = {"sparse": d.item_sparse_unique, "dense": d.item_dense_unique}
        self.has_seq = hasattr(model, "max_seq_len") and getattr(model, "uses_sequence", False)
        self.seq_mode = getattr(model, "seq_mode", None)
        self.max_seq_len = getattr(model, "max_seq_len", None)
        self.dual_seq = (model.long_max_len, model.short_max_len) if hasattr(model, "long_max_len") else None
        self.separate_features = separate_features
        self.seed = model.seed
        self.temperature = temperature
        self.user_consumed_set = None
        self.neg_probs = None
        self.np_rng = None
        self._seq_builder: Optional[SequenceBuilder] = None

    # ---- lazily created state --------------------------------------------------------------
    def _ensure_rng(self):
        if self.np_rng is None:
            info = torch.utils.data.get_worker_info()
            seed = (self.seed if info is None else info.seed) % 3407 * 11
            random.seed(seed)
            torch.manual_seed(seed)
            self.np_rng = np.random.default_rng(seed)

    def _ensure_consumed_sets(self):
        if self.user_consumed_set is None:
            self.user_consumed_set = [set(self.user_consumed[u]) for u in range(self.n_users)]

    # ---- pieces ------------------------------------------------------------------------------
    def features(self, batch, kind):
        if kind not in batch:
            return None
        f = batch[kind]
        if self.separate_features:
            u_cols, i_cols = self.cols[kind]
            return PairFeats(_take_cols(f, u_cols), _take_cols(f, i_cols))
        return f

    def seqs(self, users, items):
        if not self.has_seq:
            return None
        self._ensure_rng()
        if self._seq_builder is None:
            self._seq_builder = SequenceBuilder(self.user_consumed, self.n_items, self.max_seq_len, self.seq_mode)
        if self.dual_seq is not None:     # SIM: [long | short] windows side by side, lengths [B, 2] (collators.py:114-127)
            lg, ln, sh, sn = self._seq_builder.training_dual_seqs(np.asarray(users), np.asarray(items), *self.dual_seq)
            return SeqFeats(np.concatenate([lg, sh], axis=1), np.stack([ln, sn], axis=1))
        s, n = self._seq_builder.training_seqs(np.asarray(users), np.asarray(items), self.np_rng)
        return SeqFeats(s, n)

    def sample_neg_items(self, batch, sampler, num_neg):
        if sampler == "unconsumed":
            self._ensure_consumed_sets()
            return negatives_from_unconsumed(self.user_consumed_set, batch["user"], batch["item"],
                                             self.n_items, num_neg)
        self._ensure_rng()
        if sampler == "popular":
            if self.neg_probs is None:
                self.neg_probs = neg_probs_from_frequency(self.item_consumed, self.n_items, self.temperature)
            return negatives_from_popular(self.np_rng, self.n_items, batch["item"], num_neg, probs=self.neg_probs)
        return negatives_from_random(self.np_rng, self.n_items, batch["item"], num_neg)

    def __call__(self, batch):
        cls = PointwiseSepFeatBatch if self.separate_features else PointwiseBatch
        return cls(batch["user"], batch["item"], batch["label"], self.features(batch, "sparse"),
                   self.features(batch, "dense"), self.seqs(batch["user"], batch["item"]))


class PointwiseCollator(BaseCollator):
    """Each positive followed by its `num_neg` negatives, labels 1,0,..,0 (collators.py:225-274)."""

    def __init__(self, model, data_info, separate_features=False):
        super().__init__(model, data_info, separate_features)
        self.sampler, self.num_neg = model.sampler, model.num_neg

    def _item_rows_as(self, kind, dtype):
        """The per-item feature rows in the batch's own dtype (the stored table may be int64 next to int32 batches):
        a contiguous copy made once per collator."""
        cache = self.__dict__.setdefault("_item_rows_cache", {})
        key = (kind, np.dtype(dtype).str)
        if key not in cache:
            cache[key] = np.ascontiguousarray(self.item_unique[kind], dtype=dtype)
        return cache[key]

    def _feats(self, batch, kind, items):
        if kind not in batch:
            return None
        u_cols, i_cols = self.cols[kind]
        if u_cols and i_cols and not self.separate_features and \
                sorted(list(u_cols) + list(i_cols)) == list(range(batch[kind].shape[1])):
            # one C pass: the positive's row repeated, item columns overwritten by the sampled item's stored features
            merged = _hostlib.merge_pointwise(batch[kind], self._item_rows_as(kind, batch[kind].dtype), i_cols, items,
                                              self.num_neg + 1)
            if merged is not None:
                return merged
        u_part = np.repeat(batch[kind][:, _col_index(u_cols)], self.num_neg + 1, axis=0) if u_cols else None
        i_part = self.item_unique[kind][items] if i_cols else None   # features of the sampled items
        if self.separate_features:
            return PairFeats(u_part, i_part)
        if u_cols and i_cols:
            return restore_column_order(u_part, i_part, u_cols, i_cols)
        return u_part if u_cols else i_part

    def __call__(self, batch):
        k = self.num_neg + 1
        users = np.repeat(batch["user"], k)
        items = np.repeat(batch["item"], k)
        labels = np.zeros_like(items, dtype=np.float32)
        labels[::k] = 1.0
        negs = self.sample_neg_items(batch, self.sampler, self.num_neg)
        for j in range(self.num_neg):
            items[j + 1::k] = negs[j::self.num_neg]
        cls = PointwiseSepFeatBatch if self.separate_features else PointwiseBatch
        return cls(users, items, labels, self._feats(batch, "sparse", items),
                   self._feats(batch, "dense", items), self.seqs(users, items))


class PairwiseCollator(BaseCollator):
    """(query, positive, negative) triples; positives are repeated per negative for the graph
    models of the reference's TF backend and not for its torch backend (batch_data.py:87-88)."""

    def __init__(self, model, data_info, repeat_positives):
        super().__init__(model, data_info, separate_features=True)
        self.sampler, self.num_neg = model.sampler, model.num_neg
        self.repeat_positives = repeat_positives

    def _feats(self, batch, kind, negs):
        if kind not in batch:
            return None
        u_cols, i_cols = self.cols[kind]
        rep = self.num_neg if (self.repeat_positives and self.num_neg > 1) else 1
        f = batch[kind]
        q = np.repeat(f[:, u_cols], rep, axis=0) if u_cols else None
        p = np.repeat(f[:, i_cols], rep, axis=0) if i_cols else None
        n = self.item_unique[kind][negs] if i_cols else None
        return TripleFeats(q, p, n)

    def __call__(self, batch):
        rep = self.repeat_positives and self.num_neg > 1
        users = np.repeat(batch["user"], self.num_neg) if rep else batch["user"]
        pos = np.repeat(batch["item"], self.num_neg) if rep else batch["item"]
        negs = self.sample_neg_items(batch, self.sampler, self.num_neg)
        seqs = self.seqs(users, pos)
        if self.has_seq and not self.repeat_positives and self.num_neg > 1:
            seqs = seqs.repeat(self.num_neg)
        return PairwiseBatch(users, (pos, negs), self._feats(batch, "sparse", negs),
                             self._feats(batch, "dense", negs), seqs)
