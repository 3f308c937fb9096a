"""Device-side batch construction for pointwise training, including the
behaviour sequences of DIN (`batch/sequence.py:33-72`, "recent" mode).

What the host `PointwiseCollator` does per batch in numpy (`batch/collators.py:225-274`): repeat
users `num_neg+1` times, interleave each positive with its negatives, labels 1,0,..,0, user-side
feature columns repeated, item-side feature columns looked up from `item_sparse_unique` /
`item_dense_unique` for EVERY row's item, original column order restored — is done here with the
whole training set resident on the device: one `randperm` per epoch, one
`lr_sample_negatives_i32` launch and a handful of gathers per batch, no host work and no PCIe
traffic in the step loop.  Opt-in (`device_sampling=True` on FM / DeepFM): the negatives come from
the counter-based device sampler, not from the reference's numpy/Python RNG streams, so runs are
reproducible per seed but not sample-for-sample identical to the host path.

Sequences (`DeviceSequences`): the histories live on the device as one CSR plus the sorted
`(user, item) -> first position` table of `batch.sequence.SequenceBuilder`; a batch is one
`searchsorted`, one uniform draw for the rows whose item is not in the user's history (the
reference's `random.randrange(len(history))`) and one windowed gather.  All of it is device-agnostic
tensor code, so it is checked row for row against the host builder on CPU tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .batch_unit import PointwiseBatch, SeqFeats
from .sequence import SequenceBuilder


def device_loader_supported(model, neg_sampling) -> bool:
    return (bool(getattr(model, "device_sampling", False)) and neg_sampling and model.task == "ranking"
            and getattr(model, "loss_type", None) in ("cross_entropy", "focal")
            and model.sampler in ("random", "unconsumed")
            and (model.model_name in ("FM", "DeepFM")
                 or (model.model_name == "DIN" and getattr(model, "seq_mode", None) == "recent")))


class DeviceSequences:
    """`SequenceBuilder.training_seqs` ("recent" windows) on device tensors."""

    def __init__(self, user_consumed, n_items, max_seq_len, device):
        host = SequenceBuilder(user_consumed, n_items, max_seq_len, "recent")
        to = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), device=device).to(dt)  # noqa: E731
        self.ptr, self.counts = to(host.ptr, torch.int64), to(host.counts, torch.int64)
        self.hist = to(host.hist, torch.int32)                  # last slot: scratch for masked reads
        self.keys, self.first_pos = to(host.keys, torch.int64), to(host.first_pos, torch.int64)
        self.stride, self.pad, self.L, self.device = host.stride, int(n_items), int(max_seq_len), device
        self.t = torch.arange(self.L, device=device, dtype=torch.int64).view(1, -1)

    def positions(self, users, items):
        """First index of `item` in the user's history, -1 when absent."""
        key = users.long() * self.stride + items.long()
        if self.keys.numel() == 0:
            return torch.full_like(key, -1)
        j = torch.searchsorted(self.keys, key).clamp_(max=self.keys.numel() - 1)
        return torch.where(self.keys[j] == key, self.first_pos[j], torch.full_like(key, -1))

    def build(self, users, items, generator=None):
        users = users.long()
        pos = self.positions(users, items)
        n_hist = self.counts[users]                   # >= 1: checked once by the loader
        draw = (torch.rand(pos.numel(), device=self.device, generator=generator, dtype=torch.float64)
                * n_hist.double()).long().clamp_(max=n_hist - 1)
        pos = torch.where(pos < 0, draw, pos)
        start, length = (pos - self.L).clamp_(min=0), pos.clamp(max=self.L)
        valid = self.t < length.view(-1, 1)
        src = torch.where(valid, (self.ptr[users] + start).view(-1, 1) + self.t,
                          torch.full_like(self.t, self.hist.numel() - 1))
        seqs = torch.where(valid, self.hist[src], torch.full_like(src, self.pad, dtype=torch.int32))
        return seqs.contiguous(), length.clamp(min=1).to(torch.int32)


class DevicePointwiseLoader:
    def __init__(self, model, data, batch_size, shuffle, seed, sample_negatives=None):
        self.model, self.n, self.bs, self.shuffle = model, len(data), int(batch_size), shuffle
        self.sample_negatives = sample_negatives or ops.sample_negatives      # tests inject the oracle sampler
        dev = self.dev = model.device
        info = model.data_info
        self.num_neg, self.n_items = int(model.num_neg), int(model.n_items)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev).to(torch.int32)  # noqa: E731
        self.users, self.items = i32(data.user_indices), i32(data.item_indices)
        feats = model.uses_features
        self.sparse = i32(data.sparse_indices) if feats and data.sparse_indices is not None else None
        self.dense = (torch.as_tensor(np.ascontiguousarray(data.dense_values), device=dev, dtype=torch.float32)
                      if feats and data.dense_values is not None else None)
        idx = lambda c: torch.as_tensor(np.asarray(c, dtype=np.int64), device=dev)  # noqa: E731
        self.i_sp_cols = idx(info.item_sparse_col.index) if self.sparse is not None and info.item_sparse_col.index else None
        self.i_dn_cols = idx(info.item_dense_col.index) if self.dense is not None and info.item_dense_col.index else None
        self.item_sparse = i32(info.item_sparse_unique) if self.i_sp_cols is not None else None
        self.item_dense = (torch.as_tensor(info.item_dense_unique, device=dev, dtype=torch.float32)
                           if self.i_dn_cols is not None else None)
        self.cptr = self.cidx = None
        if model.sampler == "unconsumed":
            ptr = np.zeros(model.n_users + 1, dtype=np.int64)
            flat = []
            for u in range(model.n_users):
                c = np.unique(np.asarray(info.user_consumed.get(u, ()), dtype=np.int64))
                flat.append(c)
                ptr[u + 1] = ptr[u] + len(c)
            self.cptr = torch.from_numpy(ptr).to(dev)
            self.cidx = torch.from_numpy(np.concatenate(flat) if flat else np.zeros(0, np.int64)).to(dev).to(torch.int32)
        self.seqs = None
        if getattr(model, "uses_sequence", False):
            self.seqs = DeviceSequences(info.user_consumed, self.n_items, model.max_seq_len, dev)
            lens = np.asarray([len(info.user_consumed.get(int(u), ())) for u in np.unique(data.user_indices)])
            if len(lens) and lens.min() <= 0:
                raise ValueError("empty range for randrange()")      # a user without history (sequence.py:52)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self.seed, self.epoch = int(seed), 0

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        dev, k = self.dev, self.num_neg + 1
        self.epoch += 1
        order = (torch.randperm(self.n, device=dev, generator=self.gen) if self.shuffle
                 else torch.arange(self.n, device=dev))
        for bi, s in enumerate(range(0, self.n, self.bs)):
            rows = order[s:s + self.bs]
            u, pos = self.users[rows].contiguous(), self.items[rows].contiguous()
            seed = (self.seed * 0x9E3779B1 + self.epoch * 1_000_003 + bi) & ((1 << 63) - 1)
            neg = self.sample_negatives(pos, self.num_neg, self.n_items, seed, users=u,
                                        consumed_ptr=self.cptr, consumed_idx=self.cidx)
            items = torch.cat([pos.view(-1, 1), neg.view(-1, self.num_neg)], dim=1).reshape(-1)   # pos,neg1..negk
            users = u.repeat_interleave(k)
            labels = torch.zeros(items.numel(), dtype=torch.float32, device=dev)
            labels[::k] = 1.0
            sparse = dense = None
            if self.sparse is not None:
                sparse = self.sparse[rows].repeat_interleave(k, dim=0)
                if self.i_sp_cols is not None:                     # every row's item side from the item table
                    sparse[:, self.i_sp_cols] = self.item_sparse[items.long()]
            if self.dense is not None:
                dense = self.dense[rows].repeat_interleave(k, dim=0)
                if self.i_dn_cols is not None:
                    dense[:, self.i_dn_cols] = self.item_dense[items.long()]
            seqs = SeqFeats(*self.seqs.build(users, items, self.gen)) if self.seqs is not None else None
            yield PointwiseBatch(users, items, labels, sparse, dense, seqs)
