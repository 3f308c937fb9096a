"""Device-side batch construction for pointwise training.

What the host `PointwiseCollator` does per batch in numpy (`batch/collators.py:225-274`): repeat
users `num_neg+1` times, interleave each positive with its negatives, labels 1,0,..,0, user-side
feature columns repeated, item-side feature columns looked up from `item_sparse_unique` /
`item_dense_unique` for EVERY row's item, original column order restored — is done here with the
whole training set resident on the device: one `randperm` per epoch, one
`lr_sample_negatives_i32` launch and a handful of gathers per batch, no host work and no PCIe
traffic in the step loop.  Opt-in (`device_sampling=True` on FM / DeepFM): the negatives come from
the counter-based device sampler, not from the reference's numpy/Python RNG streams, so runs are
reproducible per seed but not sample-for-sample identical to the host path.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .batch_unit import PointwiseBatch


def device_loader_supported(model, neg_sampling) -> bool:
    return (bool(getattr(model, "device_sampling", False)) and neg_sampling and model.task == "ranking"
            and getattr(model, "loss_type", None) in ("cross_entropy", "focal")
            and model.sampler in ("random", "unconsumed") and model.model_name in ("FM", "DeepFM"))


class DevicePointwiseLoader:
    def __init__(self, model, data, batch_size, shuffle, seed):
        self.model, self.n, self.bs, self.shuffle = model, len(data), int(batch_size), shuffle
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
            neg = ops.sample_negatives(pos, self.num_neg, self.n_items, seed, users=u,
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
            yield PointwiseBatch(users, items, labels, sparse, dense, None)
