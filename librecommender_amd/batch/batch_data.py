"""Batch iteration (`libreco/batch/batch_data.py:46-105`): the index order is the one torch's CPU
`RandomSampler` + `BatchSampler` produce for a given seed (so rows are visited in the reference's
order), the collator turns index slices into batches."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader, Sampler

from .. import _hostlib
from .collators import BaseCollator, PairwiseCollator, PointwiseCollator


class BatchData(torch.utils.data.Dataset):
    def __init__(self, data, use_features):
        self.user_indices, self.item_indices, self.labels = data.user_indices, data.item_indices, data.labels
        self.sparse_indices = data.sparse_indices if use_features else None
        self.dense_values = data.dense_values if use_features else None

    def __getitem__(self, idx):
        out = {"user": self.user_indices[idx], "item": self.item_indices[idx], "label": self.labels[idx]}
        if self.sparse_indices is not None:
            out["sparse"] = _hostlib.gather_rows(self.sparse_indices, idx)
        if self.dense_values is not None:
            out["dense"] = _hostlib.gather_rows(self.dense_values, idx)
        return out

    def __len__(self):
        return len(self.labels)


class ShuffledBatches(Sampler):
    """Index arrays of `BatchSampler(RandomSampler(ds) | SequentialSampler(ds), batch_size, False)`
    without the per-index Python iteration: every pass draws the permutation seed from torch's
    global CPU generator exactly like `RandomSampler.__iter__`, then slices one `randperm`."""

    def __init__(self, n, batch_size, shuffle):
        self.n, self.batch_size, self.shuffle = int(n), int(batch_size), bool(shuffle)

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        if self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
            order = torch.randperm(self.n, generator=gen).numpy()
        else:
            order = np.arange(self.n)
        for s in range(0, self.n, self.batch_size):
            yield order[s:s + self.batch_size]


def get_collate_fn(model, neg_sampling):
    """Collator choice of `batch_data.py:67-90`."""
    info = model.data_info
    sep = model.model_name == "TwoTower"
    if model.model_name == "YouTubeRetrieval" or (model.model_name == "TwoTower" and model.loss_type == "softmax"):
        return BaseCollator(model, info, sep)    # listwise training: positives only (+ history windows)
    if model.task == "rating" or not neg_sampling:
        return BaseCollator(model, info, sep)
    if model.loss_type in ("cross_entropy", "focal"):
        return PointwiseCollator(model, info, sep)
    return PairwiseCollator(model, info, repeat_positives=model.graph_backend == "tf")


def get_batch_loader(model, data, neg_sampling, batch_size, shuffle, num_workers=0, seed=42):
    from .device_loader import DevicePointwiseLoader, device_loader_mode
    mode = device_loader_mode(model, neg_sampling)
    if mode is not None:                                 # opt-in device-side sampling + collation
        return DevicePointwiseLoader(model, data, batch_size, shuffle, seed, mode=mode)
    torch.manual_seed(seed)
    ds = BatchData(data, use_features=model.uses_features)
    return DataLoader(ds, batch_size=None, sampler=ShuffledBatches(len(ds), batch_size, shuffle),
                      collate_fn=get_collate_fn(model, neg_sampling), num_workers=num_workers)


def adjust_batch_size(model, original_batch_size):
    """Post-sampling batch ~= the user's batch_size (`batch_data.py:93-105`)."""
    if model.model_name == "YouTubeRetrieval" or (model.model_name == "TwoTower" and model.loss_type == "softmax"):
        return original_batch_size   # listwise training
    if model.sampler is not None:
        if model.loss_type in ("cross_entropy", "focal"):
            return max(1, int(original_batch_size / (model.num_neg + 1)))
        return max(1, int(original_batch_size / model.num_neg))
    return original_batch_size
