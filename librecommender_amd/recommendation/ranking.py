"""`rank_recommendations` (`libreco/recommendation/ranking.py:10-56`) for callers that hold a [B, n_items] block of model
predictions: the consumed filter, the top-`n_rec` selection and the final ordering run on the device (one scatter of the
batch's consumed ids + one `topk`) instead of the reference's per-user numpy loop.  The model classes do not go through
here — embed models fuse scoring and top-k (`recommend.py:recommend_from_embedding` -> `lr_score_topk_f32`), feature models
rank the blocks their catalog scorer produces (`bases/feat_base.py`) — it is the seam kept for code written against the
reference's function."""
from __future__ import annotations

import numpy as np
import torch

from .recommend import random_select_device


def rank_recommendations(task, user_ids, model_preds, n_rec, n_items, user_consumed, filter_consumed=True,
                         random_rec=False, return_scores=False, device="cuda"):
    """-> ids [B, n_rec] (numpy, best first) and, with `return_scores`, their scores (`expit` of them for ranking tasks).

    A user's history is filtered only if `n_rec + len(history) <= n_items` (`ranking.py:38`); `random_rec` draws
    `n_rec` items without replacement with weights softmax(pred)^0.75 + 1e-8 (unseeded in the reference too)."""
    if n_rec > n_items:
        raise ValueError(f"`n_rec` {n_rec} exceeds num of items {n_items}")
    from ..bases.base import hip_device

    dev = model_preds.device if isinstance(model_preds, torch.Tensor) and model_preds.is_cuda else hip_device(device)
    preds = torch.as_tensor(np.asarray(model_preds) if not isinstance(model_preds, torch.Tensor) else model_preds)
    preds = preds.to(dev)
    if preds.ndim == 1:
        assert preds.numel() % n_items == 0
        preds = preds.view(-1, n_items)
    B = preds.shape[0]
    rows, cols = [], []
    for i in range(B):
        consumed = user_consumed[user_ids[i]] if user_ids[i] in user_consumed else []
        if filter_consumed and len(consumed) and n_rec + len(consumed) <= n_items:
            c = np.unique(np.asarray(consumed, dtype=np.int64))
            rows.append(np.full(len(c), i, dtype=np.int64))
            cols.append(c)
    banned = None
    if rows:
        banned = torch.zeros((B, n_items), dtype=torch.bool, device=dev)
        banned[torch.from_numpy(np.concatenate(rows)).to(dev), torch.from_numpy(np.concatenate(cols)).to(dev)] = True
    if random_rec:
        ids = random_select_device(preds, banned, n_rec)
    else:
        masked = preds if banned is None else preds.masked_fill(banned, float("-inf"))
        ids = torch.topk(masked, n_rec, dim=1, sorted=True).indices
    if not return_scores:
        return ids.cpu().numpy()
    scores = torch.gather(preds, 1, ids)
    if task == "ranking":
        scores = torch.sigmoid(scores)
    return ids.cpu().numpy(), scores.cpu().numpy()
