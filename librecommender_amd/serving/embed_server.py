"""In-process embedding server over an exported model directory (`serving.save_embed`).

The reference's embed serving (`libserving/sanic_serving/embed_deploy.py:21-62`) answers `/embed/recommend` by a
faiss inner-product search over the item vectors for `n_rec + len(consumed)` candidates and drops the user's consumed
items.  Here the request handler's compute is one exact `lr_score_topk_f32` launch per batch of users with the
consumed lists filtered inside the kernel; the HTTP / Redis plumbing around it is out of scope."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Sequence

import numpy as np
import torch

from .. import ops


class InvalidUser(KeyError):
    """`embed_deploy.py:27-28`: unknown users are rejected (HTTP 400 in the reference)."""


class EmbedServer:
    def __init__(self, path: str, device="cuda"):
        def load(name):
            with open(os.path.join(path, name)) as f:
                return json.load(f)

        self.device = torch.device(device)
        self.model_name = load("model_name.json")["model_name"]
        self.user2id: Dict[str, int] = load("user2id.json")
        self.id2item: Dict[str, int] = load("id2item.json")
        ue, ie = load("user_embed.json"), load("item_embed.json")
        self.n_users, self.n_items = len(ue), len(ie)
        to_dev = lambda d, n: torch.tensor([d[str(i)] for i in range(n)], dtype=torch.float32, device=self.device)  # noqa: E731
        self.user_embeds, self.item_embeds = to_dev(ue, self.n_users).contiguous(), to_dev(ie, self.n_items).contiguous()
        if self.user_embeds.shape[1] != self.item_embeds.shape[1]:
            raise ValueError("user_embed dimension != item_embed dimension")
        from ..recommendation.recommend import ConsumedIndex
        consumed = ConsumedIndex({int(u): v for u, v in load("user_consumed.json").items()}, self.n_users)
        self._ptr = consumed.ptr                                       # CSR of ascending unique ids (the kernel's contract)
        self._consumed = torch.from_numpy(consumed.items).to(self.device)
        self._item_raw = np.array([self.id2item[str(i)] for i in range(self.n_items)])

    def recommend(self, users: Sequence, n_rec: int) -> Dict[object, List]:
        """{raw user: [raw item ids]}: the `n_rec` highest inner products among the items the user has not consumed
        (fewer when the user consumed almost everything, like the reference's candidate loop)."""
        uids = []
        for u in users:
            if str(u) not in self.user2id:
                raise InvalidUser(f"Invalid user {u} doesn't exist")
            uids.append(self.user2id[str(u)])
        uid = np.asarray(uids, dtype=np.int64)
        k = min(n_rec, self.n_items)
        starts, ends = self._ptr[uid], self._ptr[uid + 1]
        ptr = np.concatenate([[0], np.cumsum(ends - starts)])
        take = np.concatenate([np.arange(a, b) for a, b in zip(starts, ends)] or [np.zeros(0, np.int64)])
        cidx = self._consumed[torch.from_numpy(take).to(self.device)] if len(take) else self._consumed[:0]
        dev = self.device
        U = self.user_embeds.index_select(0, torch.from_numpy(uid).to(dev)).contiguous()
        scores, ids = ops.score_topk(U, self.item_embeds, k, torch.from_numpy(ptr.astype(np.int64)).to(dev),
                                     cidx.to(torch.int32).contiguous(), torch.ones(len(uid), dtype=torch.uint8, device=dev))
        ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        out = {}
        for j, u in enumerate(users):
            keep = np.isfinite(scores[j]) & (ids[j] >= 0)            # slots left empty when < n_rec items remain
            out[u] = self._item_raw[ids[j][keep]].tolist()
        return out
