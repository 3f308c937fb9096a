"""Full-catalog recommendation on device (`libreco/recommendation/recommend.py:57-78` +
`ranking.py:10-56`): one fused `lr_score_topk_f32` launch instead of a B x N numpy GEMM and a
per-user Python ranking loop."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


class ConsumedIndex:
    """Per-user consumed items as one CSR of ascending unique ids + the history length the reference
    uses in its "can we filter" test (`ranking.py:38`: `n_rec + len(consumed) <= n_items`, on the
    raw history, repeats included)."""

    def __init__(self, user_consumed, n_users):
        us, its = [], []
        self.n_users = n_users
        self.hist_len = np.zeros(n_users + 1, dtype=np.int64)
        for u, items in user_consumed.items():
            if 0 <= u < n_users:
                self.hist_len[u] = len(items)
                us.append(np.full(len(items), u, dtype=np.int64))
                its.append(np.asarray(items, dtype=np.int64))
        us = np.concatenate(us) if us else np.zeros(0, np.int64)
        its = np.concatenate(its) if its else np.zeros(0, np.int64)
        stride = int(its.max(initial=0)) + 1
        uniq = np.unique(us * stride + its)                       # by user, then ascending item
        self.items = (uniq % stride).astype(np.int32)
        self.ptr = np.concatenate([[0], np.cumsum(np.bincount(uniq // stride, minlength=n_users + 1))]).astype(np.int64)

    def consumed(self, u):
        """Ascending unique items of one user (None: unknown user / empty history)."""
        u = int(u)
        if not 0 <= u < self.n_users or self.hist_len[u] == 0:
            return None
        return self.items[self.ptr[u]:self.ptr[u + 1]]

    def batch_csr(self, user_ids, n_rec, n_items, filter_consumed, device):
        """(ptr int64 [B+1], ids int32, flag uint8 [B]) of the users whose history is filtered."""
        u = np.asarray(user_ids, dtype=np.int64)
        known = (u >= 0) & (u <= self.n_users)
        uu = np.where(known, u, self.n_users)
        n_hist = self.hist_len[uu]
        flags = known & bool(filter_consumed) & (n_hist > 0) & (n_rec + n_hist <= n_items)
        lens = np.where(flags, self.ptr[uu + 1] - self.ptr[uu], 0)
        ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(ptr[-1])
        if total:
            idx = self.items[np.repeat(self.ptr[uu] - ptr[:-1], lens) + np.arange(total, dtype=np.int64)]
        else:
            idx = np.zeros(1, dtype=np.int32)
        to = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
        return to(ptr), to(idx.astype(np.int32)), to(flags.astype(np.uint8))


def construct_rec(data_info, user_ids, computed_recs, inner_id):
    """Inner ids -> raw ids (`recommend.py:8-19`)."""
    out = {}
    for j, u in enumerate(user_ids):
        if inner_id:
            out[u] = np.asarray(computed_recs[j])
        else:
            out[data_info.id2user[u]] = np.array([data_info.id2item[r] for r in np.asarray(computed_recs[j]).tolist()])
    return out


def check_dynamic_rec_feats(model_name, user, user_feats, seq, sequence_models=("DIN", "YouTubeRanking", "YouTubeRetrieval", "Transformer", "SIM")):
    if seq is not None and model_name not in sequence_models:
        raise ValueError(f"`{model_name}` doesn't support arbitrary seq inference.")
    if not np.isscalar(user):
        if user_feats is not None:
            raise ValueError(f"Batch inference doesn't support assigning arbitrary features: {user}")
        if seq is not None:
            raise ValueError(f"Batch inference doesn't support arbitrary item sequence: {user}")
    if seq is not None and not isinstance(seq, (list, np.ndarray)):
        raise ValueError("`seq` must be list or numpy.ndarray.")
    if user_feats is not None and not isinstance(user_feats, dict):
        raise ValueError("`user_feats` must be `dict`.")


def random_select_device(scores: torch.Tensor, banned_mask, n_rec: int) -> torch.Tensor:
    """`random_rec=True` (`ranking.py:65-73`): sample n_rec items without replacement with
    probability softmax(score)^0.75 (+1e-8), then order them by score.  Unseeded by design."""
    p = torch.softmax(scores.double(), dim=1).pow(0.75) + 1e-8
    if banned_mask is not None:
        p = p.masked_fill(banned_mask, 0.0)
    picks = torch.multinomial(p, n_rec, replacement=False)
    order = torch.argsort(torch.gather(scores, 1, picks), dim=1, descending=True)
    return torch.gather(picks, 1, order)


def random_select_streaming(U: torch.Tensor, I: torch.Tensor, ptr, cidx, flag, n_rec: int,
                            max_elems: int = 1 << 27) -> torch.Tensor:
    """`random_rec=True` without the [B, N] score matrix (400 GB at cfg 4): the same distribution as
    `random_select_device` — n_rec items without replacement with weights w = softmax(score)^0.75 + 1e-8, consumed
    items excluded — drawn by the Gumbel-top-k identity: the n_rec largest of log w + G (G ~ Gumbel(0,1) i.i.d.)
    are a sample without replacement with probabilities proportional to w (sequential re-normalised draws, i.e.
    what `Generator.choice(replace=False, p=...)` of `ranking.py:65-73` does).  Two passes over item chunks of at
    most `max_elems` scores: the softmax normaliser, then a running top-n_rec of the perturbed keys."""
    m, ssum = random_rec_normaliser(U, I, max_elems)
    lse = m + torch.log(ssum)
    best_k, best_i, best_s = random_rec_local_topk(U, I, lse, ptr, cidx, flag, n_rec, 0, max_elems)
    order = torch.argsort(best_s, dim=1, descending=True)
    return torch.gather(best_i, 1, order)


def random_rec_normaliser(U: torch.Tensor, I: torch.Tensor, max_elems: int = 1 << 27):
    """Pass 1 of the streaming draw over ONE block of items: per user (running max, sum of exp(score - max)) in fp64.
    Blocks combine like online-softmax states (`ShardedItemEmbeds.random_topk` combines the ranks' blocks)."""
    dev, (B, N) = U.device, (U.shape[0], I.shape[0])
    chunk = max(1024, min(max(N, 1), max_elems // max(B, 1)))
    m = torch.full((B,), -float("inf"), dtype=torch.float64, device=dev)
    ssum = torch.zeros(B, dtype=torch.float64, device=dev)
    for s0 in range(0, N, chunk):
        sc = (U @ I[s0:s0 + chunk].T).double()
        mn = torch.maximum(m, sc.max(dim=1).values)
        ssum = ssum * torch.exp(m - mn) + torch.exp(sc - mn[:, None]).sum(dim=1)
        m = mn
    return m, ssum


def random_rec_local_topk(U, I, lse, ptr, cidx, flag, n_rec: int, item_base: int = 0, max_elems: int = 1 << 27, generator=None):
    """Pass 2 over ONE block of items (global ids item_base ... item_base + len(I) - 1): the n_rec largest keys
    log w + Gumbel with w = exp(0.75 (score - lse)) + 1e-8, consumed (global) ids excluded.
    -> (keys [B, n_rec] fp64, global ids, scores); fewer than n_rec candidates leave keys of -inf."""
    dev, (B, N) = U.device, (U.shape[0], I.shape[0])
    chunk = max(1024, min(max(N, 1), max_elems // max(B, 1)))
    n_cons = (ptr[1:] - ptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(B, device=dev), n_cons)
    cons = cidx[: rows.numel()].long() - int(item_base)
    banned_rows = flag.bool()[rows] if flag is not None else torch.ones_like(rows, dtype=torch.bool)
    best_k = torch.full((B, n_rec), -float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.zeros((B, n_rec), dtype=torch.int64, device=dev)
    best_s = torch.zeros((B, n_rec), dtype=U.dtype, device=dev)
    for s0 in range(0, N, chunk):                                # running top-n_rec of log w + Gumbel
        sc = U @ I[s0:s0 + chunk].T
        w = torch.exp(0.75 * (sc.double() - lse[:, None])) + 1e-8
        gum = -torch.log(-torch.log(torch.rand(sc.shape, dtype=torch.float64, device=dev, generator=generator).clamp_(1e-300, 1.0)))
        key = torch.log(w) + gum
        inc = banned_rows & (cons >= s0) & (cons < s0 + sc.shape[1])
        key[rows[inc], cons[inc] - s0] = -float("inf")
        allk = torch.cat([best_k, key], dim=1)
        top = torch.topk(allk, n_rec, dim=1)
        ids = torch.cat([best_i, torch.arange(s0, s0 + sc.shape[1], device=dev).expand(B, -1) + int(item_base)], dim=1)
        scs = torch.cat([best_s, sc], dim=1)
        best_k, best_i, best_s = top.values, torch.gather(ids, 1, top.indices), torch.gather(scs, 1, top.indices)
    return best_k, best_i, best_s


def recommend_from_embedding(model, user_ids, n_rec, user_embeddings, item_embeddings,
                             filter_consumed, random_rec, return_scores=False, user_vectors=None):
    """`user_embeddings[user_ids] @ item_embeddings[:n_items].T` + ranking, on device (parameter names of
    `recommend.py:57-65`).  `user_vectors` ([len(user_ids), D]) replaces the table lookup for dynamically computed
    user embeddings."""
    user_embeds, item_embeds = user_embeddings, item_embeddings
    n_items = model.n_items
    if n_rec > n_items:
        raise ValueError(f"`n_rec` {n_rec} exceeds num of items {n_items}")
    dev = item_embeds.device
    if not isinstance(item_embeds, torch.Tensor):
        # item embeddings sharded over the ranks (distributed.ShardedItemEmbeds): local fused score + top-k on every
        # rank's block, all-gather + merge of the [B, k] candidates; every rank returns the same lists
        if user_vectors is not None:
            U = user_vectors.to(dev).contiguous()
        else:
            U = user_embeds.index_select(0, torch.as_tensor(np.asarray(user_ids, dtype=np.int64), device=dev)).contiguous()
        ptr, cidx, flag = model.consumed_index.batch_csr(user_ids, n_rec, n_items, filter_consumed, dev)
        if random_rec:      # streaming Gumbel top-k per shard + merge of the [B, n_rec] candidates (ranking.py:65-73)
            return item_embeds.random_topk(U, n_rec, ptr, cidx, flag).cpu().numpy()
        s, ids = item_embeds.topk(U, n_rec, ptr, cidx, flag)
        if return_scores:
            sc = torch.sigmoid(s) if model.task == "ranking" else s
            return ids.cpu().numpy(), sc.cpu().numpy()
        return ids.cpu().numpy()
    if user_vectors is not None:
        U = user_vectors.to(dev).contiguous()
    else:
        uid = torch.as_tensor(np.asarray(user_ids, dtype=np.int64), device=dev)
        U = user_embeds.index_select(0, uid).contiguous()
    I = item_embeds[:n_items]
    ptr, cidx, flag = model.consumed_index.batch_csr(user_ids, n_rec, n_items, filter_consumed, dev)
    if random_rec:
        if U.shape[0] * n_items > (1 << 27):          # large catalogues: never materialise [B, N]
            return random_select_streaming(U, I, ptr, cidx, flag, n_rec).cpu().numpy()
        scores = U @ I.T
        banned = None
        if int(flag.sum()) > 0:
            banned = torch.zeros_like(scores, dtype=torch.bool)
            rows = torch.repeat_interleave(torch.arange(len(user_ids), device=dev), ptr[1:] - ptr[:-1])
            keep = flag.bool()[rows]
            banned[rows[keep], cidx[: rows.numel()].long()[keep]] = True
        ids = random_select_device(scores, banned, n_rec)
        return ids.cpu().numpy()
    s, ids = ops.score_topk(U, I.contiguous() if not I.is_contiguous() else I, n_rec, ptr, cidx, flag)
    if return_scores:
        sc = torch.sigmoid(s) if model.task == "ranking" else s
        return ids.cpu().numpy(), sc.cpu().numpy()
    return ids.cpu().numpy()
