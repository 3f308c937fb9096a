"""Kernel provider backed by the CPU oracle — lets the multi-process exchange logic of
librecommender_amd.parallel run under gloo on CPU.  TEST CODE ONLY."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import ops_np


class OracleKernels:
    def segments(self, idx, V, want_slots=False, tag=""):
        flat = idx.reshape(-1).numpy()
        pos, rows, start = ops_np.segments(flat, V)
        n = idx.numel()
        P = torch.zeros(max(n, 1), dtype=torch.int32); P[: len(pos)] = torch.from_numpy(pos)
        R = torch.zeros(max(n, 1), dtype=torch.int32); R[: len(rows)] = torch.from_numpy(rows)
        S = torch.zeros(max(n, 1) + 1, dtype=torch.int32); S[: len(start)] = torch.from_numpy(start)
        seg = SimpleNamespace(pos=P, rows=R, start=S, n_seg=torch.tensor([len(rows)], dtype=torch.int32), n=n, V=V,
                              slots=None)
        if want_slots:
            seg.slots = self._slots(seg, n).to(torch.int32)
        return seg

    def gather(self, table, ids):
        return torch.from_numpy(ops_np.embedding_lookup(table.numpy(), ids.numpy()))

    @staticmethod
    def _slots(seg, n_pos):
        n = int(seg.n_seg)
        start = seg.start[: n + 1].long()
        run = torch.repeat_interleave(torch.arange(n), start[1:] - start[:-1])
        slots = torch.full((n_pos,), -1, dtype=torch.long)
        slots[seg.pos[: run.numel()].long()] = run
        return slots

    def bag_pool(self, table, idx, combiner, oov):
        return torch.from_numpy(ops_np.bag_pool(table.detach().numpy(), idx.numpy(), combiner, oov))

    def bag_pool_bwd(self, gout, idx, V, combiner, oov):
        return torch.from_numpy(ops_np.bag_pool_bwd(gout.detach().numpy(), idx.numpy(), V, combiner, oov))

    def fm_pairwise(self, e):
        pair, fsum = ops_np.fm_pairwise(e.detach().numpy())
        return torch.from_numpy(pair), torch.from_numpy(fsum)

    def fm_pairwise_bwd(self, e, fsum, gpair):
        return torch.from_numpy(ops_np.fm_pairwise_bwd(e.detach().numpy(), gpair.detach().numpy()))

    def fm_fwd(self, cache, lin_cache, slots, want_e=True):
        e = cache[slots.long()]
        pair, fsum = ops_np.fm_pairwise(e.numpy())
        lin = lin_cache[slots.long()].squeeze(-1) if lin_cache is not None else None
        return e.clone(), torch.from_numpy(pair), torch.from_numpy(fsum), lin.clone()

    def fm_bwd_rows(self, cache, gdeep, gpair, fsum, B, F, seg, glin, bn_a, bn_c):
        K = cache.shape[1]
        slots = self._slots(seg, B * F)
        e = cache[slots].view(B, F, K)
        ge = gpair[:, None, :] * (fsum[:, None, :] - e)
        if gdeep is not None:
            ge = ge + gdeep.view(B, F, K)
        if bn_a is not None:
            ge = ge - bn_a.view(1, F, K) - bn_c.view(1, F, K) * e
        grows = torch.zeros_like(cache).index_add_(0, slots, ge.reshape(-1, K))
        glin_rows = torch.zeros(cache.shape[0]).index_add_(0, slots, glin.reshape(-1)) if glin is not None else None
        return grows, glin_rows

    def fm_bwd_adam(self, t, gdeep, gpair, fsum, B, F, seg, glin, bn_a, bn_c, hp):
        """Fused backward + row-wise Adam on full tables: per-run gradients from the compact row view."""
        rows = seg.rows[: int(seg.n_seg)].long()
        grows, glin_rows = self.fm_bwd_rows(t.embed[rows], gdeep, gpair, fsum, B, F, seg, glin, bn_a, bn_c)
        for tab, m, v, g in ((t.embed, t.m, t.v, grows), (t.lin, t.lin_m, t.lin_v, glin_rows.view(-1, 1))):
            w2, m2, v2 = ops_np.adam_step(tab[rows].numpy(), m[rows].numpy(), v[rows].numpy(), g.numpy(),
                                          hp["lr"], hp["step"], eps=hp["eps"])
            tab[rows], m[rows], v[rows] = torch.from_numpy(w2), torch.from_numpy(m2), torch.from_numpy(v2)

    def scatter_adam(self, table, m, v, grads, seg, hp):
        n = int(seg.n_seg)
        rows = seg.rows[:n].long()
        run = self._slots(seg, seg.n)
        g = torch.zeros((n, table.shape[1])).index_add_(0, run, grads.reshape(seg.n, -1))
        w2, m2, v2 = ops_np.adam_step(table[rows].numpy(), m[rows].numpy(), v[rows].numpy(), g.numpy(),
                                      hp["lr"], hp["step"], eps=hp["eps"])
        table[rows], m[rows], v[rows] = torch.from_numpy(w2), torch.from_numpy(m2), torch.from_numpy(v2)

    def adam_dense_rows(self, table, m, v, grads, seg, hp, l2=0.0):
        n = int(seg.n_seg)
        g = torch.zeros_like(table)
        if n:
            run = self._slots(seg, seg.n)
            keep = run >= 0
            g.index_add_(0, seg.rows[:n].long()[run[keep]], grads.reshape(seg.n, -1)[keep])
        g = g + 2.0 * l2 * table
        w2, m2, v2 = ops_np.adam_step(table.numpy(), m.numpy(), v.numpy(), g.numpy(), hp["lr"], hp["step"], eps=hp["eps"])
        table.copy_(torch.from_numpy(w2)); m.copy_(torch.from_numpy(m2)); v.copy_(torch.from_numpy(v2))

    def segment_sum(self, grads, seg):
        n = int(seg.n_seg)
        run = self._slots(seg, seg.n)
        keep = run >= 0
        return torch.zeros((max(seg.n, 1), grads.shape[1])).index_add_(0, run[keep], grads.reshape(seg.n, -1)[keep])

    def dense_adam(self, flat, m, v, grad, hp):
        w2, m2, v2 = ops_np.adam_step(flat.detach().numpy(), m.numpy(), v.numpy(), grad.numpy(),
                                      hp["lr"], hp["step"], eps=hp["eps"],
                                      weight_decay=hp.get("weight_decay", 0.0), tf_style=hp.get("tf_style", True))
        with torch.no_grad():
            flat.copy_(torch.from_numpy(w2)); m.copy_(torch.from_numpy(m2)); v.copy_(torch.from_numpy(v2))

    def adam_table(self, table, m, v, grad, hp, vmax=None):
        """torch.optim.Adam's own single-tensor update (the reference's optimiser, torch_trainer.py:63-69)."""
        from torch.optim.adam import adam
        adam([table], [grad], [m], [v], [vmax] if vmax is not None else [], [torch.tensor(float(hp["step"] - 1))],
             amsgrad=vmax is not None, beta1=0.9, beta2=0.999, lr=hp["lr"], weight_decay=hp.get("weight_decay", 0.0),
             eps=hp["eps"], maximize=False, foreach=False, capturable=False, differentiable=False, fused=False,
             has_complex=False)

    def adam_hp(self, lr, step, eps):
        return {"lr": lr, "step": step, "eps": eps}

    def adam_hp_torch(self, lr, step, eps, weight_decay=0.0):
        return {"lr": lr, "step": step, "eps": eps, "weight_decay": weight_decay, "tf_style": False}

    def spmm(self, rowptr, col, val, X, out, acc):
        y = torch.from_numpy(ops_np.spmm_csr(rowptr.numpy(), col.numpy(), val.numpy(), X.numpy()))
        if out is not None:         # (out None with acc: accumulate only)
            out.copy_(y)
        if acc is not None:
            acc.add_(y)
        return out

    def scatter_add(self, table, grads, seg, alpha=1.0):
        n = int(seg.n_seg)
        run = self._slots(seg, seg.n)
        g = torch.zeros((n, table.shape[1])).index_add_(0, run, grads.reshape(seg.n, -1))
        table[seg.rows[:n].long()] += alpha * g

    def score_topk(self, users, items, k, ptr, cidx, flag, item_base):
        P = users.numpy().astype(np.float64) @ items.numpy().astype(np.float64).T
        B, N = P.shape
        ids = np.full((B, k), -1, np.int64); sc = np.full((B, k), -np.inf, np.float32)
        for u in range(B):
            p = P[u].copy()
            if ptr is not None and (flag is None or flag[u]):
                c = cidx[int(ptr[u]): int(ptr[u + 1])].numpy() - item_base
                c = c[(c >= 0) & (c < N)]
                p[c] = -np.inf
            order = np.lexsort((np.arange(N), -p))[:k]
            order = order[np.isfinite(p[order])]
            ids[u, : len(order)] = order + item_base
            sc[u, : len(order)] = P[u][order]
        return torch.from_numpy(sc), torch.from_numpy(ids)

    def softmax_ce(self, X, Y, col_bias, row_ids, col_ids, pos0):
        # tfops/loss.py:71-75 over adjust_logits (two_tower.py:458-479), materialised
        import torch.nn.functional as Fn

        B = X.shape[0]
        logits = X @ Y.T
        if col_bias is not None:
            logits = logits + col_bias.view(1, -1)
        target = torch.arange(B, device=X.device) + pos0
        if row_ids is not None:
            same = row_ids.view(-1, 1) == col_ids.view(1, -1)
            same[torch.arange(B, device=X.device), target] = False
            logits = torch.where(same, torch.full_like(logits, torch.finfo(torch.float32).min), logits)
        return Fn.cross_entropy(logits, target, reduction="none")

    def din_attention(self, q, keys, lens, W1, b1, W2, b2):
        # layers/attention.py:28-64 in torch ops (differentiable)
        B, L, Kp = keys.shape
        qt = q[:, None, :].expand(-1, L, -1)
        h = torch.sigmoid(torch.cat([qt, keys, qt - keys, qt * keys], dim=2) @ W1 + b1)
        s = ((h @ W2.view(-1, 1)).squeeze(-1) + b2) * (Kp ** -0.5)
        mask = torch.arange(L, device=keys.device)[None, :] < lens[:, None]
        s = torch.where(mask, s, torch.full_like(s, -(2.0 ** 32) + 1))
        return (torch.softmax(s, dim=1)[:, None, :] @ keys).squeeze(1)

    def topk_merge(self, scores, ids):
        S, B, k = scores.shape
        s = scores.permute(1, 0, 2).reshape(B, S * k).numpy()
        i = ids.permute(1, 0, 2).reshape(B, S * k).numpy()
        out_s = np.full((B, k), -np.inf, np.float32); out_i = np.full((B, k), -1, np.int64)
        for u in range(B):
            ok = i[u] >= 0
            order = np.lexsort((i[u][ok], -s[u][ok].astype(np.float64)))[:k]
            out_s[u, : len(order)] = s[u][ok][order]; out_i[u, : len(order)] = i[u][ok][order]
        return torch.from_numpy(out_s), torch.from_numpy(out_i)
