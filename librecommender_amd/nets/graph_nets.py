"""LightGCN (`libreco/algorithms/torch_modules/lightgcn_module.py:7-96`) on the HIP path.

* The normalised bipartite Laplacian  A^ = D^-1/2 A D^-1/2  is built once, vectorised, straight
  into CSR (the reference fills a scipy dok matrix user by user, lightgcn_module.py:36-48) and
  stays on the device; its sparsity pattern is symmetric, so the transposed operator needed in
  the backward is the same CSR with values permuted by a precomputed transpose map (only edge
  dropout makes A^ itself asymmetric).
* Propagation  E^{l+1} = A^ E^l  is `lr_spmm_csr_f32` with the layer sum fused (`acc += Y`).
* Backward of the layer mean is the same recursion on the gradient:  G_L = D, G_l = D + A^T G_{l+1},
  with D the (deterministically scattered) gradient of the batch rows.
* Optimiser: torch-style Adam over the whole table (`training/torch_trainer.py:63-69`; every row
  receives gradient through the graph), one `lr_adam_dense_f32` launch.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops


def build_laplacian_csr(n_users: int, n_items: int, user_consumed):
    """CSR (rowptr int64, col int32, val fp32) of D^-1/2 A D^-1/2 over users then items, plus the
    map `tperm` with  A^T.val = A.val[tperm]  (same pattern)."""
    us = np.concatenate([np.full(len(v), u, dtype=np.int64) for u, v in user_consumed.items() if 0 <= u < n_users] or [np.zeros(0, np.int64)])
    its = np.concatenate([np.asarray(v, dtype=np.int64) for u, v in user_consumed.items() if 0 <= u < n_users] or [np.zeros(0, np.int64)])
    pairs = np.unique(us * n_items + its)                     # binary adjacency: repeats collapse
    u, i = pairs // n_items, pairs % n_items
    n = n_users + n_items
    rows = np.concatenate([u, n_users + i])
    cols = np.concatenate([n_users + i, u])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    val = (dinv[rows] * dinv[cols]).astype(np.float32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
    # transpose map: entry (r,c) at position p  <->  entry (c,r) at position tperm[p]
    key = rows * n + cols
    tkey = cols * n + rows
    tperm = np.searchsorted(key, tkey).astype(np.int64)
    return rowptr, cols.astype(np.int32), val, tperm


class LightGCNNet:
    def __init__(self, n_users, n_items, embed_size, n_layers, dropout_rate, user_consumed, device,
                 seed=42, lr=1e-3, epsilon=1e-8, reg=None, margin=1.0):
        self.n_users, self.n_items, self.K, self.L = n_users, n_items, embed_size, n_layers
        self.device, self.dropout = device, float(dropout_rate or 0.0)
        self.lr, self.epsilon, self.reg, self.margin = lr, epsilon, float(reg or 0.0), margin
        # same RNG protocol as the reference module: nn.Embedding construction, then normal_(0, 0.1)
        torch.manual_seed(seed)
        ue = torch.nn.Embedding(n_users, embed_size)
        ie = torch.nn.Embedding(n_items, embed_size)
        torch.nn.init.normal_(ue.weight, 0.0, 0.1)
        torch.nn.init.normal_(ie.weight, 0.0, 0.1)
        self.E = torch.cat([ue.weight.detach(), ie.weight.detach()]).to(device).contiguous()
        self.m = torch.zeros_like(self.E)
        self.v = torch.zeros_like(self.E)
        rp, ci, va, tp = build_laplacian_csr(n_users, n_items, user_consumed)
        self.rowptr = torch.from_numpy(rp).to(device)
        self.col = torch.from_numpy(ci).to(device)
        self.val = torch.from_numpy(va).to(device)
        self.tperm = torch.from_numpy(tp).to(device)
        self.step = 0
        n = n_users + n_items
        self._bufs = [torch.empty((n, embed_size), dtype=torch.float32, device=device) for _ in range(3)]

    # ---- propagation ------------------------------------------------------------------------
    def _edge_values(self, use_dropout: bool):
        if use_dropout and self.dropout > 0:          # lightgcn_module.py:90-96
            keep = 1.0 - self.dropout
            mask = torch.floor(torch.rand(self.val.numel(), device=self.device) + keep)
            return self.val * mask / keep
        return self.val

    def propagate(self, val: torch.Tensor) -> torch.Tensor:
        """mean(E^0 .. E^L), E^{l+1} = A^ E^l  (lightgcn_module.py:66-88)."""
        acc = self.E.clone()
        cur, nxt = self.E, self._bufs[0]
        for _ in range(self.L):
            ops.spmm_csr(self.rowptr, self.col, val, cur, out=nxt, acc=acc)
            cur, nxt = nxt, (self._bufs[1] if nxt is self._bufs[0] else self._bufs[0])
        return acc.div_(self.L + 1)

    def _backprop(self, D: torch.Tensor, val_t: torch.Tensor) -> torch.Tensor:
        G = D
        for _ in range(self.L):
            A = D.clone()
            ops.spmm_csr(self.rowptr, self.col, val_t, G, out=self._bufs[2], acc=A)
            G = A
        return G

    # ---- losses (`torchops/loss.py:5-90`) -----------------------------------------------------
    def _loss(self, loss_type, u, p, ng, labels):
        if loss_type in ("cross_entropy", "focal"):
            logits = (u * p).sum(1)
            lab = torch.as_tensor(labels, device=self.device, dtype=torch.float32)
            if loss_type == "cross_entropy":
                return F.binary_cross_entropy_with_logits(logits, lab)
            w = lab * 0.25 + (1 - lab) * 0.75
            prob = torch.sigmoid(logits)
            p_t = lab * prob + (1 - lab) * (1 - prob)
            bce = F.binary_cross_entropy_with_logits(logits, lab, reduction="none")
            return (w * (1 - p_t) ** 2.0 * bce).mean()
        if len(u) == len(p) == len(ng):
            pos, neg = (u * p).sum(1), (u * ng).sum(1)
        else:                                              # several negatives per positive
            f = len(ng) // len(p)
            pos = (u * p).sum(1).repeat_interleave(f)
            neg = torch.einsum("ik,ijk->ij", u, ng.view(len(p), f, -1)).reshape(-1)
        if loss_type == "bpr":
            return -F.logsigmoid(pos - neg).mean()
        return F.margin_ranking_loss(pos, neg, torch.ones_like(pos), margin=self.margin)

    def train_step(self, loss_type, users, items, items_neg=None, labels=None, lr=None):
        self.step += 1
        dev = self.device
        val = self._edge_values(use_dropout=True)
        out = self.propagate(val)
        ti = lambda x, off=0: torch.as_tensor(np.ascontiguousarray(x), device=dev).to(torch.int32) + off  # noqa: E731
        parts = [ti(users), ti(items, self.n_users)]
        if items_neg is not None:
            parts.append(ti(items_neg, self.n_users))
        idx = torch.cat(parts).contiguous()
        rows = ops.embed_gather(out, idx)
        rows.requires_grad_(True)
        nu, ni = len(parts[0]), len(parts[1])
        loss = self._loss(loss_type, rows[:nu], rows[nu:nu + ni], rows[nu + ni:] if items_neg is not None else None, labels)
        loss.backward()
        with torch.no_grad():
            D = torch.zeros_like(self.E)
            ops.embed_scatter_add(D, rows.grad, ops.build_segments(idx, D.shape[0]), alpha=1.0 / (self.L + 1))
            val_t = val[self.tperm] if (self.dropout > 0) else val      # A^ symmetric without dropout
            G = self._backprop(D, val_t)
            hp = ops.adam_hp(self.lr if lr is None else lr, self.step, eps=self.epsilon,
                             weight_decay=self.reg, tf_style=False)
            ops.adam_dense(self.E, self.m, self.v, hp, grows=G)
        return loss.detach(), G

    @torch.no_grad()
    def embeddings(self):
        out = self.propagate(self.val)
        return out[: self.n_users].contiguous(), out[self.n_users:].contiguous()


def cosine_warm_restart_lr(base_lr: float, epoch_float: float, T_0: int = 1, T_mult: int = 2) -> float:
    """torch CosineAnnealingWarmRestarts(T_0=1, T_mult=2) evaluated at `epoch + i/n_batches`
    (`training/torch_trainer.py:71-75,119-121`)."""
    n = int(math.log(epoch_float / T_0 * (T_mult - 1) + 1, T_mult)) if epoch_float >= T_0 else 0
    t_i = T_0 * T_mult ** n
    t_cur = epoch_float - T_0 * (T_mult ** n - 1) / (T_mult - 1)
    return base_lr * (1 + math.cos(math.pi * t_cur / t_i)) / 2
