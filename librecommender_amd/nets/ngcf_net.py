"""NGCF (`libreco/algorithms/torch_modules/ngcf_module.py:8-146`) over the same HIP kernels as LightGCN
(an adjacent model that reuses the hot path).

* Laplacian  L = D^-1 (A + I)  (row-normalised, self loops; `ngcf_module.py:62-88`) built vectorised
  straight into CSR.  L is NOT symmetric, but its pattern is, so  L^T  is the same CSR with the
  values permuted by a precomputed transpose map.
* Per layer:  S = L X  (`lr_spmm_csr_f32`),  X' = normalize(leaky_relu(S W_self + b_self +
  (S * X) W_pair + b_pair)).  The two [n, in] x [in, out] products are library GEMMs; the sparse
  product and its transpose in the backward, the batch-row gather, its scatter-add backward and
  Adam are the hand-written kernels.
* Output = concat of all layers ([n, K + sum(hidden)]); torch-style Adam (weight_decay = reg,
  optional AMSGrad; `training/torch_trainer.py:63-69`) over the node table and the layer weights.

Kernels are reached through a provider (`parallel.HipKernels`; CPU tests inject the oracle)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from ..utils.device import to_device
from .graph_nets import LightGCNNet


def build_ngcf_laplacian_csr(n_users: int, n_items: int, user_consumed):
    """CSR (rowptr int64, col int32, val fp32) of D^-1 (A + I) over users then items, and `tperm`
    with  L^T.val = L.val[tperm]."""
    known = [(u, v) for u, v in user_consumed.items() if 0 <= u < n_users and len(v)]
    us = np.concatenate([np.full(len(v), u, dtype=np.int64) for u, v in known] or [np.zeros(0, np.int64)])
    its = np.concatenate([np.asarray(v, dtype=np.int64) for _, v in known] or [np.zeros(0, np.int64)])
    pairs = np.unique(us * n_items + its)                     # binary adjacency
    u, i = pairs // n_items, pairs % n_items
    n = n_users + n_items
    diag = np.arange(n, dtype=np.int64)
    rows = np.concatenate([u, n_users + i, diag])
    cols = np.concatenate([n_users + i, u, diag])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    counts = np.bincount(rows, minlength=n)
    # `+ ssp.eye(...)` promotes the matrix to float64 (:75): the reciprocal row sums are formed in
    # double and rounded once when the COO values become a float32 tensor (:84-87)
    inv = np.power(counts.astype(np.float64), -1.0).astype(np.float32)
    val = inv[rows]
    rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    tperm = np.searchsorted(rows * n + cols, cols * n + rows).astype(np.int64)
    return rowptr, cols.astype(np.int32), val, tperm


class _SparseProduct(torch.autograd.Function):
    """Y = L X ;  dX = L^T dY  (values of the transpose supplied by the caller)."""

    @staticmethod
    def forward(ctx, X, net, val, val_t):
        ctx.net, ctx.val_t = net, val_t
        out = torch.empty_like(X)
        net.kern.spmm(net.rowptr, net.col, val, X.contiguous(), out, None)
        return out

    @staticmethod
    def backward(ctx, gY):
        net = ctx.net
        gX = torch.empty_like(gY)
        net.kern.spmm(net.rowptr, net.col, ctx.val_t, gY.contiguous(), gX, None)
        return gX, None, None, None


class _BatchRows(torch.autograd.Function):
    """rows = table[idx] ;  dtable = deterministic scatter-add of drows."""

    @staticmethod
    def forward(ctx, table, idx, net):
        ctx.net, ctx.idx, ctx.shape = net, idx, table.shape
        return net.kern.gather(table.contiguous(), idx)

    @staticmethod
    def backward(ctx, grows):
        net = ctx.net
        D = torch.zeros(ctx.shape, dtype=torch.float32, device=grows.device)
        net.kern.scatter_add(D, grows.contiguous(), net.kern.segments(ctx.idx, ctx.shape[0], tag="ngcf"), 1.0)
        return D, None, None


class NGCFNet:
    def __init__(self, n_users, n_items, embed_size, hidden_units, node_dropout, message_dropout,
                 user_consumed, device, seed=42, lr=1e-3, epsilon=1e-8, reg=None, margin=1.0,
                 amsgrad=False, kern=None):
        if kern is None:
            from ..parallel import HipKernels
            kern = HipKernels()
        self.kern = kern
        self.n_users, self.n_items, self.K = n_users, n_items, embed_size
        self.layers = list(hidden_units)
        self.device = device
        self.node_dropout, self.message_dropout = float(node_dropout or 0.0), float(message_dropout or 0.0)
        self.lr, self.epsilon, self.reg, self.margin = lr, epsilon, float(reg or 0.0), margin
        # RNG protocol of the reference module (`init_weights`, :33-60): Xavier-uniform user table,
        # item table, then per layer W_self, W_pair (biases are zeros and draw nothing)
        torch.manual_seed(seed)
        xav = lambda *shape: torch.nn.init.xavier_uniform_(torch.empty(*shape))  # noqa: E731
        ue, ie = xav(n_users, embed_size), xav(n_items, embed_size)
        self.params = {"embed": torch.cat([ue, ie])}
        dims = [embed_size, *self.layers]
        for k in range(len(self.layers)):
            self.params[f"W_self_{k}"] = xav(dims[k], dims[k + 1])
            self.params[f"b_self_{k}"] = torch.zeros(1, dims[k + 1])
            self.params[f"W_pair_{k}"] = xav(dims[k], dims[k + 1])
            self.params[f"b_pair_{k}"] = torch.zeros(1, dims[k + 1])
        self.params = {k: p.to(device).contiguous() for k, p in self.params.items()}
        self.m = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.vmax = {k: torch.zeros_like(p) for k, p in self.params.items()} if amsgrad else None
        rp, ci, va, tp = build_ngcf_laplacian_csr(n_users, n_items, user_consumed)
        self.rowptr = torch.from_numpy(rp).to(device)
        self.col = torch.from_numpy(ci).to(device)
        self.val = torch.from_numpy(va).to(device)
        self.tperm = torch.from_numpy(tp).to(device)
        self.val_t = self.val[self.tperm].contiguous()
        self.step = 0

    @property
    def E(self):
        return self.params["embed"]

    @property
    def out_dim(self):
        return self.K + sum(self.layers)

    # ---- propagation (`embedding_propagation`, :93-134) ---------------------------------------
    def _edge_values(self, use_dropout):
        if use_dropout and self.node_dropout > 0:          # `sparse_dropout` :136-146
            keep = 1.0 - self.node_dropout
            mask = torch.floor(torch.rand(self.val.numel(), device=self.device) + keep)
            val = self.val * mask / keep
            return val, val[self.tperm].contiguous()
        return self.val, self.val_t

    def propagate(self, P, use_dropout):
        val, val_t = self._edge_values(use_dropout)
        X = P["embed"]
        outs = [X]
        for k in range(len(self.layers)):
            S = _SparseProduct.apply(X, self, val, val_t)
            msg = S @ P[f"W_self_{k}"] + P[f"b_self_{k}"] + (S * X) @ P[f"W_pair_{k}"] + P[f"b_pair_{k}"]
            msg = F.leaky_relu(msg, negative_slope=0.2)
            if use_dropout and self.message_dropout > 0:
                msg = F.dropout(msg, p=self.message_dropout)
            X = F.normalize(msg, p=2, dim=1)
            outs.append(X)
        return torch.cat(outs, dim=1)

    def train_step(self, loss_type, users, items, items_neg=None, labels=None, lr=None):
        self.step += 1
        dev = self.device
        P = {k: p.detach().requires_grad_(True) for k, p in self.params.items()}
        out = self.propagate(P, use_dropout=True)
        ti = lambda x, off=0: to_device(x, dev).to(torch.int32) + off  # noqa: E731
        parts = [ti(users), ti(items, self.n_users)]
        if items_neg is not None:
            parts.append(ti(items_neg, self.n_users))
        idx = torch.cat(parts).contiguous()
        rows = _BatchRows.apply(out, idx, self)
        nu, ni = len(parts[0]), len(parts[1])
        loss = LightGCNNet._loss(self, loss_type, rows[:nu], rows[nu:nu + ni],
                                 rows[nu + ni:] if items_neg is not None else None, labels)
        grads = torch.autograd.grad(loss, list(P.values()))
        with torch.no_grad():
            hp = self.kern.adam_hp_torch(self.lr if lr is None else lr, self.step, self.epsilon, self.reg)
            for (k, p), g in zip(self.params.items(), grads):
                self.kern.adam_table(p, self.m[k], self.v[k], g.contiguous(), hp,
                                     vmax=None if self.vmax is None else self.vmax[k])
        return loss.detach(), dict(zip(self.params, grads))

    @torch.no_grad()
    def embeddings(self):
        out = self.propagate(self.params, use_dropout=False)
        return out[: self.n_users].contiguous(), out[self.n_users:].contiguous()
