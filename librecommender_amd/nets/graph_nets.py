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

from ..utils.device import to_device
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


def interactions_from_consumed(n_users: int, user_consumed, device):
    """(users, items) int32 device arrays of the interaction list behind `user_consumed` (one entry per list element;
    repeats collapse in the Laplacian build, like the reference's dok assignment, lightgcn_module.py:36-40)."""
    lens = np.fromiter((len(user_consumed.get(u, ())) for u in range(n_users)), dtype=np.int64, count=n_users)
    us = np.repeat(np.arange(n_users, dtype=np.int32), lens)
    its = np.fromiter((i for u in range(n_users) for i in user_consumed.get(u, ())), dtype=np.int32, count=int(lens.sum()))
    return torch.from_numpy(us).to(device), torch.from_numpy(its).to(device)


class LightGCNNet:
    def __init__(self, n_users, n_items, embed_size, n_layers, dropout_rate, user_consumed, device,
                 seed=42, lr=1e-3, epsilon=1e-8, reg=None, margin=1.0, amsgrad=False, interactions=None,
                 want_tperm=None, torch_init=True):
        """`interactions` = (users, items) int32 DEVICE arrays: the Laplacian is built from them on the device
        (`lr_csr_laplacian_build`) instead of from `user_consumed`; on a HIP device the dict is flattened to that list
        too (the host numpy build `build_laplacian_csr` is kept as the definition the device build is tested against).
        `want_tperm`: keep the transpose map (needed for edge dropout; default: only when `dropout_rate` > 0).
        `torch_init=False`: initial rows drawn on the device (same N(0, 0.1) law, not the reference's CPU RNG stream —
        for tables too large to initialise through `torch.nn.Embedding` on the host)."""
        self.n_users, self.n_items, self.K, self.L = n_users, n_items, embed_size, n_layers
        self.device, self.dropout = device, float(dropout_rate or 0.0)
        self.lr, self.epsilon, self.reg, self.margin = lr, epsilon, float(reg or 0.0), margin
        if torch_init:
            # same RNG protocol as the reference module: nn.Embedding construction, then normal_(0, 0.1)
            torch.manual_seed(seed)
            ue = torch.nn.Embedding(n_users, embed_size)
            ie = torch.nn.Embedding(n_items, embed_size)
            torch.nn.init.normal_(ue.weight, 0.0, 0.1)
            torch.nn.init.normal_(ie.weight, 0.0, 0.1)
            self.E = torch.cat([ue.weight.detach(), ie.weight.detach()]).to(device).contiguous()
        else:
            g = torch.Generator(device=device).manual_seed(seed)
            self.E = torch.empty((n_users + n_items, embed_size), dtype=torch.float32, device=device).normal_(0.0, 0.1, generator=g)
        self.m = torch.zeros_like(self.E)
        self.v = torch.zeros_like(self.E)
        self.vmax = torch.zeros_like(self.E) if amsgrad else None      # torch_trainer.py:63-69
        if want_tperm is None:
            want_tperm = self.dropout > 0
        if interactions is None and torch.device(device).type == "cuda":
            interactions = interactions_from_consumed(n_users, user_consumed, device)
        if interactions is not None:
            eu, ei = interactions
            self.rowptr, self.col, self.val, tp = ops.csr_laplacian(eu.to(torch.int32).contiguous(),
                                                                    ei.to(torch.int32).contiguous(), n_users, n_items,
                                                                    want_tperm=want_tperm)
            self.tperm = tp.long() if tp is not None else None
        else:
            rp, ci, va, tp = build_laplacian_csr(n_users, n_items, user_consumed)
            self.rowptr = torch.from_numpy(rp).to(device)
            self.col = torch.from_numpy(ci).to(device)
            self.val = torch.from_numpy(va).to(device)
            self.tperm = torch.from_numpy(tp).to(device)
        self.step = 0
        n = n_users + n_items
        self._bufs = [torch.empty((n, embed_size), dtype=torch.float32, device=device) for _ in range(3)]
        self._D = None

    # ---- propagation ------------------------------------------------------------------------
    def _plan(self):
        """Chunk lists of the long rows: built by the first product, reused by every later one (the graph is static)."""
        if getattr(self, "_spmm_plan", None) is None or not self._spmm_plan.matches(self.rowptr, self.col.numel(), self.E.shape[1]):
            self._spmm_plan = ops.SpmmPlan(self.rowptr, self.col.numel(), self.E.shape[1])
        return self._spmm_plan

    def _edge_values(self, use_dropout: bool):
        if use_dropout and self.dropout > 0:          # lightgcn_module.py:90-96
            keep = 1.0 - self.dropout
            mask = torch.floor(torch.rand(self.val.numel(), device=self.device) + keep)
            return self.val * mask / keep
        return self.val

    def propagate(self, val: torch.Tensor, mean: bool = True) -> torch.Tensor:
        """mean(E^0 .. E^L), E^{l+1} = A^ E^l  (lightgcn_module.py:66-88).  `mean=False` returns the SUM of the layers
        (the training step scales the few rows it gathers instead of the whole node table)."""
        acc = self.E.clone()
        cur, nxt = self.E, self._bufs[0]
        for _ in range(self.L):
            ops.spmm_csr(self.rowptr, self.col, val, cur, out=nxt, acc=acc, plan=self._plan())
            cur, nxt = nxt, (self._bufs[1] if nxt is self._bufs[0] else self._bufs[0])
        return acc.div_(self.L + 1) if mean else acc

    def _backprop(self, D: torch.Tensor, val_t: torch.Tensor, grad_rows: torch.Tensor, seg, alpha: float,
                  batch_rows=None, last: bool = True) -> torch.Tensor:
        """G_L of the recursion G_0 = D, G_{l+1} = D + A^T G_l (the gradient of mean(E^0..E^L) w.r.t. E^0).  D is nonzero
        on the batch's rows only: instead of cloning it into the accumulator of every layer (a pass over the whole
        node table), each layer's product is written plainly and the batch rows' gradient is scattered onto it —
        the same sums in the same order (x + y == y + x), 10 GB less traffic per layer at cfg 5."""
        G = D
        for l in range(self.L if last else self.L - 1):          # (`last=False`: the caller fuses the last product with Adam)
            out = self._bufs[l % 2]              # (the forward's layer buffers are free by now)
            ops.spmm_csr(self.rowptr, self.col, val_t, G, out=out, plan=self._plan(),
                         x_rows=batch_rows if l == 0 else None)      # G_0 = D: nonzero on the batch's rows only
            ops.embed_scatter_add(out, grad_rows, seg, alpha=alpha)
            G = out
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
        """One optimiser step.  Returns (loss, G): G is d loss / d E^0 as a table — or None when the optimiser step ran as the
        epilogue of the last backward product (`spmm_adam`, the default at L >= 2 on the HIP kernels): the gradient table is
        then never written.  `fuse_adam = False` on the net keeps the table for diagnostics."""
        self.step += 1
        dev = self.device
        val = self._edge_values(use_dropout=True)
        ti = lambda x, off=0: to_device(x, dev).to(torch.int32) + off  # noqa: E731
        parts = [ti(users), ti(items, self.n_users)]
        if items_neg is not None:
            parts.append(ti(items_neg, self.n_users))
        idx = torch.cat(parts).contiguous()
        # mean(E^0 .. E^L) is only needed at the batch's rows: the layers are kept (three buffers) and summed on the
        # gathered rows in the order the accumulating form adds them, ((E^0 + E^1) + E^2) + ... — no clone of the node
        # table and no accumulator read-modify-write in the L products (40 GB per step at cfg 5)
        # Two of the 2 L products of a step touch the batch's rows only: the LAST forward layer is read at the batch's rows and
        # nowhere else (`y_rows`: the other rows are not computed), and the FIRST backward product multiplies by a gradient
        # that is zero outside them (`x_rows`: zero rows are not read) — same sums, bit for bit (csrc/spmm.hip)
        bm = None
        if self.E.shape[1] in (16, 32, 64, 128) and self.L >= 1:
            if getattr(self, "_batch_rows", None) is None:
                self._batch_rows = ops.RowBitmap(self.E.shape[0], dev)
            bm = self._batch_rows.set(idx)
        cur = self.E
        rows = ops.embed_gather(cur, idx)
        for l in range(self.L):
            nxt = self._bufs[l % 3]
            ops.spmm_csr(self.rowptr, self.col, val, cur, out=nxt, plan=self._plan(),
                         y_rows=bm if l == self.L - 1 else None)
            rows.add_(ops.embed_gather(nxt, idx))
            cur = nxt
        rows.div_(self.L + 1)
        rows.requires_grad_(True)
        nu, ni = len(parts[0]), len(parts[1])
        loss = self._loss(loss_type, rows[:nu], rows[nu:nu + ni], rows[nu + ni:] if items_neg is not None else None, labels)
        loss.backward()
        with torch.no_grad():
            if self._D is None:                   # persistent d loss / d (layer sum): zero outside the batch's rows
                self._D = torch.zeros_like(self.E)
            D, alpha = self._D, 1.0 / (self.L + 1)
            seg = ops.build_segments(idx, D.shape[0])
            ops.embed_scatter_add(D, rows.grad, seg, alpha=alpha)
            val_t = val[self.tperm] if (self.dropout > 0) else val      # A^ symmetric without dropout
            hp = ops.adam_hp(self.lr if lr is None else lr, self.step, eps=self.epsilon,
                             weight_decay=self.reg, tf_style=False)
            # the LAST backward product's rows are d loss / d E^0: the optimiser step is that product's epilogue (no gradient
            # table written and read in between); its own batch rows enter through a row -> segment map
            fuse = bm is not None and self.L >= 2 and getattr(self, "fuse_adam", True)
            G = self._backprop(D, val_t, rows.grad, seg, alpha, bm, last=not fuse)
            if self.L == 0:
                G = D.clone()
            D.index_fill_(0, idx.long(), 0.0)      # back to zeros (the batch's rows only)
            if bm is not None:
                bm.clear(idx)
            if fuse:
                if getattr(self, "_row_slot", None) is None:
                    self._row_slot = torch.full((self.E.shape[0],), -1, dtype=torch.int32, device=dev)
                gsum = ops.embed_segment_sum(rows.grad, seg)
                if getattr(self, "_slots_marked", False):       # a step that raised in between left marks behind
                    self._row_slot.fill_(-1)
                self._slots_marked = True
                ops.row_slots(seg, self._row_slot, True)
                ops.spmm_csr_adam(self.rowptr, self.col, val_t, G, self.E, self.m, self.v, hp, self._plan(), vmax=self.vmax,
                                  row_slot=self._row_slot, gsum=gsum, alpha=alpha)
                ops.row_slots(seg, self._row_slot, False)
                self._slots_marked = False
                G = None
            else:
                ops.adam_dense(self.E, self.m, self.v, hp, grows=G, vmax=self.vmax)
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


class ShardedLightGCNNet:
    """LightGCN over several GPUs: 1-D row partition of the node table.

    Rank r owns rows [lo_r, hi_r) of E = [users | items], the matching row slice of A^ (CSR with
    global column ids) and the Adam moments of its rows.  One layer = all-gather of the current
    layer's rows (n*K*4 bytes: on a random bipartite graph a row slice references nearly every
    column, so "boundary rows" == all rows) + a local `lr_spmm_csr_f32`.  The batch is
    data-parallel: each rank gathers its samples' rows from the all-gathered output, the row
    gradients travel to their owners by all-to-all and are summed there in a fixed order, the
    backward recursion G_l = D + A^T G_{l+1} reuses the same partition (A^ is symmetric), and
    every rank applies torch-style Adam (optionally AMSGrad) to its own rows.  Edge dropout (lightgcn_module.py:90-96: every
    stored entry of the Laplacian kept with probability 1 - p, the kept ones scaled by 1 / (1 - p), a fresh mask per step)
    needs the mask of entry (c, r) on the rank that owns row r for the backward product (A_drop^T), while the forward of
    the rank owning row c drew it: the mask is therefore a counter-based function of (step, row, column) — any rank computes
    the mask of any entry, the law is the reference's (independent Bernoulli per entry), the stream is not (unseeded there).

    Overlap (round 5; `chunks` > 1, the default under more than one rank): a product does not wait for the whole all-gather.
    The layer input travels as `chunks` all-gathers of [per / chunks] rows per rank — every one of them uses all xGMI links, a
    gather per SOURCE RANK would use one — issued together on the collective stream; the rank's slice of A^ is held
    column-blocked (block c = the columns whose row offset inside its owner falls into chunk c), and block c is multiplied as
    soon as chunk c has arrived, adding into the output (`acc += A_c X_c`, `lr_spmm_csr_*` with Y == NULL) while chunk c + 1 is
    on the wire.  The last backward product keeps the one-piece form (its epilogue is the optimiser step).

    Compute goes through a kernel provider (`parallel.HipKernels`; tests inject the oracle)."""

    def __init__(self, n_users, n_items, embed_size, n_layers, user_consumed, device, kern=None,
                 seed=42, lr=1e-3, epsilon=1e-8, reg=None, margin=1.0, group=None, interactions=None, torch_init=True,
                 dropout=0.0, amsgrad=False, chunks=None):
        """`interactions` = (users, items) device tensors of the interaction list: the graph without the host dict;
        `torch_init=False` draws this rank's slice of the N(0, 0.1) table on the device (a counter-free generator stream
        per rank: for tables too large to initialise through `torch.nn.Embedding` on the host)."""
        import torch.distributed as dist
        from ..parallel import HipKernels

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.kern = kern or HipKernels()
        self.n_users, self.n_items, self.K, self.L = n_users, n_items, embed_size, n_layers
        self.device, self.lr, self.epsilon, self.reg, self.margin = device, lr, epsilon, float(reg or 0.0), margin
        n = n_users + n_items
        self.n = n
        per = (n + self.world - 1) // self.world               # equal row blocks (all-gather friendly)
        self.per = per
        self.lo, self.hi = min(self.rank * per, n), min((self.rank + 1) * per, n)
        self.E = torch.zeros((per, embed_size), dtype=torch.float32, device=device)   # zero-padded tail
        if torch_init:
            # identical initial table on every rank (reference RNG protocol), then keep the own slice
            torch.manual_seed(seed)
            ue = torch.nn.Embedding(n_users, embed_size)
            ie = torch.nn.Embedding(n_items, embed_size)
            torch.nn.init.normal_(ue.weight, 0.0, 0.1)
            torch.nn.init.normal_(ie.weight, 0.0, 0.1)
            full = torch.cat([ue.weight.detach(), ie.weight.detach()])
            self.E[: self.hi - self.lo] = full[self.lo:self.hi].to(device)
            del full
        else:
            g = torch.Generator(device=device).manual_seed(seed + 7919 * self.rank)
            self.E[: self.hi - self.lo].normal_(0.0, 0.1, generator=g)
        self.m, self.v = torch.zeros_like(self.E), torch.zeros_like(self.E)
        self.vmax = torch.zeros_like(self.E) if amsgrad else None      # torch_trainer.py:63-69
        self.dropout, self._drop_seed, self._row_of_nnz = float(dropout or 0.0), int(seed) * 1_000_003 + 12345, None
        if torch.device(device).type == "cuda" and isinstance(self.kern, HipKernels):
            # Laplacian built on the device (lr_csr_laplacian_build), this rank's row slice cut out of it
            if interactions is not None:
                eu, ei = (x.to(torch.int32).contiguous() for x in interactions)
            else:
                eu, ei = interactions_from_consumed(n_users, user_consumed, device)
            rp_d, ci_d, va_d, _ = ops.csr_laplacian(eu, ei, n_users, n_items, want_tperm=False)
            a, b = int(rp_d[self.lo]), int(rp_d[self.hi])
            rp_loc = torch.full((per + 1,), b - a, dtype=torch.int64, device=device)
            rp_loc[: self.hi - self.lo + 1] = rp_d[self.lo:self.hi + 1] - a
            self.rowptr, self.col, self.val = rp_loc, ci_d[a:b].clone(), va_d[a:b].clone()
            del rp_d, ci_d, va_d
        else:
            rp, ci, va, _ = build_laplacian_csr(n_users, n_items, user_consumed)
            a, b = int(rp[self.lo]), int(rp[self.hi])
            rp_loc = np.full(per + 1, b - a, dtype=np.int64)                              # padded rows: empty
            rp_loc[: self.hi - self.lo + 1] = rp[self.lo:self.hi + 1] - a
            self.rowptr = torch.from_numpy(rp_loc).to(device)
            self.col = torch.from_numpy(self._pad_cols(ci[a:b])).to(device)
            self.val = torch.from_numpy(va[a:b]).to(device)
        self.step = 0
        import os

        if chunks is None:      # (the environment variable also forces the blocked form onto one rank: measurements)
            env = os.environ.get("LIBRECO_LGCN_CHUNKS")
            chunks = int(env) if env else (4 if self.world > 1 else 1)
        self.chunks = max(1, min(int(chunks), per))
        self._blocks = None

    # ---- column-blocked slice + chunked all-gather ---------------------------------------------
    def _col_blocks(self):
        """The slice of A^ once more, column-blocked for the chunked all-gather: block c holds the entries whose column is row
        `off` of its owner with off in [c * pc, (c + 1) * pc); its column ids are rows of the chunk buffer
        [chunk c | rank-major | pc rows] the chunk's all-gather fills.  -> list of (rowptr, col, index of the entry in the
        slice's own arrays); built once (the graph is static), row order and the order inside a row are the slice's."""
        if self._blocks is not None:
            return self._blocks
        per, W, dev = self.per, self.world, self.col.device
        pc = (per + self.chunks - 1) // self.chunks
        nC = (per + pc - 1) // pc
        col = self.col.long()
        owner = torch.div(col, per, rounding_mode="floor")
        off = col - owner * per
        c = torch.div(off, pc, rounding_mode="floor")
        pos = c * (W * pc) + owner * torch.clamp(per - c * pc, max=pc) + (off - c * pc)
        deg = self.rowptr[1:] - self.rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(per, device=dev, dtype=torch.int64), deg)
        perm = torch.argsort(c, stable=True)
        counts = torch.bincount(c * per + row, minlength=nC * per).view(nC, per)
        blocks, start = [], 0
        for k in range(nC):
            rp = torch.zeros(per + 1, dtype=torch.int64, device=dev)
            rp[1:] = torch.cumsum(counts[k], 0)
            n_k = int(rp[-1])
            pk = perm[start:start + n_k]
            blocks.append((rp, pos[pk].to(torch.int32).contiguous(), pk))
            start += n_k
        self._pc, self._nC = pc, nC
        self._chunk_buf = torch.zeros((nC * W * pc, self.K), dtype=torch.float32, device=self.E.device)
        self._block_val = [self.val[pk].contiguous() for _, _, pk in blocks]
        self._blocks = blocks
        return blocks

    def _block_vals(self, val):
        """Per-block values of a per-entry value array of the slice (the Laplacian's own, or this step's dropped ones)."""
        if val is self.val:
            return self._block_val
        return [val[pk].contiguous() for _, _, pk in self._blocks]

    def _chunk_pos(self, idx):
        """Rows of the chunk buffer that hold the global node ids `idx`."""
        per, pc, W = self.per, self._pc, self.world
        i = idx.long()
        owner = torch.div(i, per, rounding_mode="floor")
        off = i - owner * per
        c = torch.div(off, pc, rounding_mode="floor")
        return (c * (W * pc) + owner * torch.clamp(per - c * pc, max=pc) + (off - c * pc)).to(torch.int32)

    def _chunked_spmm(self, local, vals, out, y_rows=None):
        """out = A^[slice] X with X = the all-gather of `local` [per, K], the transfer in `chunks` pieces and block c of the
        slice multiplied as soon as piece c is there.  Leaves X in the chunk buffer (`_chunk_pos` addresses it)."""
        from ..parallel import _all_gather_into_async

        blocks = self._col_blocks()
        per, pc, W, buf = self.per, self._pc, self.world, self._chunk_buf
        works = []
        for k in range(self._nC):           # all pieces are enqueued at once: the collective stream runs them back to back
            n = min(pc, per - k * pc)
            works.append(_all_gather_into_async(buf[k * W * pc: k * W * pc + W * n], local[k * pc: k * pc + n], group=self.group))
        kw = {} if y_rows is None else {"y_rows": y_rows}
        for k, (rp, colk, _) in enumerate(blocks):
            if works[k] is not None:
                works[k].wait()
            if k == 0:
                self.kern.spmm(rp, colk, vals[0], buf, out, None, **kw)
            else:
                self.kern.spmm(rp, colk, vals[k], buf, None, out, **kw)         # out += A_k X_k, nothing else stored

    def _pad_cols(self, cols):
        """global node id -> row of the all-gathered [W*per, K] buffer (blocks are padded to `per`)."""
        return cols.astype(np.int32)     # equal blocks: position == global id (only the tail is padded)

    # ---- edge dropout -------------------------------------------------------------------------
    @staticmethod
    def _keep_mask(rows: torch.Tensor, cols: torch.Tensor, n: int, seed: int, keep: float) -> torch.Tensor:
        """Bernoulli(keep) per ORDERED pair (row, col) as a function of (seed, row, col): splitmix64's finaliser over the
        pair's index row * n + col xor the seed (int64 arithmetic wraps; logical shifts emulated by masking)."""
        def c64(x):
            return x - (1 << 64) if x >= (1 << 63) else x
        x = (rows.to(torch.int64) * int(n) + cols.to(torch.int64)) ^ c64((seed * 0x9E3779B97F4A7C15) & ((1 << 64) - 1))
        x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * c64(0xBF58476D1CE4E5B9)
        x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * c64(0x94D049BB133111EB)
        x = x ^ ((x >> 31) & ((1 << 33) - 1))
        u = (x & ((1 << 24) - 1)).to(torch.float32) * (1.0 / (1 << 24))
        return u < keep

    def _dropped_values(self):
        """(values of A_drop's local rows, values of A_drop^T's local rows) for this step."""
        if self._row_of_nnz is None:
            deg = (self.rowptr[1:] - self.rowptr[:-1])
            self._row_of_nnz = torch.repeat_interleave(torch.arange(self.per, device=self.val.device, dtype=torch.int64), deg) + self.lo
        keep = 1.0 - self.dropout
        seed = self._drop_seed + self.step
        r, c = self._row_of_nnz, self.col
        fwd = self.val * self._keep_mask(r, c, self.n, seed, keep).to(self.val.dtype) / keep
        bwd = self.val * self._keep_mask(c, r, self.n, seed, keep).to(self.val.dtype) / keep      # entry (c, r) of A_drop, A^ symmetric
        return fwd, bwd

    # ---- collectives ------------------------------------------------------------------------
    def _all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        from ..parallel import _all_gather_into

        full = torch.empty((self.world * self.per, self.K), dtype=local.dtype, device=local.device)
        _all_gather_into(full, local.contiguous(), group=self.group)
        return full

    def propagate(self) -> torch.Tensor:
        """Own rows of mean(E^0..E^L)."""
        acc = self.E.clone()
        cur = self.E
        for _ in range(self.L):
            nxt = torch.empty_like(self.E)
            self.kern.spmm(self.rowptr, self.col, self.val, self._all_gather_rows(cur), nxt, acc)
            cur = nxt
        return acc.div_(self.L + 1)

    def _routing(self, idx: torch.Tensor):
        """Exchange plan of a list of global node ids (this rank's batch rows): owner-major order, per-peer counts, and the
        owners' view (the local rows every peer asked this rank for)."""
        from ..parallel import _a2a_single, _all_to_all_rows

        owner = torch.div(idx, self.per, rounding_mode="floor").long()
        order = torch.argsort(owner, stable=True)
        send_counts_t = torch.bincount(owner, minlength=self.world)
        recv_counts_t = torch.empty_like(send_counts_t)
        _a2a_single(recv_counts_t, send_counts_t, group=self.group)
        sc, rc = torch.stack([send_counts_t, recv_counts_t]).tolist()
        asked = _all_to_all_rows((idx[order] - owner[order] * self.per).to(torch.int32), sc, rc, self.group)
        return order, sc, rc, asked

    def _fetch_rows(self, local: torch.Tensor, route) -> torch.Tensor:
        """Rows of a row-partitioned table at the ids of `route` (`_routing`): the owners gather what they were asked for,
        one all-to-all brings the rows back ([batch rows, K] instead of an all-gather of the table)."""
        from ..parallel import _all_to_all_rows

        order, sc, rc, asked = route
        back = _all_to_all_rows(self.kern.gather(local, asked).contiguous(), rc, sc, self.group)
        out = torch.empty_like(back)
        out[order] = back
        return out

    def _route_to_owners(self, route, grads: torch.Tensor):
        """all-to-all of the row gradients to the owning ranks (ids already exchanged by `_routing`); returns local rows."""
        from ..parallel import _all_to_all_rows

        order, sc, rc, asked = route
        return asked, _all_to_all_rows(grads[order].contiguous(), sc, rc, self.group)

    def train_step(self, loss_type, users, items, items_neg=None, labels=None, lr=None):
        """One optimiser step.  Returns (loss, G): G is d loss / d E^0 as a table — or None when the optimiser step ran as the
        epilogue of the last backward product (`spmm_adam`, the default at L >= 2 on the HIP kernels): the gradient table is
        then never written.  `fuse_adam = False` on the net keeps the table for diagnostics."""
        self.step += 1
        dev = self.device
        ti = lambda x, off=0: to_device(x, dev).to(torch.int32) + off  # noqa: E731
        parts = [ti(users), ti(items, self.n_users)]
        if items_neg is not None:
            parts.append(ti(items_neg, self.n_users))
        idx = torch.cat(parts).contiguous()
        # mean(E^0 .. E^L) is only needed at the batch's rows: E^0 .. E^{L-1} are all-gathered anyway (inputs of the
        # products), so their batch rows are read from those buffers; the batch rows of E^L come from their owners by one
        # small all-reduce ([batch rows, K]) — the all-gather of the whole mean table per step is gone.  Sum order as in
        # the accumulating form: ((E^0 + E^1) + E^2) + ...
        val_f, val_b = self._dropped_values() if self.dropout > 0 else (self.val, self.val)
        route = self._routing(idx)                       # (ids only: independent of the products)
        masks = hasattr(self.kern, "row_bitmap") and self.K in (16, 32, 64, 128)
        ybm = None
        if masks and self.L >= 1:
            # the LAST layer is read at the batch's rows only: this rank computes the rows its peers asked it for
            if getattr(self, "_asked_rows", None) is None:
                self._asked_rows = self.kern.row_bitmap(self.per, dev)
            ybm = self._asked_rows.set(route[3])
        chunked = self.chunks > 1 and self.L >= 1
        if chunked:
            self._col_blocks()
            vf, vb, pos_idx = self._block_vals(val_f), self._block_vals(val_b), self._chunk_pos(idx)
        cur, rows = self.E, None
        for l in range(self.L):
            nxt = torch.empty_like(self.E)
            if chunked:
                self._chunked_spmm(cur, vf, nxt, y_rows=ybm if l == self.L - 1 else None)
                r = self.kern.gather(self._chunk_buf, pos_idx)
            else:
                full = self._all_gather_rows(cur)
                r = self.kern.gather(full, idx)
                if ybm is not None and l == self.L - 1:
                    self.kern.spmm(self.rowptr, self.col, val_f, full, nxt, None, y_rows=ybm)
                else:
                    self.kern.spmm(self.rowptr, self.col, val_f, full, nxt, None)
            rows = r if rows is None else rows.add_(r)
            cur = nxt
        last = self._fetch_rows(cur, route)
        if ybm is not None:
            ybm.clear(route[3])
        rows = last if rows is None else rows.add_(last)
        rows = rows.div_(self.L + 1)
        rows.requires_grad_(True)
        nu, ni = len(parts[0]), len(parts[1])
        loss = LightGCNNet._loss(self, loss_type, rows[:nu], rows[nu:nu + ni],
                                 rows[nu + ni:] if items_neg is not None else None, labels)
        (loss / self.world).backward()                      # mean over the global batch
        with torch.no_grad():
            loc_rows, g = self._route_to_owners(route, rows.grad)
            # G_0 = D, G_{l+1} = D + A^T G_l with D nonzero on the batch's rows only: each layer's product is written
            # plainly and the routed row gradients are scattered onto it (LightGCNNet._backprop: no clone of D per layer)
            seg = self.kern.segments(loc_rows, self.per, tag="lgcn") if loc_rows.numel() else None
            alpha = 1.0 / (self.L + 1)
            G = None
            if self.L == 0:                      # no propagation: the gradient of E^0 is the routed rows themselves
                G = torch.zeros_like(self.E)
                if seg is not None:
                    self.kern.scatter_add(G, g, seg, alpha)
            bufs = [torch.empty_like(self.E) for _ in range(min(self.L, 2))]
            hp = self.kern.adam_hp_torch(self.lr if lr is None else lr, self.step, self.epsilon, self.reg)
            fused = False
            for l in range(self.L):
                out = bufs[l % 2]
                if l == 0 and self.L >= 1:
                    # G_0 = D is nonzero on the GLOBAL batch's rows only: instead of all-gathering the table-sized D (4.5 GB
                    # received per rank at cfg 5 / 8 GPUs) every rank receives the owners' compact (row, gradient) lists
                    # (~ batch rows x K) and scatters them into a persistent zero table; the product skips its zero rows
                    Xf, xbm, ids_all = self._sparse_operand(loc_rows, g, alpha, masks)
                    if xbm is not None:
                        self.kern.spmm(self.rowptr, self.col, val_b, Xf, out, None, x_rows=xbm)
                        xbm.clear(ids_all)
                    else:
                        self.kern.spmm(self.rowptr, self.col, val_b, Xf, out, None)
                    Xf.index_fill_(0, ids_all.long(), 0.0)                        # back to zeros (the listed rows only)
                    self._D_dirty = False
                elif (l == self.L - 1 and self.L >= 2 and masks and hasattr(self.kern, "spmm_adam")
                      and getattr(self, "fuse_adam", True)):
                    # the last product's rows are d loss / d E^0 of this rank's slice: the optimiser step is its epilogue
                    if getattr(self, "_row_slot", None) is None:
                        self._row_slot = torch.full((self.per,), -1, dtype=torch.int32, device=dev)
                    Gall = self._all_gather_rows(G)              # gathered ONCE: the plain product below reuses it when the
                    fused = self.kern.spmm_adam(self.rowptr, self.col, val_b, Gall, self.E, self.m, self.v, hp,   # epilogue form
                                                self.vmax, seg, g, alpha, self._row_slot)                       # is not compiled
                    if fused:
                        G = None
                        break
                    self.kern.spmm(self.rowptr, self.col, val_b, Gall, out, None)
                elif chunked:
                    self._chunked_spmm(G, vb, out)
                else:
                    self.kern.spmm(self.rowptr, self.col, val_b, self._all_gather_rows(G), out, None)
                if seg is not None:
                    self.kern.scatter_add(out, g, seg, alpha)
                G = out
            if not fused:
                if self.vmax is not None:
                    self.kern.adam_table(self.E, self.m, self.v, G, hp, vmax=self.vmax)
                else:
                    self.kern.dense_adam(self.E, self.m, self.v, G, hp)
        return loss.detach(), G

    def _sparse_operand(self, loc_rows, g, alpha, masks):
        """The all-gathered G_0 = D without moving D: every rank contributes the (global row, routed gradient) pairs of the rows it
        OWNS (padded to the largest list: one 8-byte all-reduce), all ranks scatter alpha * g into a persistent zero table of the
        all-gathered shape.  -> (table [W * per, K], bitmap of its nonzero rows or None, the row ids to clear afterwards)."""
        from ..parallel import _all_gather_into

        dev, W = self.device, self.world
        if getattr(self, "_Dfull", None) is None:
            self._Dfull = torch.zeros((W * self.per, self.K), dtype=torch.float32, device=dev)
        elif getattr(self, "_D_dirty", False):
            # a step that raised between the scatter below and the `index_fill_` that undoes it left nonzero rows behind: the
            # unmasked product would read them (the row bitmaps have the same reset, RowBitmap._marked)
            self._Dfull.zero_()
            if getattr(self, "_D_rows", None) is not None:
                self._D_rows = None
        self._D_dirty = True
        n_loc = torch.tensor([loc_rows.numel()], dtype=torch.int64, device=dev)
        n_max = n_loc.clone()
        if W > 1:
            self.dist.all_reduce(n_max, op=self.dist.ReduceOp.MAX, group=self.group)
        n_max = max(int(n_max.item()), 1)
        # (the padding names this rank's first row with a zero gradient: adds nothing, no special id for the kernels to drop)
        ids = torch.full((n_max,), self.rank * self.per, dtype=torch.int32, device=dev)
        gr = torch.zeros((n_max, self.K), dtype=torch.float32, device=dev)
        if loc_rows.numel():
            ids[: loc_rows.numel()] = loc_rows.to(torch.int32) + self.rank * self.per
            gr[: loc_rows.numel()] = g
        ids_all = torch.empty(W * n_max, dtype=torch.int32, device=dev)
        gr_all = torch.empty((W * n_max, self.K), dtype=torch.float32, device=dev)
        _all_gather_into(ids_all, ids, group=self.group)
        _all_gather_into(gr_all, gr, group=self.group)
        seg = self.kern.segments(ids_all, W * self.per, tag="lgcn_all")
        self.kern.scatter_add(self._Dfull, gr_all, seg, alpha)
        xbm = None
        if masks:
            if getattr(self, "_D_rows", None) is None:
                self._D_rows = self.kern.row_bitmap(W * self.per, dev)
            xbm = self._D_rows.set(ids_all)
        return self._Dfull, xbm, ids_all

    @torch.no_grad()
    def embeddings(self):
        out = self._all_gather_rows(self.propagate())[: self.n]
        return out[: self.n_users].contiguous(), out[self.n_users:].contiguous()

    @torch.no_grad()
    def embeddings_sharded(self):
        """(user embeddings [n_users, K] on every rank, this rank's item rows [n_local, K], first item id, n_local):
        the propagated rows of this rank's node range, the user part all-gathered (n_users x K is small next to a
        sharded catalogue), the item part left where it is."""
        own = self.propagate()                                          # nodes [lo, hi)
        users = self._all_gather_rows(own)[: self.n_users].contiguous()
        a = max(self.lo, self.n_users)                                  # first item NODE of this rank
        n_local = max(0, self.hi - a)
        loc = own[a - self.lo: a - self.lo + n_local].contiguous() if n_local else own[:0].contiguous()
        return users, loc, a - self.n_users, n_local
