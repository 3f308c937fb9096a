"""Two-tower retrieval graph (`libreco/algorithms/two_tower.py:189-410,458-479`).

Every id / sparse-feature row a batch needs (user side, item side, negative-item side) is fetched
by ONE `lr_embed_gather_f32` launch over a concatenated global-row index matrix; the tower MLPs
and the loss run through torch autograd; the row gradients go back through
`lr_segments_build` + `lr_embed_scatter_adam_f32` (duplicates summed in a fixed order, row-wise
Adam), the dense parameters through one `lr_adam_dense_f32`.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F

from ..utils.device import to_device

from .. import ops
from ..layers import DenseParams, DenseStack, FieldTables


class TwoTowerNet:
    def __init__(self, n_users, n_items, sparse_feature_size, n_user_sparse, n_item_sparse,
                 user_dense_cols: Sequence[int], item_dense_cols: Sequence[int], n_dense_total,
                 embed_size=16, hidden_units=(128, 64, 32), use_bn=True, dropout_rate=0.0,
                 norm_embed=False, lr=1e-3, epsilon=1e-5, seed=42, device=None, margin=1.0,
                 temperature=1.0, use_correction=True, remove_accidental_hits=False, dense_adam=False):
        self.device = device or torch.device("cuda")
        self.K = embed_size
        # `item_embeds_var` has NO OOV row in the reference (two_tower.py:266-271)
        self.tables = FieldTables(n_users, n_items, sparse_feature_size, embed_size, self.device, seed,
                                  with_linear=False, item_oov_row=False)
        self.n_us, self.n_is = n_user_sparse, n_item_sparse
        self.ud_cols, self.id_cols = list(user_dense_cols), list(item_dense_cols)
        self.P = DenseParams(self.device, seed)
        if n_dense_total:
            self.P.add("embedding/dense_embeds_var", (n_dense_total, embed_size), "glorot_uniform")
        u_in = embed_size * (1 + n_user_sparse + len(self.ud_cols))
        i_in = embed_size * (1 + n_item_sparse + len(self.id_cols))
        self.user_tower = DenseStack(self.P, "user_tower", u_in, hidden_units, use_bn, dropout_rate)
        self.item_tower = DenseStack(self.P, "item_tower", i_in, hidden_units, use_bn, dropout_rate)
        self.learn_temperature = temperature <= 0.0
        if self.learn_temperature:
            self.P.add("temperature_var", (1,), "ones")
        self.P.finalize()
        self.norm_embed, self.margin, self.temperature = norm_embed, margin, temperature
        self.use_correction, self.remove_accidental_hits = use_correction, remove_accidental_hits
        self.lr, self.epsilon, self.step = lr, epsilon, 0
        self.out_dim = self.user_tower.n_out
        self.dense_adam, self._row_slot = dense_adam, None   # True: TF1 semantics (every row decays each step)

    # ---- index helpers ----------------------------------------------------------------------
    def _dev_i32(self, x):
        return to_device(x, self.device).to(torch.int32)

    def user_rows(self, users, user_sparse):
        cols = [self._dev_i32(users).view(-1, 1) + self.tables.user_off]
        if self.n_us:
            cols.append(self._dev_i32(user_sparse) + self.tables.sparse_off)
        return torch.cat(cols, dim=1)

    def item_rows(self, items, item_sparse):
        cols = [self._dev_i32(items).view(-1, 1) + self.tables.item_off]
        if self.n_is:
            cols.append(self._dev_i32(item_sparse) + self.tables.sparse_off)
        return torch.cat(cols, dim=1)

    # ---- towers -----------------------------------------------------------------------------
    def _dense_part(self, values, cols):
        if not cols:
            return None
        w = self.P["embedding/dense_embeds_var"][cols]                      # [Fd, K]
        v = to_device(values, self.device, torch.float32)
        return (v[:, :, None] * w[None]).flatten(1)                        # two_tower.py:377-398

    def _tower(self, tower, rows, dense, training):
        x = rows.flatten(1)
        if dense is not None:
            x = torch.cat([x, dense], dim=1)
        out = tower(x, training)
        return F.normalize(out, dim=1, eps=1e-12) if self.norm_embed else out   # tf.linalg.l2_normalize

    def _hp(self):
        return ops.adam_hp(self.lr, self.step, eps=self.epsilon, tf_style=True)

    def _adjusted_logits(self, ue, ie, items, corrections):
        """`adjust_logits(ue @ ie^T)` (two_tower.py:458-479) with the temperature and the logQ
        correction folded into the GEMM operands —  [ue/T | 1] @ [ie | -logQ]^T  — so the B x B
        matrix (17 GB at B = 65,536) is not re-read and re-written once per adjustment."""
        t = self.P["temperature_var"] if self.learn_temperature else self.temperature
        a = ue / t
        b = ie
        if self.use_correction and corrections is not None:
            logq = torch.log(torch.clamp(corrections, 1e-8, 1.0)).view(-1, 1)
            a = torch.cat([a, torch.ones_like(a[:, :1])], dim=1)
            b = torch.cat([b, -logq], dim=1)
        logits = a @ b.T
        if self.remove_accidental_hits:
            it = items.view(-1)
            same = (it.view(1, -1) == it.view(-1, 1)) & ~torch.eye(len(it), dtype=torch.bool, device=it.device)
            logits = torch.where(same, torch.full_like(logits, torch.finfo(torch.float32).min), logits)
        return logits

    def _softmax_ce(self, ue, ie, items, corrections, all_adjust=True):
        """Per-sample `softmax_cross_entropy` (tfops/loss.py:71-75).  Tower widths the streaming kernel takes
        (D <= 128, D % 4 == 0) never materialise the B x B logits (csrc/softmax_ce.hip); other widths use the
        B x B device GEMM below."""
        t = self.P["temperature_var"] if self.learn_temperature else self.temperature
        B, D = ue.shape
        if ue.is_cuda and ops.softmax_ce_supported(B, B, D):
            bias = ids = None
            if all_adjust and self.use_correction and corrections is not None:
                bias = -torch.log(torch.clamp(corrections, 1e-8, 1.0))
            if all_adjust and self.remove_accidental_hits:
                ids = items.view(-1).to(torch.int32).contiguous()
            return ops.softmax_ce(ue / t, ie, bias, ids, ids, 0)
        if all_adjust:
            logits = self._adjusted_logits(ue, ie, items, corrections)
        else:
            logits = (ue / t) @ ie.T
        return F.cross_entropy(logits, torch.arange(B, device=ue.device), reduction="none")

    # ---- training ---------------------------------------------------------------------------
    def train_step(self, loss_type, users, items, labels=None, items_neg=None, user_sparse=None,
                   item_sparse=None, item_sparse_neg=None, user_dense=None, item_dense=None,
                   item_dense_neg=None, corrections=None, ssl_left=None, ssl_right=None, ssl_dense=None,
                   alpha=0.2):
        self.step += 1
        t = self.tables
        u_idx = self.user_rows(users, user_sparse)
        i_idx = self.item_rows(items, item_sparse)
        blocks = [u_idx, i_idx]
        if loss_type == "max_margin":
            blocks.append(self.item_rows(items_neg, item_sparse_neg))
        n_ssl = 0
        if ssl_left is not None:
            # index j of the "ssl table" [zero row | item_embeds_var | sparse_embeds_var]
            # (two_tower.py:295-304) is global row item_off + j - 1 here (item and sparse rows are
            # adjacent); j == 0 -> out-of-range id: gathered as a zero row, dropped from the gradient
            for v in (ssl_left, ssl_right):
                j = self._dev_i32(v)
                blocks.append(torch.where(j > 0, j - 1 + t.item_off, torch.full_like(j, -1)))
            n_ssl = blocks[-1].shape[1]
        idx = torch.cat(blocks, dim=1).contiguous()
        rows = ops.embed_gather(t.embed, idx)
        rows.requires_grad_(True)
        self.P.zero_grad()
        nu, ni = u_idx.shape[1], i_idx.shape[1]
        ue = self._tower(self.user_tower, rows[:, :nu], self._dense_part(user_dense, self.ud_cols), True)
        ie = self._tower(self.item_tower, rows[:, nu:nu + ni], self._dense_part(item_dense, self.id_cols), True)
        if loss_type == "cross_entropy":
            lab = torch.as_tensor(labels, device=self.device, dtype=torch.float32)
            loss = F.binary_cross_entropy_with_logits((ue * ie).sum(1), lab)      # two_tower.py:197
        elif loss_type == "max_margin":
            ne = self._tower(self.item_tower, rows[:, nu + ni:], self._dense_part(item_dense_neg, self.id_cols), True)
            loss = F.relu(self.margin + (ue * ne).sum(1) - (ue * ie).sum(1)).mean()  # tfops/loss.py:65-68
        elif loss_type == "softmax":
            it = self._dev_i32(items)
            corr = None if corrections is None else torch.as_tensor(corrections, device=self.device, dtype=torch.float32)
            loss = self._softmax_ce(ue, ie, it, corr).mean()                       # tfops/loss.py:71-75
            if n_ssl:   # self-supervised term: two masked views through the item tower (loss.py:38-47)
                sd = self._dense_part(ssl_dense, self.id_cols) if ssl_dense is not None else None
                o = nu + ni
                sl = self._tower(self.item_tower, rows[:, o:o + n_ssl], sd, True)
                sr = self._tower(self.item_tower, rows[:, o + n_ssl:o + 2 * n_ssl], sd, True)
                loss = loss + alpha * self._softmax_ce(sl, sr, None, None, all_adjust=False).mean()
        else:
            raise ValueError(f"Unsupported `loss_type`: `{loss_type}`")
        loss.backward()
        with torch.no_grad():
            hp = self._hp()
            seg = t.segments(idx)
            if self.dense_adam:
                if self._row_slot is None:
                    self._row_slot = torch.full((t.V,), -1, dtype=torch.int32, device=self.device)
                ops.adam_dense(t.embed, t.m, t.v, hp, grows=ops.embed_segment_sum(rows.grad.view(-1, self.K), seg),
                               seg=seg, row_slot=self._row_slot)
            else:
                ops.embed_scatter_adam(t.embed, t.m, t.v, rows.grad.view(-1, self.K), seg, hp)
            self.P.adam_step(hp)
        return loss.detach()

    # ---- inference --------------------------------------------------------------------------
    @torch.no_grad()
    def embed_users(self, users, user_sparse=None, user_dense=None):
        rows = ops.embed_gather(self.tables.embed, self.user_rows(users, user_sparse).contiguous())
        return self._tower(self.user_tower, rows, self._dense_part(user_dense, self.ud_cols), False)

    @torch.no_grad()
    def embed_items(self, items, item_sparse=None, item_dense=None):
        rows = ops.embed_gather(self.tables.embed, self.item_rows(items, item_sparse).contiguous())
        return self._tower(self.item_tower, rows, self._dense_part(item_dense, self.id_cols), False)


class _AllGatherRows(torch.autograd.Function):
    """Rank-major all-gather of a [B_local, D] block with its transpose in the backward (reduce-scatter of
    the gradient: every rank contributed to the loss of every other rank's users through the in-batch
    negatives)."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist

        from ..parallel import _all_gather_into

        ctx.group, ctx.n = group, x.shape[0]
        W = dist.get_world_size(group)
        out = torch.empty((W * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_gather_into(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        from ..parallel import _reduce_scatter_sum

        out = torch.empty((ctx.n, *g.shape[1:]), dtype=g.dtype, device=g.device)
        _reduce_scatter_sum(out, g.contiguous(), group=ctx.group)
        return out, None


class ShardedTwoTowerNet:
    """Two-tower retrieval with the id / feature table ROW-SHARDED over the ranks (BASELINE cfg 4: a
    100 M x 128 item table over 8 GPUs; `libreco/algorithms/two_tower.py:189-410,458-479`).

    One step, one process per GPU, data-parallel batch of global row ids:
      ids all-to-all -> owners gather -> rows all-to-all (`ShardedFieldTables.lookup`, de-duplicated) ->
      towers (replicated dense parameters) -> loss -> row gradients summed per distinct row ->
      all-to-all to the owners -> owners sum across peers + row-wise Adam; dense gradients: one all-reduce.
    `softmax` is the GLOBAL in-batch softmax: item-tower outputs, item ids and logQ corrections are
    all-gathered, every rank scores its users against all W*B items (the gradient of the gathered block
    returns by reduce-scatter), so N ranks reproduce one rank on the concatenated batch.  BatchNorm
    statistics are those of the GLOBAL batch.

    Dense feature columns (`two_tower.py:173-187,375-398`: value x the column's `dense_embeds_var` row): the column rows are
    rows of the SAME sharded table (`dense_row0 + column`), asked for through the same exchange as every other row of the
    sample (de-duplicated: one row per column and rank), multiplied by the sample's values in front of the towers; their
    gradient returns with the row gradients.  `dropout_rate`: `tf.layers.dropout` after each hidden layer's BatchNorm
    (`layers/dense.py:44-47`); every rank draws its own masks.

    Self-supervised term (`ssl_left` / `ssl_right`; `two_tower.py:295-304,348-353`, `tfops/loss.py:38-47`): two masked views of
    a second draw of items through the item tower, in-batch softmax between them with weight `alpha`.  The views' rows travel
    in the step's ONE exchange; a masked column (the reference's constant zero row) asks for `pad_row` — a table row nothing
    else addresses — and is multiplied by zero in front of the tower, so that row collects a zero gradient and never moves.
    Like the main term the softmax is the global one: the right views are all-gathered, N ranks == one rank."""

    def __init__(self, n_rows_global, n_user_fields, n_item_fields, embed_size=16, hidden_units=(128, 64, 32),
                 use_bn=True, norm_embed=False, lr=1e-3, epsilon=1e-5, seed=42, device=None, margin=1.0,
                 temperature=1.0, use_correction=True, remove_accidental_hits=False, kern=None, group=None,
                 user_dense_cols=(), item_dense_cols=(), dense_row0=None, dropout_rate=0.0, pad_row=None):
        import torch.distributed as dist

        from ..parallel import HipKernels, ShardedFieldTables

        self.kern, self.group = kern or HipKernels(), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device or torch.device("cuda")
        self.K, self.nu, self.ni = embed_size, int(n_user_fields), int(n_item_fields)
        self.ud_cols, self.id_cols = [int(c) for c in user_dense_cols], [int(c) for c in item_dense_cols]
        if (self.ud_cols or self.id_cols) and dense_row0 is None:
            raise ValueError("dense columns need `dense_row0`, the global table row of dense column 0")
        self.dense_row0 = int(dense_row0) if dense_row0 is not None else 0
        self.pad_row = None if pad_row is None else int(pad_row)      # stands in for the ssl table's zero row (below)
        self.tables = ShardedFieldTables(n_rows_global, embed_size, self.device, self.kern, with_linear=False,
                                         group=group, seed=seed)
        self.P = DenseParams(self.device, seed)
        self.user_tower = DenseStack(self.P, "user_tower", embed_size * (self.nu + len(self.ud_cols)), hidden_units, use_bn,
                                     float(dropout_rate or 0.0))
        self.item_tower = DenseStack(self.P, "item_tower", embed_size * (self.ni + len(self.id_cols)), hidden_units, use_bn,
                                     float(dropout_rate or 0.0))
        from ..parallel import rank_average

        sync = rank_average(group)            # BatchNorm over the GLOBAL batch: N ranks == 1 rank on the concatenated batch
        self.user_tower.set_sync(sync)
        self.item_tower.set_sync(sync)
        self.P.finalize()
        self.norm_embed, self.margin, self.temperature = norm_embed, margin, temperature
        self.use_correction, self.remove_accidental_hits = use_correction, remove_accidental_hits
        self.lr, self.epsilon, self.step = lr, epsilon, 0

    def _tower(self, tower, rows, training, dense=None, dense_rows=None):
        x = rows.flatten(1)
        if dense_rows is not None:                                  # two_tower.py:375-398: value x the column's row
            vals = torch.as_tensor(dense, device=self.device, dtype=torch.float32)
            x = torch.cat([x, (vals[:, :, None] * dense_rows).flatten(1)], dim=1)
        out = tower(x, training)
        return F.normalize(out, dim=1, eps=1e-12) if self.norm_embed else out

    def _dense_ids(self, B, side=None):
        """[B, n] global rows of the dense columns' embedding rows (both sides, or one for the export)."""
        cols = (self.ud_cols + self.id_cols) if side is None else (self.ud_cols if side == "user" else self.id_cols)
        if not cols:
            return None
        ids = torch.tensor(cols, dtype=torch.int32, device=self.device) + self.dense_row0
        return ids.view(1, -1).expand(B, -1)

    def _gather_rows(self, ctx):
        B, nf = ctx.slots.shape
        return self.kern.gather(ctx.cache, ctx.slots.reshape(-1).contiguous()).view(B, nf, self.K)

    def train_step(self, loss_type, user_idx, item_idx, labels=None, item_neg_idx=None, items=None,
                   corrections=None, next_idx=None, idx=None, user_dense=None, item_dense=None, item_dense_neg=None,
                   ssl_left=None, ssl_right=None, ssl_dense=None, alpha=0.2):
        """`user_idx` [B, nu] / `item_idx` [B, ni] (/ `item_neg_idx`): GLOBAL table rows of this rank's
        samples; `items` [B]: item ids for the accidental-hit mask; `corrections` [B]: sampling probability
        Q(item) of each local item (two_tower.py:425-435).  `idx`: the caller's own concatenation
        [user_idx | item_idx (| item_neg_idx)] (int32, contiguous) — the tensor a previous step's `next_idx` named, so
        that its prefetched exchange plan is recognised; `next_idx`: the NEXT step's `idx`.  `ssl_left` / `ssl_right`
        [B, 1 + item sparse columns]: GLOBAL rows of the two views' columns, -1 where the view masks the column;
        `ssl_dense` [B, n]: the drawn items' dense values (softmax loss only)."""
        import torch.distributed as dist

        from ..parallel import _all_gather_into, allreduce_sum_

        self.step += 1
        W, dev = self.world, self.device
        n_ud, n_id = len(self.ud_cols), len(self.id_cols)
        n_ssl, ssl_keep = 0, None
        if ssl_left is not None:
            if loss_type != "softmax":
                raise ValueError("`ssl`(self-supervised learning) can only be used in `softmax` loss.")
            if self.pad_row is None:
                raise ValueError("the self-supervised views need `pad_row`: a table row that stands in for their zero row")
            if idx is not None:
                raise ValueError("with self-supervised views the net assembles the id block itself (`idx` must be None)")
            views = [torch.as_tensor(v, device=dev).to(torch.int32) for v in (ssl_left, ssl_right)]
            n_ssl = views[0].shape[1]
            ssl_keep = [(v >= 0) for v in views]
            views = [torch.where(k, v, torch.full_like(v, self.pad_row)) for v, k in zip(views, ssl_keep)]
        if idx is None:
            blocks = [user_idx, item_idx] + ([item_neg_idx] if loss_type == "max_margin" else [])
            if n_ssl:
                blocks += views
            if n_ud or n_id:
                blocks.append(self._dense_ids(user_idx.shape[0]))
            idx = torch.cat([b.to(torch.int32) for b in blocks], dim=1).contiguous()
        elif n_ud or n_id:
            raise ValueError("with dense columns the net assembles the id block itself (`idx` must be None)")
        ctx = self.tables.lookup(idx)
        rows = self._gather_rows(ctx)
        rows.requires_grad_(True)
        self.P.zero_grad()
        nu, ni = self.nu, self.ni
        d0 = rows.shape[1] - n_ud - n_id                              # the dense columns' rows close the block
        ud_rows = rows[:, d0:d0 + n_ud] if n_ud else None
        id_rows = rows[:, d0 + n_ud:] if n_id else None
        ue = self._tower(self.user_tower, rows[:, :nu], True, user_dense, ud_rows)
        ie = self._tower(self.item_tower, rows[:, nu:nu + ni], True, item_dense, id_rows)
        B = idx.shape[0]
        if loss_type == "cross_entropy":
            lab = torch.as_tensor(labels, device=dev, dtype=torch.float32)
            loss = F.binary_cross_entropy_with_logits((ue * ie).sum(1), lab)             # two_tower.py:197
            scaled = loss / W
        elif loss_type == "max_margin":
            ne = self._tower(self.item_tower, rows[:, nu + ni:nu + 2 * ni], True, item_dense_neg, id_rows)
            loss = F.relu(self.margin + (ue * ne).sum(1) - (ue * ie).sum(1)).mean()      # tfops/loss.py:65-68
            scaled = loss / W
        elif loss_type == "softmax":
            ie_all = _AllGatherRows.apply(ie, self.group)                                # [W*B, D]
            bias = None
            if self.use_correction and corrections is not None:                          # two_tower.py:458-479
                c_loc = torch.as_tensor(corrections, device=dev, dtype=torch.float32).contiguous()
                c_all = torch.empty(W * B, dtype=torch.float32, device=dev)
                _all_gather_into(c_all, c_loc, group=self.group)
                bias = -torch.log(torch.clamp(c_all, 1e-8, 1.0))
            it = it_all = None
            if self.remove_accidental_hits:
                it = torch.as_tensor(items, device=dev).to(torch.int32).contiguous()
                it_all = torch.empty(W * B, dtype=torch.int32, device=dev)
                _all_gather_into(it_all, it, group=self.group)
            # this rank's users against all W*B items; the [B, W*B] logits stay in registers (csrc/softmax_ce.hip)
            loss_sum = self.kern.softmax_ce(ue / self.temperature, ie_all, bias, it, it_all, self.rank * B).sum()
            scaled = loss_sum / (W * B)            # this rank's share of the global-batch mean
            if n_ssl:                              # tfops/loss.py:38-47: in-batch softmax between the two masked views
                o = nu + ni
                keep = [k.to(rows.dtype)[:, :, None] for k in ssl_keep]
                sl = self._tower(self.item_tower, rows[:, o:o + n_ssl] * keep[0], True, ssl_dense, id_rows)
                sr = self._tower(self.item_tower, rows[:, o + n_ssl:o + 2 * n_ssl] * keep[1], True, ssl_dense, id_rows)
                sr_all = _AllGatherRows.apply(sr, self.group)
                ssl_sum = self.kern.softmax_ce(sl / self.temperature, sr_all, None, None, None, self.rank * B).sum()
                scaled = scaled + float(alpha) * ssl_sum / (W * B)
            lt = scaled.detach().clone()
            if W > 1:
                allreduce_sum_(lt.view(1), self.group)
            loss = lt
        else:
            raise ValueError(f"Unsupported `loss_type`: `{loss_type}`")
        scaled.backward()
        with torch.no_grad():
            hp = self.kern.adam_hp(self.lr, self.step, self.epsilon)
            grows = self.kern.segment_sum(rows.grad.reshape(-1, self.K).contiguous(), ctx.seg)
            self.tables.apply_gradients(ctx, grows, None, hp)
            if W > 1:
                allreduce_sum_(self.P.grad, self.group)
            self.kern.dense_adam(self.P.flat, self.P.m, self.P.v, self.P.grad, hp)
            if next_idx is not None:
                self.tables.prefetch(next_idx)
        return loss.detach()

    @torch.no_grad()
    def embed(self, side: str, idx: torch.Tensor, dense=None) -> torch.Tensor:
        """Tower outputs of `idx` [B, nf] global rows (`side` = "user" | "item"), rows fetched from their owners; `dense`
        [B, n] the side's dense feature values when it has dense columns."""
        idx = idx.to(torch.int32)
        dids = self._dense_ids(idx.shape[0], side)
        if dids is not None:
            if dense is None:
                raise ValueError(f"the {side} tower has dense columns: pass their values")
            idx = torch.cat([idx, dids], dim=1)
        ctx = self.tables.lookup(idx.contiguous())
        rows = self._gather_rows(ctx)
        tower = self.user_tower if side == "user" else self.item_tower
        nf = self.nu if side == "user" else self.ni
        return self._tower(tower, rows[:, :nf], False, dense, rows[:, nf:] if dids is not None else None)
