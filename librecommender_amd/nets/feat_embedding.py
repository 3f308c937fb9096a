"""General feature embedding layer of the TF-graph models (FM / DeepFM / DIN):
`_build_user_item` + `compute_sparse_feats` + `compute_dense_feats`
(`algorithms/deepfm.py:175-264`, `tfops/features.py:6-148`).

Produces the field matrix  E [B, F', K]  (and the linear weights  LIN [B, F'])  in the reference's
field order  [user, item, plain sparse columns, pooled multi-sparse fields, dense columns]:

* id + plain sparse columns  -> one `lr_embed_gather_f32` over global rows,
* multi-sparse fields (combiner sum / mean / sqrtn) -> `lr_embed_bag_pool_f32` (OOV -> 0),
* dense columns -> `dense_embeds_var[f] * value` (torch, parameters live in DenseParams).

Gathered blocks are autograd leaves; after `loss.backward()` their gradients are merged into ONE
(index, gradient) stream per table and applied with `lr_segments_build` +
`lr_embed_scatter_adam_f32` (row-wise Adam) or `lr_adam_dense_f32` (TF1 dense semantics).
The all-plain, no-dense case has a faster fully fused path in `fm_nets.py`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from ..utils.device import to_device

from .. import ops
from ..layers import DenseParams, FieldTables


@dataclass
class FeatSpec:
    """What the embedding layer needs to know about a `DataInfo`."""
    n_users: int
    n_items: int
    n_sparse_cols: int = 0
    sparse_rows: int = 0
    n_dense_cols: int = 0
    combiner: str = "normal"
    field_offset: List[int] = field(default_factory=list)   # multi-sparse fields (positions in sparse cols)
    field_len: List[int] = field(default_factory=list)
    field_oov: List[int] = field(default_factory=list)      # OOV row of each field inside the sparse table

    @classmethod
    def from_data_info(cls, d, combiner="normal"):
        from ..utils.validate import sparse_feat_size

        n_sp = len(d.sparse_col.name)
        spec = cls(d.n_users, d.n_items, n_sp, int(sparse_feat_size(d) or 0) if n_sp else 0,
                   len(d.dense_col.name), combiner)
        m = d.multi_sparse_combine_info
        if m is not None and combiner in ("sum", "mean", "sqrtn"):
            spec.field_offset = [int(x) for x in m.field_offset]
            spec.field_len = [int(x) for x in m.field_len]
            spec.field_oov = [int(x) for x in m.feat_oov]
        return spec

    @property
    def pooled(self) -> bool:
        return len(self.field_offset) > 0

    @property
    def plain_cols(self) -> List[int]:
        taken = set()
        for o, n in zip(self.field_offset, self.field_len):
            taken.update(range(o, o + n))
        return [c for c in range(self.n_sparse_cols) if c not in taken]

    @property
    def n_fields(self) -> int:
        """F' = 2 + plain sparse + pooled fields + dense (`true_sparse_field_size` + ...)."""
        return 2 + len(self.plain_cols) + len(self.field_offset) + self.n_dense_cols


@dataclass
class FeatCtx:
    idx_plain: torch.Tensor
    rows: torch.Tensor                  # leaf [B, Fp, K]
    lin_rows: Optional[torch.Tensor]    # leaf [B, Fp]
    pooled_idx: List[torch.Tensor]
    pooled: List[torch.Tensor]          # leaves [B, K]
    pooled_lin: List[torch.Tensor]      # leaves [B, 1]
    sctx: Optional[object] = None       # row-sharded tables: the step's exchange (`parallel.LookupCtx`) ...
    slot_plain: Optional[torch.Tensor] = None                   # ... and the positions' rows in its cache: [B, Fp]
    slot_pooled: List[torch.Tensor] = field(default_factory=list)   # [B, n] per pooled field, -1 = OOV entry
    slot_extra: Optional[torch.Tensor] = None                   # [B, n_extra]: the caller's own rows (DIN: items of the window)


class FeatEmbedding:
    def __init__(self, spec: FeatSpec, embed_size: int, device, P: DenseParams, seed=42,
                 with_linear=True):
        self.spec, self.K, self.device, self.P = spec, embed_size, device, P
        self.tables = FieldTables(spec.n_users, spec.n_items, spec.sparse_rows, embed_size, device,
                                  seed, with_linear=with_linear)
        self.with_linear = with_linear
        if spec.n_dense_cols:
            P.add("embedding/dense_embeds_var", (spec.n_dense_cols, embed_size), "glorot_uniform")
            if with_linear:
                P.add("embedding/dense_linear_var", (spec.n_dense_cols,), "glorot_uniform")
        self._row_slot = None

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32)

    def forward(self, users, items, sparse, dense, grad=True):
        """-> (ctx, E [B,F',K], LIN [B,F'] or None)."""
        t, s = self.tables, self.spec
        cols = [self._i32(users).view(-1, 1) + t.user_off, self._i32(items).view(-1, 1) + t.item_off]
        sp = self._i32(sparse) if s.n_sparse_cols else None
        plain = s.plain_cols
        if plain:
            cols.append(sp[:, plain] + t.sparse_off)
        idx = torch.cat(cols, dim=1).contiguous()
        rows = ops.embed_gather(t.embed, idx).requires_grad_(grad)
        lin = ops.embed_gather(t.lin, idx).view(idx.shape).requires_grad_(grad) if self.with_linear else None
        parts, lparts = [rows], [lin]
        pidx, pooled, pooled_lin = [], [], []
        for o, n, oov in zip(s.field_offset, s.field_len, s.field_oov):
            fi = (sp[:, o:o + n] + t.sparse_off).contiguous()
            pidx.append(fi)
            pe = ops.embed_bag_pool(t.embed, fi, s.combiner, oov + t.sparse_off).requires_grad_(grad)
            pooled.append(pe)
            parts.append(pe.unsqueeze(1))
            if self.with_linear:
                pl = ops.embed_bag_pool(t.lin, fi, s.combiner, oov + t.sparse_off).requires_grad_(grad)
                pooled_lin.append(pl)
                lparts.append(pl)
        if s.n_dense_cols:
            dv = to_device(dense, self.device, torch.float32)
            parts.append(dv[:, :, None] * self.P["embedding/dense_embeds_var"][None])   # features.py:121-148
            if self.with_linear:
                lparts.append(dv * self.P["embedding/dense_linear_var"][None])
        E = torch.cat(parts, dim=1) if len(parts) > 1 else rows
        LIN = None
        if self.with_linear:
            LIN = torch.cat(lparts, dim=1) if len(lparts) > 1 else lin
        return FeatCtx(idx, rows, lin, pidx, pooled, pooled_lin), E, LIN

    def _streams(self, ctx: FeatCtx, extra=None, grads=None):
        """(indices [n], grads [n,K]) for the embedding table and (indices, grads [n,1]) for lin.  `grads` =
        (g_rows [B,Fp,K], g_lin_rows [B,Fp] | None, [g_pooled [B,K]], [g_pooled_lin [B,1]]): the gradients of the blocks
        `forward` produced, given explicitly (the hand-written steps) instead of read off autograd leaves."""
        t, s = self.tables, self.spec
        if grads is None:
            grads = (ctx.rows.grad, ctx.lin_rows.grad if self.with_linear else None, [pe.grad for pe in ctx.pooled],
                     [pl.grad for pl in ctx.pooled_lin])
        g_rows, g_lin, g_pool, g_pool_lin = grads
        ids = [ctx.idx_plain.reshape(-1)]
        g = [g_rows.reshape(-1, self.K)]
        gl = [g_lin.reshape(-1, 1)] if self.with_linear else []
        for fi, pg, oov in zip(ctx.pooled_idx, g_pool, s.field_oov):
            ids.append(fi.reshape(-1))
            g.append(ops.embed_bag_pool_bwd(pg.contiguous(), fi, t.V, s.combiner, oov + t.sparse_off))
        if self.with_linear:
            for fi, pl, oov in zip(ctx.pooled_idx, g_pool_lin, s.field_oov):
                gl.append(ops.embed_bag_pool_bwd(pl.contiguous(), fi, t.V, s.combiner, oov + t.sparse_off))
        if extra is not None:            # e.g. DIN attention positions: (idx [n], grads [n,K])
            ids.append(extra[0].reshape(-1))
            if self.with_linear:
                gl.append(torch.zeros((extra[0].numel(), 1), device=self.device))
            if len(extra) == 3:
                # (idx, combined buffer [n_head + n, K], n_head): the extra gradients already sit behind `n_head`
                # reserved rows of ONE buffer (written there by their kernel) — only the head is copied in
                buf, n_head = extra[1], int(extra[2])
                head = torch.cat(g) if len(g) > 1 else g[0]
                if head.shape[0] != n_head:
                    raise ValueError("reserved head of the combined gradient buffer does not match the plain streams")
                buf[:n_head].copy_(head)
                return torch.cat(ids).contiguous(), buf, (torch.cat(gl).contiguous() if gl else None)
            g.append(extra[1].reshape(-1, self.K))
        return torch.cat(ids).contiguous(), torch.cat(g).contiguous(), (torch.cat(gl).contiguous() if gl else None)

    def n_plain_positions(self, ctx: FeatCtx) -> int:
        """Rows the plain streams of `_streams` occupy in front of an extra stream."""
        return int(ctx.idx_plain.numel() + sum(fi.numel() for fi in ctx.pooled_idx))

    def apply_gradients(self, ctx: FeatCtx, hp, dense_adam=False, l2=0.0, extra=None, grads=None):
        t = self.tables
        ids, g, gl = self._streams(ctx, extra, grads)
        seg = ops.build_segments(ids, t.V)
        if not dense_adam:
            ops.embed_scatter_adam(t.embed, t.m, t.v, g, seg, hp)
            if gl is not None:
                ops.embed_scatter_adam(t.lin, t.lin_m, t.lin_v, gl, seg, hp)
            return
        if self._row_slot is None:
            self._row_slot = torch.full((t.V,), -1, dtype=torch.int32, device=self.device)
        ops.adam_dense(t.embed, t.m, t.v, hp, grows=ops.embed_segment_sum(g, seg), seg=seg,
                       row_slot=self._row_slot, l2=l2)
        if gl is not None:
            ops.adam_dense(t.lin, t.lin_m, t.lin_v, hp, grows=ops.embed_segment_sum(gl, seg), seg=seg,
                           row_slot=self._row_slot, l2=l2)

    @torch.no_grad()
    def assign_oov(self, sparse_oov_rows):
        """OOV rows := mean of the real rows (`bases/tf_base.py:310-353`): user / item tables, and
        per sparse field the mean over that field's slice."""
        t, s = self.tables, self.spec
        for tab in (t.embed, t.lin):
            if tab is None:
                continue
            tab[t.user_off + s.n_users] = tab[t.user_off: t.user_off + s.n_users].mean(dim=0)
            tab[t.item_off + s.n_items] = tab[t.item_off: t.item_off + s.n_items].mean(dim=0)
            start = 0
            for oov in (sparse_oov_rows if sparse_oov_rows is not None else []):
                oov = int(oov)
                if start >= oov:       # columns of one multi-sparse field share an OOV row
                    continue
                tab[t.sparse_off + oov] = tab[t.sparse_off + start: t.sparse_off + oov].mean(dim=0)
                start = oov + 1


class ShardedFeatEmbedding(FeatEmbedding):
    """The same layer over ROW-SHARDED tables (one process per GPU): every index stream of the batch — the
    [user, item, plain sparse] positions and the entries of the pooled multi-sparse fields — goes through ONE exchange
    (`ShardedFieldTables.lookup`: de-duplicated ids out, rows back), the gathers and the bag pooling then run on the step's
    row cache with cache slots for ids, and the (slot, gradient) streams are summed per cache row and sent to the rows'
    owners (`apply_gradients`).  Dense-column parameters live in `DenseParams` (replicated; the caller all-reduces them).
    OOV entries of a pooled field (`tfops/features.py:90-118`: they contribute nothing and do not count) are not asked
    for: their position requests the sample's user row instead and is masked out of the bag (slot -1)."""

    def __init__(self, spec: FeatSpec, embed_size: int, device, P: DenseParams, seed=42, with_linear=True, group=None,
                 kern=None):
        from ..parallel import HipKernels, ShardedFieldTables

        self.spec, self.K, self.device, self.P = spec, embed_size, device, P
        self.kern = kern or HipKernels()
        V = spec.n_users + 1 + spec.n_items + 1 + spec.sparse_rows
        self.tables = ShardedFieldTables(V, embed_size, device, self.kern, with_linear=with_linear, group=group, seed=seed)
        self.tables.set_layout(spec.n_users, spec.n_items)
        self.with_linear = with_linear
        if spec.n_dense_cols:
            P.add("embedding/dense_embeds_var", (spec.n_dense_cols, embed_size), "glorot_uniform")
            if with_linear:
                P.add("embedding/dense_linear_var", (spec.n_dense_cols,), "glorot_uniform")

    def forward(self, users, items, sparse, dense, grad=True, extra_idx=None):
        """`extra_idx` int32 [B, n_extra] GLOBAL rows the caller reads itself (DIN: the item / item-feature rows of the target
        and of the behaviour window): they ride in the same exchange; their cache slots come back as `ctx.slot_extra` and
        their gradients go in through `apply_gradients(extra=(slots, grads))`."""
        t, s, kern = self.tables, self.spec, self.kern
        cols = [self._i32(users).view(-1, 1) + t.user_off, self._i32(items).view(-1, 1) + t.item_off]
        sp = self._i32(sparse) if s.n_sparse_cols else None
        plain = s.plain_cols
        if plain:
            cols.append(sp[:, plain] + t.sparse_off)
        idx = torch.cat(cols, dim=1).contiguous()
        Fp = idx.shape[1]
        pidx, dead = [], []
        for o, n, oov in zip(s.field_offset, s.field_len, s.field_oov):
            fi = sp[:, o:o + n] + t.sparse_off
            d_ = fi == (oov + t.sparse_off)
            pidx.append(torch.where(d_, idx[:, :1].expand(-1, n), fi))      # an OOV entry asks for a row the batch holds anyway
            dead.append(d_)
        blocks = [idx] + pidx + ([extra_idx.to(torch.int32)] if extra_idx is not None else [])
        sctx = t.lookup(torch.cat(blocks, dim=1).contiguous() if len(blocks) > 1 else idx)
        slots = sctx.slots
        sl_extra = slots[:, slots.shape[1] - extra_idx.shape[1]:].contiguous() if extra_idx is not None else None
        sl_plain = slots[:, :Fp].contiguous()
        rows = kern.gather(sctx.cache, sl_plain).requires_grad_(grad)
        lin = kern.gather(sctx.lin_cache, sl_plain).view(sl_plain.shape).requires_grad_(grad) if self.with_linear else None
        parts, lparts = [rows], [lin]
        sl_pooled, pooled, pooled_lin = [], [], []
        c0 = Fp
        for (n, d_) in zip(s.field_len, dead):
            fs = torch.where(d_, torch.full_like(slots[:, c0:c0 + n], -1), slots[:, c0:c0 + n]).contiguous()
            c0 += n
            sl_pooled.append(fs)
            pe = kern.bag_pool(sctx.cache, fs, s.combiner, -1).requires_grad_(grad)
            pooled.append(pe)
            parts.append(pe.unsqueeze(1))
            if self.with_linear:
                pl = kern.bag_pool(sctx.lin_cache, fs, s.combiner, -1).requires_grad_(grad)
                pooled_lin.append(pl)
                lparts.append(pl)
        if s.n_dense_cols:
            dv = to_device(dense, self.device, torch.float32)
            parts.append(dv[:, :, None] * self.P["embedding/dense_embeds_var"][None])
            if self.with_linear:
                lparts.append(dv * self.P["embedding/dense_linear_var"][None])
        E = torch.cat(parts, dim=1) if len(parts) > 1 else rows
        LIN = None
        if self.with_linear:
            LIN = torch.cat(lparts, dim=1) if len(lparts) > 1 else lin
        return FeatCtx(idx, rows, lin, pidx, pooled, pooled_lin, sctx=sctx, slot_plain=sl_plain, slot_pooled=sl_pooled,
                       slot_extra=sl_extra), E, LIN

    def apply_gradients(self, ctx: FeatCtx, hp, dense_adam=False, l2=0.0, extra=None, grads=None):
        # (`dense_adam` / `l2` are properties of the sharded tables: their owners apply them, `ShardedFieldTables._apply_gradients`)
        s, kern, sctx = self.spec, self.kern, ctx.sctx
        U = sctx.n_rows
        if grads is None:
            grads = (ctx.rows.grad, ctx.lin_rows.grad if self.with_linear else None, [pe.grad for pe in ctx.pooled],
                     [pl.grad for pl in ctx.pooled_lin])
        g_rows, g_lin, g_pool, g_pool_lin = grads
        ids = [ctx.slot_plain.reshape(-1)]
        g = [g_rows.reshape(-1, self.K)]
        gl = [g_lin.reshape(-1, 1)] if self.with_linear else []
        for fs, pg in zip(ctx.slot_pooled, g_pool):
            ids.append(fs.reshape(-1))
            g.append(kern.bag_pool_bwd(pg.contiguous(), fs, U, s.combiner, -1))
        if self.with_linear:
            for fs, pl in zip(ctx.slot_pooled, g_pool_lin):
                gl.append(kern.bag_pool_bwd(pl.contiguous(), fs, U, s.combiner, -1))
        if extra is not None:            # (cache slots [n], gradients [n, K]) of the rows asked for through `extra_idx`
            ids.append(extra[0].reshape(-1).to(torch.int32))
            g.append(extra[1].reshape(-1, self.K))
            if self.with_linear:
                gl.append(torch.zeros((extra[0].numel(), 1), device=g[0].device))
        # every cache row is held by at least one live position (see `forward`), so the runs of the slot stream are the cache
        # rows 0 .. U-1 in order: the per-run sums ARE the per-row gradients the owners expect
        # (key bound = the stream LENGTH, fixed per batch shape — slots are < U <= length.  Keyed by U, the per-batch count of
        # distinct rows, every step allocated and kept a fresh builder: device memory grew without bound over a long run.)
        ids_all = torch.cat(ids).contiguous()
        seg = kern.segments(ids_all, int(ids_all.numel()), tag="featslots")
        grows = kern.segment_sum(torch.cat(g).contiguous(), seg)
        glin_rows = kern.segment_sum(torch.cat(gl).contiguous(), seg).reshape(-1) if self.with_linear else None
        self.tables.apply_gradients(sctx, grows, glin_rows, hp)

    @torch.no_grad()
    def assign_oov(self, sparse_oov_rows):
        self.tables.assign_oov(sparse_oov_rows)


class FMPairwise(torch.autograd.Function):
    """0.5*((sum_f e)^2 - sum_f e^2) over `lr_fm_pairwise_fwd/bwd_f32` (`kern`: the kernel provider of a row-sharded net)."""

    @staticmethod
    def forward(ctx, e, kern=None):
        e = e.contiguous()
        pair, fsum = ops.fm_pairwise_fwd(e) if kern is None else kern.fm_pairwise(e)
        ctx.save_for_backward(e, fsum)
        ctx.kern = kern
        return pair

    @staticmethod
    def backward(ctx, gpair):
        e, fsum = ctx.saved_tensors
        if ctx.kern is None:
            return ops.fm_pairwise_bwd(e, fsum, gpair.contiguous()), None
        return ctx.kern.fm_pairwise_bwd(e, fsum, gpair.contiguous()), None
