"""Device-resident embedding variables of the feature models (FM / DeepFM / DIN).

The reference keeps one TF variable per id space (``user_embeds_var [U+1,K]``,
``item_embeds_var [N+1,K]``, ``sparse_embeds_var [S,K]`` plus their ``*_linear_var`` twins —
algorithms/deepfm.py:181-234, tfops/features.py:6-44).  Here they are *views into one HBM
allocation* addressed by global row ids

    [ user rows 0..U | item rows 0..N | sparse rows 0..S-1 ]

so that one fused gather (one wavefront per sample), one segment build and one fused
backward+Adam launch serve every field of a batch.  Adam moments share the layout.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import ops


def glorot_uniform_(t: torch.Tensor, shape, gen: torch.Generator) -> None:
    """tf.glorot_uniform_initializer on a variable of `shape` (fan_in/out = the two dims;
    rank-1 variables use fan_in = fan_out = len) — algorithms/deepfm.py:185 etc."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    t.uniform_(-limit, limit, generator=gen)


class FieldTables:
    """Concatenated user/item/sparse embedding tables (+ optional linear twins) with Adam state."""

    def __init__(self, n_users: int, n_items: int, sparse_feature_size: int, embed_size: int,
                 device: torch.device, seed: int = 42, with_linear: bool = True,
                 item_oov_row: bool = True, sparse_offsets=None):
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.sparse_size = int(sparse_feature_size or 0)
        self.K = int(embed_size)
        self.device = device
        u_rows = self.n_users + 1
        i_rows = self.n_items + (1 if item_oov_row else 0)
        self.user_off, self.item_off, self.sparse_off = 0, u_rows, u_rows + i_rows
        self.V = u_rows + i_rows + self.sparse_size
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.embed = torch.empty((self.V, self.K), dtype=torch.float32, device=device)
        self.lin = torch.empty((self.V, 1), dtype=torch.float32, device=device) if with_linear else None
        for off, rows in ((self.user_off, u_rows), (self.item_off, i_rows),
                          (self.sparse_off, self.sparse_size)):
            if rows == 0:
                continue
            glorot_uniform_(self.embed[off:off + rows], (rows, self.K), gen)
            if with_linear:
                shape = (rows, 1) if off != self.sparse_off else (rows,)
                glorot_uniform_(self.lin[off:off + rows], shape, gen)
        self.m = torch.zeros_like(self.embed)
        self.v = torch.zeros_like(self.embed)
        if with_linear:
            self.lin_m = torch.zeros_like(self.lin)
            self.lin_v = torch.zeros_like(self.lin)
        self._seg_builder: Optional[ops.SegmentBuilder] = None
        # global row range of every field [user, item, sparse columns...] (for lr_fm_field_stats_f32);
        # known when the caller passes the per-column offsets of the sparse table (all plain columns)
        self.field_row_start = None
        if sparse_offsets is not None or self.sparse_size == 0:
            offs = [] if sparse_offsets is None else [int(o) for o in sparse_offsets]
            starts = [self.user_off, self.item_off] + [self.sparse_off + o for o in offs] + [self.V]
            # columns of one multi-sparse field under the "normal" combiner share an offset: their
            # ranges are not disjoint, so the per-field kernels (statistics, LDS segment sort) do not apply
            if all(b > a for a, b in zip(starts[:-1], starts[1:])):
                self.field_row_start = torch.tensor(starts, dtype=torch.int32, device=device)

    # ---- views named like the reference's variables (save/load, OOV assignment) ----------
    def variable(self, name: str) -> torch.Tensor:
        u, i, s = self.user_off, self.item_off, self.sparse_off
        spans = {"user": (u, i), "item": (i, s), "sparse": (s, self.V)}
        kind, which = name.split("_", 1)
        lo, hi = spans[kind]
        base = self.embed if which == "embeds_var" else self.lin
        return base[lo:hi]

    def global_idx(self, users: torch.Tensor, items: torch.Tensor,
                   sparse_indices: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B, 2 + Fs] int32 global rows: the field order of deepfm.py:210-214 (user, item, sparse)."""
        cols = [users.to(torch.int32).view(-1, 1) + self.user_off,
                items.to(torch.int32).view(-1, 1) + self.item_off]
        if sparse_indices is not None:
            cols.append(sparse_indices.to(torch.int32) + self.sparse_off)
        return torch.cat(cols, dim=1).contiguous()

    def segments(self, idx: torch.Tensor) -> ops.Segments:
        n = idx.numel()
        if self._seg_builder is None or self._seg_builder.n_max < n:
            self._seg_builder = ops.SegmentBuilder(n, self.V, self.device)
        return self._seg_builder.build(idx.reshape(-1))

    def bytes(self) -> int:
        n = self.embed.numel() * 3
        if self.lin is not None:
            n += self.lin.numel() * 3
        return n * 4
