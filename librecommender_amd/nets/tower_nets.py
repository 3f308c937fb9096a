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
            logits = self._adjusted_logits(ue, ie, it, corr)
            loss = F.cross_entropy(logits, torch.arange(len(it), device=self.device))  # tfops/loss.py:71-75
            if n_ssl:   # self-supervised term: two masked views through the item tower (loss.py:38-47)
                sd = self._dense_part(ssl_dense, self.id_cols) if ssl_dense is not None else None
                o = nu + ni
                sl = self._tower(self.item_tower, rows[:, o:o + n_ssl], sd, True)
                sr = self._tower(self.item_tower, rows[:, o + n_ssl:o + 2 * n_ssl], sd, True)
                tt = self.P["temperature_var"] if self.learn_temperature else self.temperature
                ssl_logits = (sl / tt) @ sr.T                                   # adjust_logits(all_adjust=False)
                loss = loss + alpha * F.cross_entropy(ssl_logits, torch.arange(len(it), device=self.device))
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
