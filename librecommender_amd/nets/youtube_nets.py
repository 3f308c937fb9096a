"""YouTubeRetrieval graph (`libreco/algorithms/youtube_retrieval.py:165-262`, loss `training/tf_trainer.py:133-245`).

User vector = dense_nn([sqrtn-pooled history rows of `seq_embeds_var` ; user sparse feats ; user dense feats]) with
`embed_size` outputs; the classes are the rows of `item_embeds_var` (+ `item_bias_var`), trained with a sampled
softmax / NCE over the true item and `num_sampled` uniformly drawn items shared by the batch.

One HBM allocation holds every row table (as the feature models' `FieldTables` do):

    [ seq rows 0..N-1 | pad row | item rows 0..N-1 | sparse rows 0..S-1 ]

The history pooling is `lr_embed_bag_pool_f32` (combiner sqrtn, pad id -> 0, an empty history -> 0-vector, which is
what `tf.nn.safe_embedding_lookup_sparse` returns — layers/embedding.py:26-51), multi-sparse user fields pool through
the same kernel, plain columns and the (true + sampled) class rows come from `lr_embed_gather_f32`; all row gradients
go back as ONE (index, gradient) stream through `lr_segments_build` + `lr_embed_scatter_adam_f32` (row-wise Adam) or
`lr_adam_dense_f32` (TF1 dense semantics).  The B x (1 + num_sampled) logits are a device GEMM: they are small
(num_sampled defaults to the batch size), unlike the in-batch softmax of TwoTower that `csrc/softmax_ce.hip` streams.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from .. import ops
from ..layers import DenseParams, DenseStack
from ..layers.embedding import glorot_uniform_
from ..utils.device import to_device
from .feat_embedding import FeatSpec

FLOAT_MAX = float(torch.finfo(torch.float32).max)          # TF masks accidental hits with -FLOAT_MAX (nn_impl.py)


class RetrievalTables:
    """`embedding/seq_embeds_var [N,K]`, `embedding/item_embeds_var [N,K]` (youtube_retrieval.py:193-205,243-257: no OOV
    rows) and `embedding/sparse_embeds_var [S,K]` as views of one allocation, with Adam moments in the same layout."""

    def __init__(self, n_items, sparse_rows, K, device, seed):
        self.n_items, self.sparse_size, self.K, self.device = int(n_items), int(sparse_rows or 0), int(K), device
        self.seq_off, self.pad_row = 0, self.n_items
        self.item_off = self.n_items + 1
        self.sparse_off = self.item_off + self.n_items
        self.V = self.sparse_off + self.sparse_size
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.embed = torch.zeros((self.V, K), dtype=torch.float32, device=device)
        for off, rows in ((self.seq_off, self.n_items), (self.item_off, self.n_items), (self.sparse_off, self.sparse_size)):
            if rows:
                glorot_uniform_(self.embed[off:off + rows], (rows, K), gen)
        self.m, self.v = torch.zeros_like(self.embed), torch.zeros_like(self.embed)
        self._seg_builder = None

    def variable(self, name):
        spans = {"seq_embeds_var": (self.seq_off, self.n_items), "item_embeds_var": (self.item_off, self.n_items),
                 "sparse_embeds_var": (self.sparse_off, self.sparse_size)}
        off, n = spans[name]
        return self.embed[off:off + n]

    def segments(self, ids):
        n = ids.numel()
        if self._seg_builder is None or self._seg_builder.n_max < n:
            self._seg_builder = ops.SegmentBuilder(n, self.V, self.device)
        return self._seg_builder.build(ids.reshape(-1))


class YouTubeRetrievalNet:
    def __init__(self, n_items, spec: FeatSpec, embed_size=16, hidden_units: Sequence[int] = (128, 64, 16), use_bn=True,
                 dropout_rate=0.0, norm_embed=False, max_seq_len=10, lr=1e-3, epsilon=1e-5, seed=42, device=None,
                 dense_adam=False, loss_type="sampled_softmax", num_sampled=None, reg=None, batch_size=None):
        self.device = device or torch.device("cuda")
        self.n_items, self.K, self.L, self.spec = int(n_items), embed_size, max_seq_len, spec
        self.tables = RetrievalTables(n_items, spec.sparse_rows, embed_size, self.device, seed)
        self.P = DenseParams(self.device, seed)
        self.P.add("embedding/item_bias_var", (self.n_items,), "zeros")
        if spec.n_dense_cols:
            self.P.add("embedding/dense_embeds_var", (spec.n_dense_cols, embed_size), "glorot_uniform")
        n_feat = len(spec.plain_cols) + len(spec.field_offset) + spec.n_dense_cols
        if hidden_units[-1] != embed_size:
            raise ValueError("the last layer of the user MLP must have `embed_size` units (youtube_retrieval.py:145)")
        self.mlp = DenseStack(self.P, "mlp", (1 + n_feat) * embed_size, hidden_units, use_bn, dropout_rate)
        self.P.finalize()
        self.norm_embed, self.loss_type, self.num_sampled = norm_embed, loss_type, num_sampled
        # `num_sampled_per_batch=None` means the CONFIGURED batch size in the reference (youtube_retrieval.py:150-152):
        # the short last batch of an epoch still draws `batch_size` classes
        self.batch_size = batch_size
        # tf.keras.regularizers.l2(reg) on the embedding variables adds 2 * reg * w to EVERY row's gradient each step
        # (tfops/configs.py:20-26): representable only with the dense TF1 update
        self.reg = float(reg or 0.0)
        if self.reg and not dense_adam:
            raise ValueError("`reg` regularises every embedding row each step (tf.keras.regularizers.l2 on the variables): "
                             "use `dense_adam=True` with it; the row-wise Adam on touched rows cannot represent that term")
        self.lr, self.epsilon, self.step, self.dense_adam, self._row_slot = lr, epsilon, 0, dense_adam, None
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32).contiguous()

    def _hp(self):
        return ops.adam_hp(self.lr, self.step, eps=self.epsilon, tf_style=True)

    # ---- user side ------------------------------------------------------------------------------
    def _user_inputs(self, seqs, sparse, dense, grad):
        """-> (leaves, MLP input).  `leaves` = [(ids [n], leaf tensor, bag spec or None)] for the gradient streams."""
        t, s = self.tables, self.spec
        leaves, parts = [], []
        sq = self._i32(seqs) + t.seq_off                                     # pad id N -> the pad row
        pooled = ops.embed_bag_pool(t.embed, sq, "sqrtn", t.pad_row).requires_grad_(grad)
        leaves.append((sq, pooled, ("sqrtn", t.pad_row)))
        parts.append(pooled)
        if s.n_sparse_cols:
            sp = self._i32(sparse)
            plain = s.plain_cols
            if plain:
                idx = (sp[:, plain] + t.sparse_off).contiguous()
                rows = ops.embed_gather(t.embed, idx).requires_grad_(grad)
                leaves.append((idx, rows, None))
                parts.append(rows.flatten(1))
            for o, n, oov in zip(s.field_offset, s.field_len, s.field_oov):  # tfops/features.py:47-118
                fi = (sp[:, o:o + n] + t.sparse_off).contiguous()
                pe = ops.embed_bag_pool(t.embed, fi, s.combiner, oov + t.sparse_off).requires_grad_(grad)
                leaves.append((fi, pe, (s.combiner, oov + t.sparse_off)))
                parts.append(pe)
        if s.n_dense_cols:
            dv = to_device(dense, self.device, torch.float32)
            parts.append((dv[:, :, None] * self.P["embedding/dense_embeds_var"][None]).flatten(1))
        return leaves, torch.cat(parts, dim=1)

    def _user_vec(self, x, training):
        out = self.mlp(x, training)
        return F.normalize(out, dim=1, eps=1e-12) if self.norm_embed else out

    @torch.no_grad()
    def embed_users(self, seqs, sparse=None, dense=None):
        _, x = self._user_inputs(seqs, sparse, dense, False)
        return self._user_vec(x, False)

    @torch.no_grad()
    def item_matrix(self):
        """[N, K] class rows (normalised if `norm_embed`) and [N] biases (dyn_embed_base.py:240-269)."""
        w = self.tables.variable("item_embeds_var")
        return (F.normalize(w, dim=1, eps=1e-12) if self.norm_embed else w), self.P["embedding/item_bias_var"].detach()

    # ---- loss -----------------------------------------------------------------------------------
    def draw_sampled(self, S):
        """`uniform_candidate_sampler(unique=True, range_max=n_items)`: S distinct classes shared by the batch."""
        return torch.randperm(self.n_items, device=self.device, generator=self.gen)[:S].to(torch.int32)

    def sampled_loss(self, ue, w_true, b_true, w_s, b_s, items, sampled):
        """`tf.nn.sampled_softmax_loss` / `tf.nn.nce_loss` (num_true = 1, remove_accidental_hits, subtract_log_q):
        logits [true | sampled] - log(expected count), sampled classes equal to the row's label masked with
        -FLOAT_MAX; softmax CE against column 0, or the sum of sigmoid CEs with labels [1, 0, ...]."""
        S = w_s.shape[0]
        true_logit = (ue * w_true).sum(1) + b_true
        samp = ue @ w_s.T + b_s[None, :]
        # inclusion probability of a class in S distinct uniform draws; TF evaluates 1 - (1 - 1/N)^tries with the
        # sampler's own (random) retry count, which only shifts every logit of a row by one constant
        log_q = math.log(min(1.0, S / self.n_items))
        hit = sampled.view(1, -1) == items.view(-1, 1)
        samp = torch.where(hit, torch.full_like(samp, -FLOAT_MAX), samp - log_q)
        logits = torch.cat([(true_logit - log_q).view(-1, 1), samp], dim=1)
        if self.loss_type == "sampled_softmax":
            return F.cross_entropy(logits, torch.zeros(len(logits), dtype=torch.long, device=logits.device))
        if self.loss_type == "nce":
            lab = torch.zeros_like(logits)
            lab[:, 0] = 1.0
            return F.binary_cross_entropy_with_logits(logits, lab, reduction="none").sum(1).mean()
        raise ValueError("Loss type must either be `nce` or `sampled_softmax`")

    def train_step(self, items, seqs, sparse=None, dense=None, sampled: Optional[torch.Tensor] = None):
        self.step += 1
        t = self.tables
        it = self._i32(items)
        B = len(it)
        S = self.num_sampled if self.num_sampled and self.num_sampled > 0 else (self.batch_size or B)
        S = min(S, self.n_items)
        if sampled is None:
            sampled = self.draw_sampled(S)
        sampled = self._i32(sampled)
        leaves, x = self._user_inputs(seqs, sparse, dense, True)
        cls_idx = (torch.cat([it, sampled]) + t.item_off).view(-1, 1).contiguous()
        cls_rows = ops.embed_gather(t.embed, cls_idx).view(-1, self.K).requires_grad_(True)
        self.P.zero_grad()
        ue = self._user_vec(x, True)
        w = F.normalize(cls_rows, dim=1, eps=1e-12) if self.norm_embed else cls_rows
        bias = self.P["embedding/item_bias_var"]
        loss = self.sampled_loss(ue, w[:B], bias[it.long()], w[B:], bias[sampled.long()], it, sampled)
        loss.backward()
        with torch.no_grad():
            ids, grads = [cls_idx.view(-1)], [cls_rows.grad]
            for idx, leaf, bag in leaves:
                ids.append(idx.reshape(-1))
                if bag is None:
                    grads.append(leaf.grad.reshape(-1, self.K))
                else:                                                       # per-position gradients, 0 at pads
                    grads.append(ops.embed_bag_pool_bwd(leaf.grad.contiguous(), idx, t.V, bag[0], bag[1]))
                    ids[-1] = torch.where(ids[-1] == bag[1], torch.full_like(ids[-1], -1), ids[-1])
            ids, g = torch.cat(ids).contiguous(), torch.cat(grads).contiguous()
            hp = self._hp()
            seg = t.segments(ids)
            if self.dense_adam:
                if self._row_slot is None:
                    self._row_slot = torch.full((t.V,), -1, dtype=torch.int32, device=self.device)
                ops.adam_dense(t.embed, t.m, t.v, hp, grows=ops.embed_segment_sum(g, seg), seg=seg, row_slot=self._row_slot,
                               l2=self.reg)
            else:
                ops.embed_scatter_adam(t.embed, t.m, t.v, g, seg, hp)
            self.P.adam_step(hp)
        return loss.detach()
