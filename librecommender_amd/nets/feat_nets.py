"""FM / DeepFM / DIN graphs over the general feature embedding layer (dense columns, pooled
multi-sparse fields, item side features).  The all-plain case of FM / DeepFM is served by the
fully fused nets in `fm_nets.py`; these nets share their dense layers and semantics."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

from ..utils.device import to_device

from .. import ops
from ..layers import DenseParams, DenseStack, TFBatchNorm, TFDense
from .feat_embedding import FeatEmbedding, FeatSpec, FMPairwise
from .fm_nets import _FieldNet


class _FeatNet:
    with_linear = True

    def __init__(self, spec: FeatSpec, embed_size, lr, epsilon, seed, device, dense_adam, reg, group=None, kern=None,
                 sharded=False):
        """`sharded` (one process per GPU): the tables are row-sharded over the ranks of `group`
        (`ShardedFeatEmbedding`), the batch is data-parallel (this rank's samples), dense parameters are replicated with one
        all-reduce of their gradients per step, the loss is the mean over the global batch (local mean / W) and every
        BatchNorm normalises with the global batch's statistics — N ranks take the step one rank takes on the concatenated
        batch."""
        self.device = device or torch.device("cuda")
        self.spec, self.K = spec, embed_size
        self.P = DenseParams(self.device, seed)
        self.kern, self.group, self.world, self._sync = None, group, 1, None
        if sharded:
            import torch.distributed as dist

            from ..parallel import HipKernels, rank_average
            from .feat_embedding import ShardedFeatEmbedding

            self.kern = kern or HipKernels()
            self.world = dist.get_world_size(group)
            self._sync = rank_average(group)
            self.emb = ShardedFeatEmbedding(spec, embed_size, self.device, self.P, seed, self.with_linear, group, self.kern)
        else:
            self.emb = FeatEmbedding(spec, embed_size, self.device, self.P, seed, self.with_linear)
        self.lr, self.epsilon, self.dense_adam, self.reg = lr, epsilon, dense_adam, reg or 0.0
        self.step = 0

    @property
    def tables(self):
        return self.emb.tables

    def _hp(self):
        if self.kern is not None:
            return self.kern.adam_hp(self.lr, self.step, self.epsilon)
        return ops.adam_hp(self.lr, self.step, eps=self.epsilon, tf_style=True)

    def _labels(self, labels):
        return to_device(labels, self.device, torch.float32)

    def _finish(self, ctx, loss, extra=None):
        (loss if self.world == 1 else loss / self.world).backward()
        with torch.no_grad():
            hp = self._hp()
            self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, extra() if callable(extra) else extra)
            if self.kern is not None:
                from ..parallel import allreduce_sum_

                allreduce_sum_(self.P.grad, self.group)
                self.kern.dense_adam(self.P.flat, self.P.m, self.P.v, self.P.grad, hp)
            else:
                self.P.adam_step(hp)
        return loss.detach()


class FeatFMNet(_FeatNet):
    """algorithms/fm.py:140-170."""

    def __init__(self, spec, embed_size=16, use_bn=True, lr=1e-3, epsilon=1e-5, seed=42, device=None,
                 dense_adam=False, reg=None, **shard):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg, **shard)
        self.linear = TFDense(self.P, "linear", spec.n_fields, 1)
        self.bn = TFBatchNorm(self.P, "bn", embed_size) if use_bn else None
        self.pair_dense = TFDense(self.P, "pair", embed_size, 1)
        self.P.finalize()
        if self.bn is not None and self._sync is not None:
            self.bn.sync = self._sync

    def _out(self, E, LIN, training):
        pair = FMPairwise.apply(E, self.kern)
        x = self.bn(pair, training) if self.bn is not None else pair
        return (self.linear(LIN) + F.elu(self.pair_dense(x))).squeeze(1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, **_):
        _, E, LIN = self.emb.forward(users, items, sparse, dense, grad=False)
        return self._out(E, LIN, False)

    def train_step(self, users, items, labels, sparse=None, dense=None, loss_type="cross_entropy", **_):
        self.step += 1
        ctx, E, LIN = self.emb.forward(users, items, sparse, dense)
        self.P.zero_grad()
        loss = _FieldNet.loss_fn(self._out(E, LIN, True), self._labels(labels), loss_type)
        return self._finish(ctx, loss)


class FeatDeepFMNet(_FeatNet):
    """algorithms/deepfm.py:143-173."""

    def __init__(self, spec, embed_size=16, hidden_units=(128, 64, 32), use_bn=True, dropout_rate=0.0,
                 lr=1e-3, epsilon=1e-5, seed=42, device=None, dense_adam=False, reg=None, **shard):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg, **shard)
        F_ = spec.n_fields
        self.linear = TFDense(self.P, "linear", F_, 1)
        self.mlp = DenseStack(self.P, "mlp", F_ * embed_size, hidden_units, use_bn, dropout_rate)
        self.out = TFDense(self.P, "out", 1 + embed_size + self.mlp.n_out, 1)
        self.P.finalize()
        if self._sync is not None:
            self.mlp.set_sync(self._sync)
        # Round 4: the reference's usual DeepFM data — plain sparse + multi-sparse (pooled) + dense columns
        # (tests/conftest.py:64-128 of the reference) — no longer drops to autograd + library GEMMs.  The field matrix
        # E [B, F', K] is assembled once (gather, bag-pool, dense-column products: tfops/features.py:47-148) and handed,
        # re-cut into 32-wide rows, to the MFMA first-layer kernels (`BlockFirstLayer`, BatchNorm folded), the rest of
        # dense_nn / output layer / loss and their backward run in csrc/deepfm_tail.hip, and d loss / d E goes back to the
        # tables through the same (index, gradient) streams as before.  Needs: cross-entropy, row-wise Adam, F' * K a multiple
        # of 32, compiled widths.  Dropout (layers/dense.py:44-47) is a counter-based mask inside the tail kernels.
        from ..layers.dense import BlockFirstLayer
        from ..layers.tail import DeepFMTail

        H1_ = hidden_units[0] if len(hidden_units) else 0
        self.block_l1 = bool(len(hidden_units) >= 2 and not dense_adam and self.kern is None     # (sharded: the autograd step)
                             and (F_ * embed_size) % 32 == 0 and BlockFirstLayer.supported(32, H1_)
                             and DeepFMTail.supported(self.mlp))
        self._blk = {}

    @torch.no_grad()
    def _block_step(self, users, items, labels, sparse, dense):
        """One training step without autograd (see `block_l1` in `__init__`); same arithmetic as `train_step`'s torch
        path up to f32 summation order (tests/test_feat_block_gpu.py)."""
        from ..layers.dense import BlockFirstLayer
        from ..layers.tail import DeepFMTail

        P, mlp, s, K = self.P, self.mlp, self.spec, self.K
        ctx, E, LIN = self.emb.forward(users, items, sparse, dense, grad=False)
        E = E.contiguous()
        B, F_ = E.shape[0], E.shape[1]
        st = self._blk.get(B)
        if st is None:
            Pn = F_ * K // 32
            st = self._blk[B] = dict(
                l1=BlockFirstLayer(P, mlp.bn_in, mlp.layers[0], Pn, 32, B, self.device, layout="rowmajor"),
                tail=DeepFMTail(P, mlp, self.linear, self.out, F_, K, self.device),
                gbuf=torch.empty((B * Pn + 1, 32), dtype=torch.float32, device=self.device))
        pair, fsum = ops.fm_pairwise_fwd(E)
        z1 = st["l1"].forward(E.view(B, F_ * K))
        loss, gl, gz1, sgz1 = st["tail"].run(z1, pair, LIN.contiguous(), self._labels(labels))
        st["l1"].backward(gz1, sgz1, st["gbuf"])
        G = st["gbuf"][: B * (F_ * K // 32)].view(B, F_, K)              # d loss / d E through the MLP (BatchNorm terms included)
        w_out = P[self.out.w]
        gpair = (gl[:, None] * w_out[1:1 + K, 0][None, :]).contiguous()   # deepfm.py:171-172: the FM term feeds one Dense(1)
        ops.fm_pairwise_bwd(E, fsum, gpair, ge=G)                          # G += gpair * (fsum - E)
        glin = gl[:, None] * (w_out[0, 0] * P[self.linear.w][:, 0])[None, :]                       # [B, F']
        Fp, nq = ctx.idx_plain.shape[1], len(ctx.pooled_idx)
        grads = (G[:, :Fp].contiguous(), glin[:, :Fp].contiguous(), [G[:, Fp + q].contiguous() for q in range(nq)],
                 [glin[:, Fp + q:Fp + q + 1].contiguous() for q in range(nq)])
        if s.n_dense_cols:                                                 # features.py:121-148: dense_embeds_var[f] * value
            dv = to_device(dense, self.device, torch.float32)
            d0 = Fp + nq
            P["embedding/dense_embeds_var"].grad.copy_(torch.einsum("bf,bfk->fk", dv, G[:, d0:]))
            P["embedding/dense_linear_var"].grad.copy_((dv * glin[:, d0:]).sum(0))
        hp = self._hp()
        self.emb.apply_gradients(ctx, hp, False, 0.0, None, grads)
        P.adam_step(hp)
        return loss

    def _out(self, E, LIN, training):
        concat = torch.cat([self.linear(LIN), FMPairwise.apply(E, self.kern), self.mlp(E.flatten(1), training)], dim=1)
        return self.out(concat).squeeze(1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, **_):
        _, E, LIN = self.emb.forward(users, items, sparse, dense, grad=False)
        return self._out(E, LIN, False)

    def train_step(self, users, items, labels, sparse=None, dense=None, loss_type="cross_entropy", **_):
        self.step += 1
        if self.block_l1 and loss_type == "cross_entropy":
            return self._block_step(users, items, labels, sparse, dense)
        ctx, E, LIN = self.emb.forward(users, items, sparse, dense)
        self.P.zero_grad()
        loss = _FieldNet.loss_fn(self._out(E, LIN, True), self._labels(labels), loss_type)
        return self._finish(ctx, loss)


def dot_attention_torch(q, keys, lens):
    """`tf_attention` (layers/attention.py:5-25): tf.keras.layers.Attention(use_scale=False) with a
    value mask — scores q.k, masked positions pushed down by 1e9 (keras `_apply_scores`), softmax,
    weighted sum of the keys."""
    L = keys.shape[1]
    s = torch.einsum("bk,blk->bl", q, keys)
    mask = torch.arange(L, device=keys.device)[None, :] < lens[:, None]
    s = s - 1e9 * (~mask).to(s.dtype)
    return (torch.softmax(s, dim=1)[:, None, :] @ keys).squeeze(1)


def din_attention_torch(q, keys, lens, W1, b1, W2, b2):
    """`din_attention` (layers/attention.py:28-64) in torch ops — used only when the key width
    K' = K*(1+item feats) is not one the fused kernel is compiled for."""
    B, L, Kp = keys.shape
    qt = q[:, None, :].expand(-1, L, -1)
    h = torch.sigmoid(torch.cat([qt, keys, qt - keys, qt * keys], dim=2) @ W1 + b1)
    s = ((h @ W2.view(-1, 1)).squeeze(-1) + b2) * (Kp ** -0.5)
    mask = torch.arange(L, device=keys.device)[None, :] < lens[:, None]
    s = torch.where(mask, s, torch.full_like(s, -(2.0 ** 32) + 1))
    return (torch.softmax(s, dim=1)[:, None, :] @ keys).squeeze(1)


class _DinDenseAttention(torch.autograd.Function):
    """`din_attention` on materialised (query, keys) rows through `lr_din_attn_dense_fwd/bwd_f32` — the same MFMA
    kernels as the fused id-gathering form, for keys that are concatenations of several tables' rows."""

    @staticmethod
    def forward(ctx, q, keys, lens, W1, b1, W2, b2):
        out, attn = ops.din_attn_dense_fwd(q, keys, lens, W1, b1, W2, b2)
        ctx.save_for_backward(q, keys, lens, W1, b1, W2, b2, attn)
        return out

    @staticmethod
    def backward(ctx, gout):
        q, keys, lens, W1, b1, W2, b2, attn = ctx.saved_tensors
        gq, gkey, gW1, gb1, gW2, gb2 = ops.din_attn_dense_bwd(q, keys, lens, W1, b1, W2, b2, attn, gout.contiguous())
        return gq, gkey, None, gW1, gb1, gW2, gb2


DIN_KERNEL_WIDTHS = (16, 32, 64, 128)       # key widths csrc/din_attention.hip is compiled for


def din_attention_dense(q, keys, lens, W1, b1, W2, b2):
    """`din_attention` (layers/attention.py:28-64) for key width K' = K * (1 + item feature columns): the rows are
    zero-padded to the next compiled width (zero rows of W1 for the padded dims; W2, b2 rescaled so the kernel's
    1/sqrt(width) stays the reference's 1/sqrt(K')), the pooled output cut back to K'.  None when K' > 128."""
    Kp = keys.shape[2]
    Kk = next((k for k in DIN_KERNEL_WIDTHS if k >= Kp), None)
    if Kk is None:
        return None
    W2 = W2.reshape(-1)
    if Kk != Kp:
        pad = Kk - Kp
        q, keys = F.pad(q, (0, pad)), F.pad(keys, (0, pad))
        W1 = F.pad(W1.view(4, Kp, -1), (0, 0, 0, pad)).reshape(4 * Kk, -1)
        c = (Kk / Kp) ** 0.5
        W2, b2 = W2 * c, b2 * c
    att = _DinDenseAttention.apply(q.contiguous(), keys.contiguous(), lens.to(torch.int32).contiguous(), W1.contiguous(),
                                   b1.contiguous(), W2.contiguous(), b2.contiguous())
    return att[:, :Kp]


class FeatDINNet(_FeatNet):
    """algorithms/din.py:165-250: [user, item, sparse, dense] embeddings + attention over the
    behaviour sequence -> MLP -> Dense(1).  No linear tables."""
    with_linear = False

    def __init__(self, spec, embed_size=16, hidden_units=(128, 64, 32), use_bn=True, dropout_rate=0.0,
                 max_seq_len=10, item_sparse_unique=None, item_dense_unique=None,
                 item_dense_cols: Sequence[int] = (), lr=1e-3, epsilon=1e-5, seed=42, device=None,
                 dense_adam=False, reg=None, use_tf_attention=False, fused_step=True, graph_step=True, **shard):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg, **shard)
        self.L = max_seq_len
        dev = self.device
        self.item_sparse = None if item_sparse_unique is None else torch.as_tensor(item_sparse_unique, device=dev).to(torch.int32)
        self.item_dense = None if item_dense_unique is None else torch.as_tensor(item_dense_unique, device=dev, dtype=torch.float32)
        self.item_dense_cols = list(item_dense_cols)
        n_is = 0 if self.item_sparse is None else self.item_sparse.shape[1]
        n_id = 0 if self.item_dense is None else self.item_dense.shape[1]
        self.Kp = embed_size * (1 + n_is + n_id)                    # width of an item's "concat" features
        self.pure = n_is == 0 and n_id == 0
        self.use_tf_attention = bool(use_tf_attention)       # plain dot-product attention: no MLP, torch path
        if not self.use_tf_attention:    # the reference graph has no attention MLP variables in that mode
            self.P.add("attention/attention_layer1/kernel", (4 * self.Kp, 16), "glorot_uniform")
            self.P.add("attention/attention_layer1/bias", (16,), "zeros")
            self.P.add("attention/attention_layer2/kernel", (16, 1), "glorot_uniform")
            self.P.add("attention/attention_layer2/bias", (1,), "zeros")
        self.mlp = DenseStack(self.P, "mlp", spec.n_fields * embed_size + self.Kp, hidden_units, use_bn, dropout_rate)
        self.out = TFDense(self.P, "out", self.mlp.n_out, 1)
        self.P.finalize()
        self.fused = self.pure and embed_size in (16, 32, 64, 128) and not self.use_tf_attention and self.kern is None
        if self._sync is not None:
            self.mlp.set_sync(self._sync)
        # the whole step as one chain of hand-written kernels, replayed as one hipGraph (nets/din_fused.py)
        self._fstep, self.graph_step = None, bool(graph_step)
        if fused_step and self.device.type == "cuda" and self.kern is None:
            from .din_fused import FusedDINStep

            if FusedDINStep.supported(self):
                self._fstep = FusedDINStep(self)

    def _attend(self, q, keys, lens, W1, b1, W2, b2):
        if self.use_tf_attention:
            return dot_attention_torch(q, keys, lens)
        if self.kern is not None:        # row-sharded tables: the provider's dense-form attention
            return self.kern.din_attention(q.contiguous(), keys.contiguous(), lens, W1, b1, W2, b2)
        att = din_attention_dense(q, keys, lens, W1, b1, W2, b2)     # item side features: materialised rows, same kernels
        return att if att is not None else din_attention_torch(q, keys, lens, W1, b1, W2, b2)

    def _att_params(self):
        P = self.P
        if self.use_tf_attention:
            return None, None, None, None
        return (P["attention/attention_layer1/kernel"], P["attention/attention_layer1/bias"],
                P["attention/attention_layer2/kernel"], P["attention/attention_layer2/bias"])

    def _item_feats(self, ids: torch.Tensor):
        """combine_seq_features(..., "concat") for the given item ids only (tfops/features.py:151-218)."""
        t = self.tables
        flat = ids.reshape(-1).long()
        parts = [t.embed[t.item_off + flat]]
        if self.item_sparse is not None:
            parts.append(t.embed[t.sparse_off + self.item_sparse[flat].long()].flatten(1))
        if self.item_dense is not None:
            w = self.P["embedding/dense_embeds_var"][self.item_dense_cols]
            parts.append((self.item_dense[flat][:, :, None] * w[None]).flatten(1))
        return torch.cat(parts, dim=1).view(*ids.shape, self.Kp)

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32).contiguous()

    # ---- row-sharded tables: the target's and the window's rows through the step's exchange ----
    def _window_rows(self, it, sq):
        """GLOBAL rows of [item, window] (1 + L per sample) and of their sparse side features ((1 + L) * n_is), [B, n_extra]."""
        t = self.tables
        ids_all = torch.cat([it.view(-1, 1), sq], dim=1)                              # [B, 1 + L]
        blocks = [ids_all + t.item_off]
        if self.item_sparse is not None:
            blocks.append((t.sparse_off + self.item_sparse[ids_all.reshape(-1).long()]).view(len(it), -1))
        return torch.cat(blocks, dim=1).to(torch.int32).contiguous()

    def _window_feats(self, ctx, it, sq, grad=True):
        """-> (feats [B, 1 + L, K'], [(cache slots, leaf rows)]) — `_item_feats` read off the step's row cache."""
        B, n1, K = len(it), 1 + self.L, self.K
        cache, sl = ctx.sctx.cache, ctx.slot_extra
        s_item = sl[:, :n1].contiguous()
        rows_item = self.kern.gather(cache, s_item).view(B * n1, K).requires_grad_(grad)
        parts, streams = [rows_item], [(s_item.reshape(-1), rows_item)]
        if self.item_sparse is not None:
            s_sp = sl[:, n1:].contiguous()
            rows_sp = self.kern.gather(cache, s_sp).view(-1, K).requires_grad_(grad)
            parts.append(rows_sp.view(B * n1, -1))
            streams.append((s_sp.reshape(-1), rows_sp))
        if self.item_dense is not None:
            flat = torch.cat([it.view(-1, 1), sq], dim=1).reshape(-1).long()
            wdn = self.P["embedding/dense_embeds_var"][self.item_dense_cols]
            parts.append((self.item_dense[flat][:, :, None] * wdn[None]).flatten(1))
        return torch.cat(parts, dim=1).view(B, n1, self.Kp), streams

    def _logits(self, E, att, training):
        return self.out(self.mlp(torch.cat([E.flatten(1), att], dim=1), training)).squeeze(1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, seqs=None, seq_lens=None, **_):
        it, sq, ln = self._i32(items), self._i32(seqs), self._i32(seq_lens)
        W1, b1, W2, b2 = self._att_params()
        if self.kern is not None:            # row-sharded tables: the window's rows ride in the lookup (a collective)
            ctx, E, _ = self.emb.forward(users, items, sparse, dense, grad=False, extra_idx=self._window_rows(it, sq))
            feats, _ = self._window_feats(ctx, it, sq, grad=False)
            return self._logits(E, self._attend(feats[:, 0], feats[:, 1:], ln, W1, b1, W2, b2), False)
        _, E, _ = self.emb.forward(users, items, sparse, dense, grad=False)
        if self.fused:
            att, _ = ops.din_attn_pool_fwd(self.tables.variable("item_embeds_var"), it, sq, ln,
                                           W1.detach(), b1.detach(), W2.detach(), b2.detach())
        else:
            att = self._attend(self._item_feats(it), self._item_feats(sq), ln, W1, b1, W2, b2)
        return self._logits(E, att, False)

    def train_step(self, users, items, labels, sparse=None, dense=None, seqs=None, seq_lens=None,
                   loss_type="cross_entropy", **_):
        self.step += 1
        if self._fstep is not None and loss_type == "cross_entropy":
            sp = self._i32(sparse) if self.spec.n_sparse_cols else None
            return self._fstep.train_step(self._i32(users), self._i32(items), sp, self._i32(seqs), self._i32(seq_lens),
                                          self._labels(labels).contiguous(), self.graph_step)
        it, sq, ln = self._i32(items), self._i32(seqs), self._i32(seq_lens)
        W1, b1, W2, b2 = self._att_params()
        if self.kern is not None:
            ctx, E, _ = self.emb.forward(users, items, sparse, dense, extra_idx=self._window_rows(it, sq))
            self.P.zero_grad()
            feats, streams = self._window_feats(ctx, it, sq)
            att = self._attend(feats[:, 0], feats[:, 1:], ln, W1, b1, W2, b2)
            loss = _FieldNet.loss_fn(self._logits(E, att, True), self._labels(labels), loss_type)
            with torch.no_grad():
                extra = lambda: (torch.cat([s_[0] for s_ in streams]), torch.cat([s_[1].grad.view(-1, self.K) for s_ in streams]))
            return self._finish(ctx, loss, extra)
        ctx, E, _ = self.emb.forward(users, items, sparse, dense)
        self.P.zero_grad()
        t = self.tables
        if self.fused:
            item_tab = t.variable("item_embeds_var")
            w = [x.detach() for x in (W1, b1, W2, b2)]
            att, attn = ops.din_attn_pool_fwd(item_tab, it, sq, ln, *w)
            att.requires_grad_(True)
            loss = _FieldNet.loss_fn(self._logits(E, att, True), self._labels(labels), loss_type)
            loss.backward()
            with torch.no_grad():
                # query / key gradients are written straight behind the plain streams' rows of ONE gradient buffer
                # (no 2 x 210 MB concatenations at cfg 3)
                B_, n_head = len(it), self.emb.n_plain_positions(ctx)
                gbuf = torch.empty((n_head + B_ * (1 + self.L), self.K), dtype=torch.float32, device=self.device)
                gq, gkey, gW1, gb1, gW2, gb2 = ops.din_attn_pool_bwd(
                    item_tab, it, sq, ln, *w, attn, att.grad.contiguous(), gq_out=gbuf[n_head:n_head + B_],
                    gkey_out=gbuf[n_head + B_:].view(B_, self.L, self.K))
                for p, g in zip((W1, b1, W2, b2), (gW1, gb1, gW2, gb2)):
                    p.grad.add_(g.view_as(p))
                valid = torch.arange(self.L, device=self.device)[None, :] < ln[:, None]
                seq_rows = torch.where(valid, sq + t.item_off, torch.full_like(sq, -1))   # pads dropped
                extra = (torch.cat([it + t.item_off, seq_rows.reshape(-1)]), gbuf, n_head)
                hp = self._hp()
                self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, extra)
                self.P.adam_step(hp)
            return loss.detach()
        # general path (item side features): keys are assembled with torch gathers; their table
        # gradients come back as one more (index, gradient) stream
        ids_all = torch.cat([it.view(-1, 1), sq], dim=1)                              # [B, 1+L]
        flat = ids_all.reshape(-1).long()
        rows_item = t.embed[t.item_off + flat].requires_grad_(True)
        parts, streams = [rows_item], [(t.item_off + flat.to(torch.int32), rows_item)]
        if self.item_sparse is not None:
            sidx = (t.sparse_off + self.item_sparse[flat]).reshape(-1)
            rows_sp = t.embed[sidx.long()].requires_grad_(True)
            parts.append(rows_sp.view(len(flat), -1))
            streams.append((sidx.to(torch.int32), rows_sp))
        if self.item_dense is not None:
            wdn = self.P["embedding/dense_embeds_var"][self.item_dense_cols]
            parts.append((self.item_dense[flat][:, :, None] * wdn[None]).flatten(1))
        feats = torch.cat(parts, dim=1).view(len(it), 1 + self.L, self.Kp)
        att = self._attend(feats[:, 0], feats[:, 1:], ln, W1, b1, W2, b2)
        loss = _FieldNet.loss_fn(self._logits(E, att, True), self._labels(labels), loss_type)
        loss.backward()
        with torch.no_grad():
            extra = (torch.cat([s[0] for s in streams]), torch.cat([s[1].grad.view(-1, self.K) for s in streams]))
            hp = self._hp()
            self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, extra)
            self.P.adam_step(hp)
        return loss.detach()


class FeatYouTubeRankingNet(_FeatNet):
    """algorithms/youtube_ranking.py:163-246: [user, item, pooled recent items, sparse, dense] embeddings -> MLP ->
    Dense(1).  The pooled field is `seq_embeds_pooling` (layers/embedding.py:54-85): rows of the ITEM table at the
    user's recent items, pad id -> 0-vector, summed and divided by sqrt(seq_len) — `lr_embed_bag_pool_f32` (sum
    combiner, OOV = pad row) and its backward, whose per-position gradients join the plain streams of the table
    update.  No linear tables."""
    with_linear = False

    def __init__(self, spec, embed_size=16, hidden_units=(128, 64, 32), use_bn=True, dropout_rate=0.0,
                 max_seq_len=10, lr=1e-3, epsilon=1e-5, seed=42, device=None, dense_adam=False, reg=None):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg)
        self.L = max_seq_len
        self.mlp = DenseStack(self.P, "mlp", (spec.n_fields + 1) * embed_size, hidden_units, use_bn, dropout_rate)
        self.out = TFDense(self.P, "out", self.mlp.n_out, 1)
        self.P.finalize()

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32).contiguous()

    def _pool(self, seqs, seq_lens):
        t = self.tables
        rows = (self._i32(seqs) + t.item_off).contiguous()                  # [B, L] global rows; pad = item OOV row
        pad = t.item_off + self.spec.n_items
        summed = ops.embed_bag_pool(t.embed, rows, "sum", pad)
        return rows, pad, summed, torch.rsqrt(self._i32(seq_lens).to(torch.float32))[:, None]

    def _logits(self, E, pooled, training):
        x = torch.cat([E[:, :2].flatten(1), pooled, E[:, 2:].flatten(1)], dim=1)   # youtube_ranking.py:196-203 order
        return self.out(self.mlp(x, training)).squeeze(1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, seqs=None, seq_lens=None, **_):
        _, E, _ = self.emb.forward(users, items, sparse, dense, grad=False)
        _, _, summed, scale = self._pool(seqs, seq_lens)
        return self._logits(E, summed * scale, False)

    def train_step(self, users, items, labels, sparse=None, dense=None, seqs=None, seq_lens=None,
                   loss_type="cross_entropy", **_):
        self.step += 1
        ctx, E, _ = self.emb.forward(users, items, sparse, dense)
        rows, pad, summed, scale = self._pool(seqs, seq_lens)
        summed.requires_grad_(True)
        self.P.zero_grad()
        loss = _FieldNet.loss_fn(self._logits(E, summed * scale, True), self._labels(labels), loss_type)
        loss.backward()
        with torch.no_grad():
            gpos = ops.embed_bag_pool_bwd(summed.grad.contiguous(), rows, self.tables.V, "sum", pad)   # 0 at pads
            ids = torch.where(rows == pad, torch.full_like(rows, -1), rows).reshape(-1)             # pads dropped
            hp = self._hp()
            self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, (ids, gpos))
            self.P.adam_step(hp)
        return loss.detach()


class ShardedDINNet:
    """DIN (algorithms/din.py:165-250, pure-id items) with the user / item table ROW-SHARDED over the ranks.

    One step, one process per GPU, data-parallel batch: the global rows of [user, item, the L window items] of every
    sample -> `ShardedFieldTables.lookup` (ids all-to-all, owners gather, de-duplicated rows back) -> attention over
    the fetched rows (the MFMA attention kernels in their dense form) -> MLP (replicated dense parameters) -> loss / W
    -> row gradients summed per distinct row -> all-to-all to the owners -> owners sum across peers + row-wise Adam;
    dense gradients: one all-reduce.  Pad positions address the item OOV row (sequence.py:56-58); the attention mask
    gives them zero weight and zero gradient.  BatchNorm statistics are those of the GLOBAL batch (`TFBatchNorm.sync`)."""

    def __init__(self, n_rows_global, embed_size=16, hidden_units=(128, 64, 32), use_bn=True, max_seq_len=10, lr=1e-3,
                 epsilon=1e-5, seed=42, device=None, kern=None, group=None):
        import torch.distributed as dist

        from ..parallel import HipKernels, ShardedFieldTables

        self.kern, self.group = kern or HipKernels(), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device or torch.device("cuda")
        self.K, self.L = embed_size, max_seq_len
        self.tables = ShardedFieldTables(n_rows_global, embed_size, self.device, self.kern, with_linear=False,
                                         group=group, seed=seed)
        P = self.P = DenseParams(self.device, seed)
        P.add("attention/attention_layer1/kernel", (4 * embed_size, 16), "glorot_uniform")
        P.add("attention/attention_layer1/bias", (16,), "zeros")
        P.add("attention/attention_layer2/kernel", (16, 1), "glorot_uniform")
        P.add("attention/attention_layer2/bias", (1,), "zeros")
        self.mlp = DenseStack(P, "mlp", 3 * embed_size, hidden_units, use_bn, 0.0)
        self.out = TFDense(P, "out", self.mlp.n_out, 1)
        P.finalize()
        self.lr, self.epsilon, self.step = lr, epsilon, 0
        from ..parallel import rank_average

        self.mlp.set_sync(rank_average(group))      # BatchNorm over the GLOBAL batch

    def _logits(self, rows, lens, training):
        P = self.P
        att = self.kern.din_attention(rows[:, 1].contiguous(), rows[:, 2:].contiguous(), lens,
                                      P["attention/attention_layer1/kernel"], P["attention/attention_layer1/bias"],
                                      P["attention/attention_layer2/kernel"], P["attention/attention_layer2/bias"])
        return self.out(self.mlp(torch.cat([rows[:, 0], rows[:, 1], att], dim=1), training)).squeeze(1)

    def _rows(self, idx):
        ctx = self.tables.lookup(idx.to(torch.int32).contiguous())
        B, nf = ctx.slots.shape
        return ctx, self.kern.gather(ctx.cache, ctx.slots.reshape(-1).contiguous()).view(B, nf, self.K)

    def train_step(self, idx, seq_lens, labels, next_idx=None):
        """`idx` [B, 2 + L]: GLOBAL table rows [user, item, window...] of this rank's samples; `seq_lens` [B]."""
        from ..parallel import allreduce_sum_

        self.step += 1
        W, dev = self.world, self.device
        ctx, rows = self._rows(idx)
        rows.requires_grad_(True)
        self.P.zero_grad()
        lens = torch.as_tensor(seq_lens, device=dev).to(torch.int32)
        lab = torch.as_tensor(labels, device=dev, dtype=torch.float32)
        loss = F.binary_cross_entropy_with_logits(self._logits(rows, lens, True), lab)
        (loss / W).backward()                                   # global-batch mean
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

    def _idx(self, users, items, seqs) -> torch.Tensor:
        """[B, 2 + L] GLOBAL rows [user, item, window items] (pad id n_items = the item OOV row) — needs
        `tables.set_layout(n_users, n_items)`."""
        t, dev = self.tables, self.device
        u = torch.as_tensor(np.asarray(users) if not isinstance(users, torch.Tensor) else users, device=dev).to(torch.int32)
        i = torch.as_tensor(np.asarray(items) if not isinstance(items, torch.Tensor) else items, device=dev).to(torch.int32)
        q = torch.as_tensor(np.asarray(seqs) if not isinstance(seqs, torch.Tensor) else seqs, device=dev).to(torch.int32)
        return torch.cat([u.view(-1, 1) + t.user_off, i.view(-1, 1) + t.item_off, q + t.item_off], dim=1).contiguous()

    def assign_oov(self, sparse_oov_rows=None):
        self.tables.assign_oov(None)

    @torch.no_grad()
    def forward(self, a, b=None, sparse=None, dense=None, seqs=None, seq_lens=None):
        """`forward(idx, seq_lens)` with GLOBAL rows, or the feature models' `forward(users, items, seqs=, seq_lens=)`.
        A collective: every rank calls it with its own rows."""
        if seqs is not None:
            idx = self._idx(a, b, seqs)
        else:
            idx, seq_lens = a, (b if b is not None else seq_lens)
        _, rows = self._rows(idx)
        return self._logits(rows, torch.as_tensor(seq_lens, device=self.device).to(torch.int32), False)
