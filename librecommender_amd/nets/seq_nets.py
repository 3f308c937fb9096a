"""Transformer sequence model (`libreco/algorithms/transformer.py:204-339`) over the general feature embedding layer.

[user, item, sparse, dense] embeddings + a target-attention read-out of the user's behaviour sequence after
`num_tfm_layers` pre-norm transformer layers (RMSNorm -> multi-head self-attention -> residual -> RMSNorm -> GELU FFN
-> residual, layers/transformer.py + layers/attention.py:67-126 in the `tf.keras.layers.MultiHeadAttention` form the
reference takes for TF >= 2.10: bias-free q / k / v / output projections, scores scaled by 1/sqrt(head_dim), masked
positions pushed down by 1e9) -> swish MLP -> Dense(1).

The embedding rows come from `lr_embed_gather_f32` / `lr_embed_bag_pool_f32` (through `FeatEmbedding`) and go back as
one (index, gradient) stream into `lr_segments_build` + `lr_embed_scatter_adam_f32`; the [B, L, D] attention blocks
(L <= ~50) are torch ops on the device — the model is adjacent to the hot path, not on it.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

from ..layers import DenseStack, TFDense
from ..utils.device import to_device
from .feat_nets import _FeatNet, dot_attention_torch
from .fm_nets import _FieldNet


def rms_norm(x, scale):
    """layers/normalization.py:25-33."""
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-8) * scale


def layer_norm(x, scale, bias):
    """layers/normalization.py:9-22 (biased variance, eps 1e-8)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + 1e-8) * scale + bias


def gelu(x):
    """layers/activation.py:4-5 (erf form)."""
    return 0.5 * x * (1.0 + torch.erf(x / 1.4142135623730951))


def sinusoidal_encoding(seq_len, d_model):
    """layers/transformer.py:131-161: column 2i and 2i+1 share the frequency 10000^(2i/d)."""
    pos = np.arange(seq_len)[:, None]
    dim = np.arange(d_model) / d_model
    dim[1::2] = dim[0::2] if d_model % 2 == 0 else dim[0::2][:-1]
    pe = pos / (10000 ** dim[None, :])
    pe[:, 0::2] = np.sin(pe[:, 0::2])
    pe[:, 1::2] = np.cos(pe[:, 1::2])
    return pe.astype(np.float32)


def multi_head_attention(q_in, kv_in, Wq, Wk, Wv, Wo, num_heads, mask):
    """q_in [B,Tq,D], kv_in [B,Tk,D], W* [D, H*hd], Wo [H*hd, Dout], mask [B,Tq,Tk] bool (True = attend)."""
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    hd = Wq.shape[1] // num_heads
    q = (q_in @ Wq).view(B, Tq, num_heads, hd).transpose(1, 2)
    k = (kv_in @ Wk).view(B, Tk, num_heads, hd).transpose(1, 2)
    v = (kv_in @ Wv).view(B, Tk, num_heads, hd).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(hd))) @ k.transpose(2, 3)
    if mask is not None:
        s = s - 1e9 * (~mask[:, None, :, :]).to(s.dtype)                    # keras `_masked_softmax`
    out = torch.softmax(s, dim=-1) @ v
    return out.transpose(1, 2).reshape(B, Tq, num_heads * hd) @ Wo


class ItemSeqFeatures:
    """`combine_seq_features` (tfops/features.py:151-236) evaluated only at the ids a batch needs: the item row,
    the rows of the item's sparse columns and its dense columns, "concat"-ed or combined "elementwise"
    (item * (sum_f LN(sparse_f) + sum_f LN(dense_f) + 1))."""

    def __init__(self, net, item_sparse_unique, item_dense_unique, item_dense_cols, mode):
        dev, P, K = net.device, net.P, net.K
        self.net, self.mode = net, mode
        self.item_sparse = None if item_sparse_unique is None else torch.as_tensor(item_sparse_unique, device=dev).to(torch.int32)
        self.item_dense = None if item_dense_unique is None else torch.as_tensor(item_dense_unique, device=dev, dtype=torch.float32)
        self.item_dense_cols = list(item_dense_cols)
        n_is = 0 if self.item_sparse is None else self.item_sparse.shape[1]
        n_id = 0 if self.item_dense is None else self.item_dense.shape[1]
        if mode == "concat":
            self.dim = K * (1 + n_is + n_id)
        else:
            self.dim = K
            if n_is:
                P.add("elementwise_sparse_feats/layer_norm/scale", (K,), "ones")
                P.add("elementwise_sparse_feats/layer_norm/bias", (K,), "zeros")
            if n_id:
                P.add("elementwise_dense_feats/layer_norm/scale", (K,), "ones")
                P.add("elementwise_dense_feats/layer_norm/bias", (K,), "zeros")

    def __call__(self, ids: torch.Tensor, grad: bool):
        """ids [..] int -> (feats [.., dim], streams [(global rows int32 [n], leaf [n, K])])."""
        net, P, K = self.net, self.net.P, self.net.K
        t = net.tables
        flat = ids.reshape(-1).long()
        rows_item = t.embed[t.item_off + flat].requires_grad_(grad)
        streams = [((t.item_off + flat).to(torch.int32), rows_item)]
        sp = dn = None
        if self.item_sparse is not None:
            sidx = (t.sparse_off + self.item_sparse[flat]).reshape(-1)
            rows_sp = t.embed[sidx.long()].requires_grad_(grad)
            streams.append((sidx.to(torch.int32), rows_sp))
            sp = rows_sp.view(len(flat), -1, K)
        if self.item_dense is not None:
            w = P["embedding/dense_embeds_var"][self.item_dense_cols]
            dn = self.item_dense[flat][:, :, None] * w[None]
        if self.mode == "concat":
            parts = [rows_item] + ([sp.flatten(1)] if sp is not None else []) + ([dn.flatten(1)] if dn is not None else [])
            out = torch.cat(parts, dim=1) if len(parts) > 1 else rows_item
        else:
            extra = 1.0
            if sp is not None:
                extra = extra + layer_norm(sp, P["elementwise_sparse_feats/layer_norm/scale"],
                                           P["elementwise_sparse_feats/layer_norm/bias"]).sum(1)
            if dn is not None:
                extra = extra + layer_norm(dn, P["elementwise_dense_feats/layer_norm/scale"],
                                           P["elementwise_dense_feats/layer_norm/bias"]).sum(1)
            out = rows_item * extra
        return out.view(*ids.shape, self.dim), streams


class FeatTransformerNet(_FeatNet):
    with_linear = False

    def __init__(self, spec, embed_size=16, hidden_units: Sequence[int] = (128, 64, 32), use_bn=True, dropout_rate=0.0,
                 max_seq_len=10, num_heads=1, num_tfm_layers=1, positional_embedding="trainable",
                 use_causal_mask=False, feat_agg_mode="concat", item_sparse_unique=None, item_dense_unique=None,
                 item_dense_cols: Sequence[int] = (), lr=1e-3, epsilon=1e-5, seed=42, device=None,
                 dense_adam=False, reg=None):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg)
        P, K = self.P, embed_size
        self.L, self.H, self.n_layers, self.causal = max_seq_len, num_heads, num_tfm_layers, use_causal_mask
        self.seq_feats = ItemSeqFeatures(self, item_sparse_unique, item_dense_unique, item_dense_cols, feat_agg_mode)
        D = self.D = self.seq_feats.dim + K                                  # item feature dim + position dim
        if D % num_heads != 0:
            raise AssertionError(f"`item_dim`({D}) should be divisible by `num_heads`({num_heads})")
        self.trainable_pos = positional_embedding not in ("sinusoidal", "sin", "sinusoid")
        if self.trainable_pos:
            P.add("transformer/positional_encoding", (max_seq_len, K), "glorot_uniform")
        else:
            self.pos_const = torch.from_numpy(sinusoidal_encoding(max_seq_len, K)).to(self.device)
        for l in range(1, num_tfm_layers + 1):
            s = f"transformer_layer{l}"
            P.add(f"{s}/rms_norm_att/scale", (D,), "ones")
            for w in ("query", "key", "value"):
                P.add(f"{s}/multi_head_attention/{w}/kernel", (D, D), "glorot_uniform")
            P.add(f"{s}/multi_head_attention/attention_output/kernel", (D, D), "glorot_uniform")
            P.add(f"{s}/rms_norm_ffn/scale", (D,), "ones")
            P.add(f"{s}/ffn/dense/kernel", (D, 4 * D), "glorot_uniform")
            P.add(f"{s}/ffn/dense_1/kernel", (4 * D, D), "glorot_uniform")
        P.add("rms_norm_last/scale", (D,), "ones")
        P.add("rms_norm_item/scale", (self.seq_feats.dim,), "ones")
        self.mlp = DenseStack(P, "mlp", spec.n_fields * K + D, hidden_units, use_bn, dropout_rate, activation=F.silu)
        self.out = TFDense(P, "out", self.mlp.n_out, 1)
        P.finalize()

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32).contiguous()

    def _seq_repr(self, feats, lens):
        """`_build_seq_repr` (transformer.py:281-309): feats [B, 1+L, dim] = target item first, then the window."""
        P, B, L = self.P, feats.shape[0], self.L
        item, seq = feats[:, 0], feats[:, 1:]
        pos = P["transformer/positional_encoding"] if self.trainable_pos else self.pos_const
        x = torch.cat([seq, pos[None].expand(B, -1, -1)], dim=2)
        key_ok = torch.arange(L, device=self.device)[None, :] < lens[:, None]          # compute_seq_mask
        mask = key_ok[:, None, :].expand(-1, L, -1)
        if self.causal:   # transformer.py:323-328: logical OR of the sequence mask and the causal mask
            mask = mask | torch.ones(L, L, dtype=torch.bool, device=self.device).tril()[None]
        for l in range(1, self.n_layers + 1):
            s = f"transformer_layer{l}"
            h = rms_norm(x, P[f"{s}/rms_norm_att/scale"])
            att = multi_head_attention(h, h, P[f"{s}/multi_head_attention/query/kernel"],
                                       P[f"{s}/multi_head_attention/key/kernel"],
                                       P[f"{s}/multi_head_attention/value/kernel"],
                                       P[f"{s}/multi_head_attention/attention_output/kernel"], self.H, mask) + x
            h = rms_norm(att, P[f"{s}/rms_norm_ffn/scale"])
            x = att + gelu(h @ P[f"{s}/ffn/dense/kernel"]) @ P[f"{s}/ffn/dense_1/kernel"]
        x = rms_norm(x, P["rms_norm_last/scale"])
        q = torch.cat([rms_norm(item, P["rms_norm_item/scale"]), torch.ones((B, self.K), device=self.device)], dim=1)
        return dot_attention_torch(q, x, lens)                                          # layers/attention.py:5-25

    def _logits(self, E, seq_out, training):
        return self.out(self.mlp(torch.cat([E.flatten(1), seq_out], dim=1), training)).squeeze(1)

    def _ids_all(self, items, seqs):
        return torch.cat([self._i32(items).view(-1, 1), self._i32(seqs)], dim=1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, seqs=None, seq_lens=None, **_):
        _, E, _ = self.emb.forward(users, items, sparse, dense, grad=False)
        feats, _ = self.seq_feats(self._ids_all(items, seqs), False)
        return self._logits(E, self._seq_repr(feats, self._i32(seq_lens)), False)

    def train_step(self, users, items, labels, sparse=None, dense=None, seqs=None, seq_lens=None,
                   loss_type="cross_entropy", **_):
        self.step += 1
        ctx, E, _ = self.emb.forward(users, items, sparse, dense)
        feats, streams = self.seq_feats(self._ids_all(items, seqs), True)
        self.P.zero_grad()
        logits = self._logits(E, self._seq_repr(feats, self._i32(seq_lens)), True)
        loss = _FieldNet.loss_fn(logits, self._labels(labels), loss_type)
        loss.backward()
        with torch.no_grad():
            extra = (torch.cat([s[0] for s in streams]), torch.cat([s[1].grad.view(-1, self.K) for s in streams]))
            hp = self._hp()
            self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, extra)
            self.P.adam_step(hp)
        return loss.detach()


class FeatSIMNet(_FeatNet):
    """Search-based interest model (`libreco/algorithms/sim.py:191-345`).

    Item features of the target, the LONG window and the SHORT window are the "concat" item feature rows projected to
    `embed_size` by one bias-free Dense (sim.py:195-197).  First stage: masked sum of the long window + target -> MLP.
    Second stage: general search unit = the `search_topk` long-window items with the largest inner product with the
    target, exact search unit = multi-head target attention over them, short window = dot-product attention
    (`tf_attention`), both concatenated with the [user, item, sparse, dense] embeddings -> MLP.  Training minimises the
    loss of alpha * first + beta * second; inference scores with the second stage alone (sim.py:205-207).

    `seqs` is [B, long_max_len + short_max_len] (long window first), `seq_lens` is [B, 2]."""
    with_linear = False

    def __init__(self, spec, embed_size=16, hidden_units: Sequence[int] = (200, 80), use_bn=True, dropout_rate=0.0,
                 alpha=1.0, beta=1.0, search_topk=10, long_max_len=100, short_max_len=10, num_heads=2,
                 item_sparse_unique=None, item_dense_unique=None, item_dense_cols: Sequence[int] = (), lr=1e-3,
                 epsilon=1e-5, seed=42, device=None, dense_adam=False, reg=None):
        super().__init__(spec, embed_size, lr, epsilon, seed, device, dense_adam, reg)
        P, K = self.P, embed_size
        if K % num_heads != 0:
            raise AssertionError(f"`item_dim`({K}) should be divisible by `num_heads`({num_heads})")
        self.alpha, self.beta, self.topk, self.Lg, self.S, self.H = alpha, beta, search_topk, long_max_len, short_max_len, num_heads
        self.seq_feats = ItemSeqFeatures(self, item_sparse_unique, item_dense_unique, item_dense_cols, "concat")
        P.add("seq_feats_proj/kernel", (self.seq_feats.dim, K), "glorot_uniform")
        for w in ("query", "key", "value", "attention_output"):
            P.add(f"multi_head_attention/{w}/kernel", (K, K), "glorot_uniform")
        self.first_stage_mlp = DenseStack(P, "first_stage_mlp", 2 * K, hidden_units, use_bn, dropout_rate)
        self.first_out = TFDense(P, "first_stage_out", self.first_stage_mlp.n_out, 1)
        self.second_stage_mlp = DenseStack(P, "second_stage_mlp", (2 + spec.n_fields) * K, hidden_units, use_bn, dropout_rate)
        self.second_out = TFDense(P, "second_stage_out", self.second_stage_mlp.n_out, 1)
        P.finalize()

    def _i32(self, x):
        return to_device(x, self.device).to(torch.int32).contiguous()

    def _stages(self, E, feats, lens, training, first: bool):
        P, Lg = self.P, self.Lg
        proj = feats @ P["seq_feats_proj/kernel"]                                       # [B, 1 + Lg + S, K]
        target, long, short = proj[:, 0], proj[:, 1:1 + Lg], proj[:, 1 + Lg:]
        long_ok = torch.arange(Lg, device=self.device)[None, :] < lens[:, :1]
        # general search unit (sim.py:254-276): top-k of the masked inner products; unsorted in TF, irrelevant below
        scores = torch.einsum("bk,blk->bl", target, long).masked_fill(~long_ok, -1e9)
        idx = torch.topk(scores.detach(), self.topk, dim=1, sorted=False).indices
        top = torch.gather(long, 1, idx[:, :, None].expand(-1, -1, long.shape[2]))
        top_ok = torch.gather(long_ok, 1, idx)
        long_out = multi_head_attention(target[:, None, :], top, P["multi_head_attention/query/kernel"],
                                        P["multi_head_attention/key/kernel"], P["multi_head_attention/value/kernel"],
                                        P["multi_head_attention/attention_output/kernel"], self.H,
                                        top_ok[:, None, :]).squeeze(1)
        short_out = dot_attention_torch(target, short, lens[:, 1])
        x2 = torch.cat([long_out, short_out, E.flatten(1)], dim=1)
        second = self.second_out(self.second_stage_mlp(x2, training)).squeeze(1)
        if not first:
            return None, second
        pooled = (long * long_ok[:, :, None].to(long.dtype)).sum(1)                     # sim.py:227-245
        x1 = torch.cat([target, pooled], dim=1)
        return self.first_out(self.first_stage_mlp(x1, training)).squeeze(1), second

    def _ids_all(self, items, seqs):
        return torch.cat([self._i32(items).view(-1, 1), self._i32(seqs)], dim=1)

    @torch.no_grad()
    def forward(self, users, items, sparse=None, dense=None, seqs=None, seq_lens=None, **_):
        _, E, _ = self.emb.forward(users, items, sparse, dense, grad=False)
        feats, _ = self.seq_feats(self._ids_all(items, seqs), False)
        return self._stages(E, feats, self._i32(seq_lens).view(-1, 2), False, first=False)[1]

    def train_step(self, users, items, labels, sparse=None, dense=None, seqs=None, seq_lens=None,
                   loss_type="cross_entropy", **_):
        self.step += 1
        ctx, E, _ = self.emb.forward(users, items, sparse, dense)
        feats, streams = self.seq_feats(self._ids_all(items, seqs), True)
        self.P.zero_grad()
        first, second = self._stages(E, feats, self._i32(seq_lens).view(-1, 2), True, first=True)
        loss = _FieldNet.loss_fn(self.alpha * first + self.beta * second, self._labels(labels), loss_type)
        loss.backward()
        with torch.no_grad():
            extra = (torch.cat([s[0] for s in streams]), torch.cat([s[1].grad.view(-1, self.K) for s in streams]))
            hp = self._hp()
            self.emb.apply_gradients(ctx, hp, self.dense_adam, self.reg, extra)
            self.P.adam_step(hp)
        return loss.detach()
