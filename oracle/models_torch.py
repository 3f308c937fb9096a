"""PyTorch-CPU restatement of the reference's TensorFlow graphs — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: TensorFlow (`tensorflow>=1.15,<2.16`, requirements.txt:5) is an un-vendored
dependency that cannot be installed in the build container and no reference test pins these
numerics; each block follows the cited reference lines and TensorFlow's documented op
semantics (BN momentum 0.99 / eps 1e-3 / batch statistics; Dense = glorot-uniform kernel + zero
bias; tf.train.AdamOptimizer sparse-apply = dense decay of every row).

Also used (bounded sample) as the `cpu_baseline` of bench.py, kind "port".
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


class TF1Adam:
    """tf.train.AdamOptimizer(lr, epsilon) (training/tf_trainer.py:120): dense variables use
    _apply_dense; IndexedSlices gradients use _apply_sparse_shared, which still decays m, v and
    moves EVERY row.  Both reduce to the same arithmetic given a densified gradient; the sparse
    path is executed with index_add_ to keep the CPU cost realistic."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-5):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.state: Dict[int, tuple] = {}

    def step(self, params: Sequence[torch.Tensor]):
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        lr_t = self.lr * math.sqrt(bc2) / bc1
        with torch.no_grad():
            for p in params:
                if p.grad is None:
                    continue
                if id(p) not in self.state:
                    self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
                m, v = self.state[id(p)]
                dt = p.dtype
                b1 = torch.tensor(self.b1, dtype=dt)
                b2 = torch.tensor(self.b2, dtype=dt)
                omb1, omb2 = (1 - b1), (1 - b2)   # formed in the variable dtype, like TF
                g = p.grad
                if g.is_sparse:
                    g = g.coalesce()
                    rows, vals = g.indices()[0], g.values()
                    m.mul_(b1).index_add_(0, rows, vals * omb1)
                    v.mul_(b2).index_add_(0, rows, vals * vals * omb2)
                else:
                    m.mul_(b1).add_(g * omb1)
                    v.mul_(b2).add_(g * g * omb2)
                p.sub_(lr_t * (m / (v.sqrt() + self.eps)))
                p.grad = None


def tf_batch_norm(x, gamma, beta, moving_mean, moving_var, training, momentum=0.99, eps=1e-3):
    """tf.layers.batch_normalization (layers/dense.py:31-41)."""
    if training:
        var, mean = torch.var_mean(x, dim=0, unbiased=False)
        with torch.no_grad():
            moving_mean.mul_(momentum).add_(mean * (1 - momentum))
            moving_var.mul_(momentum).add_(var * (1 - momentum))
    else:
        mean, var = moving_mean, moving_var
    return (x - mean) * (gamma * torch.rsqrt(var + eps)) + beta


class _Vars:
    def __init__(self, dtype):
        self.dtype = dtype
        self.v: Dict[str, torch.Tensor] = {}
        self.buffers: Dict[str, torch.Tensor] = {}

    def add(self, name, tensor, trainable=True):
        t = tensor.detach().to(self.dtype).clone()
        if trainable:
            t.requires_grad_(True)
            self.v[name] = t
        else:
            self.buffers[name] = t
        return t

    def trainable(self):
        return list(self.v.values())


class DenseNN:
    """dense_nn (layers/dense.py:12-49) over explicitly given weights."""

    def __init__(self, V: _Vars, name, weights: Dict[str, torch.Tensor], n_layers, use_bn,
                 activation=F.relu):
        self.V, self.name, self.n_layers, self.use_bn, self.act = V, name, n_layers, use_bn, activation
        # `dropout_fn(layer, x)` (layer = 0 for the first hidden layer): tf.layers.dropout after a hidden layer's BatchNorm
        # (layers/dense.py:44-47).  TF's mask stream cannot be reproduced; a test hands in the mask the kernels draw.
        self.dropout_fn = None
        for k, w in weights.items():
            V.add(k, w, trainable=not ("moving" in k))

    def __call__(self, x, training):
        V, n = self.V, self.name
        g = lambda k: V.v[k] if k in V.v else V.buffers[k]  # noqa: E731
        if self.use_bn:
            x = tf_batch_norm(x, g(f"{n}/bn_in/gamma"), g(f"{n}/bn_in/beta"),
                              g(f"{n}/bn_in/moving_mean"), g(f"{n}/bn_in/moving_var"), training)
        for i in range(1, self.n_layers + 1):
            x = x @ g(f"{n}/{n}_layer{i}/kernel") + g(f"{n}/{n}_layer{i}/bias")
            if i != self.n_layers:
                x = self.act(x)
                if self.use_bn:
                    x = tf_batch_norm(x, g(f"{n}/bn{i}/gamma"), g(f"{n}/bn{i}/beta"),
                                      g(f"{n}/bn{i}/moving_mean"), g(f"{n}/bn{i}/moving_var"), training)
                if training and self.dropout_fn is not None:
                    x = self.dropout_fn(i - 1, x)
        return x


class DeepFMOracle:
    """algorithms/deepfm.py:143-264 for data without dense / multi-sparse columns.

    `weights` maps reference variable names (user_embeds_var, item_embeds_var, sparse_embeds_var,
    user_linear_var, item_linear_var, sparse_linear_var) and dense-layer names to tensors."""

    def __init__(self, weights: Dict[str, torch.Tensor], hidden_units=(128, 64, 32), use_bn=True,
                 lr=1e-3, epsilon=1e-5, dtype=torch.float32):
        self.V = _Vars(dtype)
        for k in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var", "user_linear_var",
                  "item_linear_var", "sparse_linear_var"):
            if k in weights:
                w = weights[k]
                # the 1-D sparse_linear_var [S] is held as [S,1] so F.embedding(sparse=True) applies
                self.V.add(k, w.reshape(-1, 1) if k.endswith("linear_var") else w)
        self.has_sparse = "sparse_embeds_var" in weights
        mlp_w = {k: w for k, w in weights.items() if k.startswith("mlp/")}
        self.mlp = DenseNN(self.V, "mlp", mlp_w, len(hidden_units), use_bn)
        for k in ("linear/kernel", "linear/bias", "out/kernel", "out/bias"):
            self.V.add(k, weights[k])
        self.opt = TF1Adam(lr, eps=epsilon)

    def _lookup(self, name, idx, sparse_grad):
        return F.embedding(idx, self.V.v[name], sparse=sparse_grad)

    def forward(self, users, items, sparse_indices, training=False, sparse_grad=False):
        V = self.V.v
        ue = self._lookup("user_embeds_var", users, sparse_grad)              # deepfm.py:193-206
        ie = self._lookup("item_embeds_var", items, sparse_grad)
        ul = self._lookup("user_linear_var", users, sparse_grad).reshape(-1, 1)
        il = self._lookup("item_linear_var", items, sparse_grad).reshape(-1, 1)
        lin, pw, deep = [ul, il], [ue[:, None, :], ie[:, None, :]], [ue, ie]
        if self.has_sparse:
            se = self._lookup("sparse_embeds_var", sparse_indices, sparse_grad)   # [B,Fs,K]
            sl = F.embedding(sparse_indices, V["sparse_linear_var"], sparse=sparse_grad).squeeze(-1)
            lin.append(sl)
            pw.append(se)
            deep.append(se.flatten(1))                                         # deepfm.py:236
        linear_embed = torch.cat(lin, dim=1)
        pairwise_embed = torch.cat(pw, dim=1)
        deep_embed = torch.cat(deep, dim=1)
        linear_term = linear_embed @ V["linear/kernel"] + V["linear/bias"]     # deepfm.py:158
        s = pairwise_embed.sum(dim=1)
        pairwise_term = 0.5 * (s * s - (pairwise_embed * pairwise_embed).sum(dim=1))  # :159-162
        deep_term = self.mlp(deep_embed, training)                             # :163-169
        concat = torch.cat([linear_term, pairwise_term, deep_term], dim=1)     # :171
        return (concat @ V["out/kernel"] + V["out/bias"]).squeeze(1)           # :172

    def loss(self, users, items, sparse_indices, labels, sparse_grad=False):
        logits = self.forward(users, items, sparse_indices, training=True, sparse_grad=sparse_grad)
        return F.binary_cross_entropy_with_logits(logits, labels.to(logits.dtype))  # tfops/loss.py:14-16

    def train_step(self, users, items, sparse_indices, labels):
        loss = self.loss(users, items, sparse_indices, labels, sparse_grad=True)
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class FeatDeepFMOracle(DeepFMOracle):
    """algorithms/deepfm.py:143-264 WITH multi-sparse (pooled) and dense columns: `multi_sparse_combine_embedding`
    / `multi_sparse_alone` (tfops/features.py:47-118: the field's rows summed with the OOV rows read as zero, divided by
    the number — or its square root — of non-OOV entries, `div_no_nan`) and `compute_dense_feats`
    (tfops/features.py:121-148: dense_embeds_var[f] * value).  Field order [user, item, plain sparse columns, pooled
    fields, dense columns] (deepfm.py:175-264).  `fields` = [(offset, length, oov_row)], `plain_cols` = the sparse
    columns outside every field.  [UNPINNED: TF]"""

    def __init__(self, weights, hidden_units=(128, 64, 32), use_bn=True, lr=1e-3, epsilon=1e-5, dtype=torch.float32,
                 plain_cols=(), fields=(), combiner="sqrtn"):
        super().__init__(weights, hidden_units, use_bn, lr, epsilon, dtype)
        self.plain_cols, self.fields, self.combiner = list(plain_cols), list(fields), combiner
        self.has_dense = "embedding/dense_embeds_var" in weights
        if self.has_dense:
            self.V.add("embedding/dense_embeds_var", weights["embedding/dense_embeds_var"])
            self.V.add("embedding/dense_linear_var", weights["embedding/dense_linear_var"])

    def _pooled(self, name, idx, oov, sparse_grad):
        e = F.embedding(idx, self.V.v[name], sparse=sparse_grad)                 # [B, n, K or 1]
        keep = (idx != oov).to(e.dtype)
        tot = (e * keep[:, :, None]).sum(dim=1)                                  # OOV rows are zero vectors
        if self.combiner in ("mean", "sqrtn"):
            n = keep.sum(dim=1, keepdim=True)
            if self.combiner == "sqrtn":
                n = n.sqrt()
            tot = torch.where(n > 0, tot / n.clamp_min(1e-30), torch.zeros_like(tot))     # tf.div_no_nan
        return tot

    def forward(self, users, items, sparse_indices, dense_values=None, training=False, sparse_grad=False):
        V = self.V.v
        ue = self._lookup("user_embeds_var", users, sparse_grad)
        ie = self._lookup("item_embeds_var", items, sparse_grad)
        lin = [self._lookup("user_linear_var", users, sparse_grad).reshape(-1, 1),
               self._lookup("item_linear_var", items, sparse_grad).reshape(-1, 1)]
        pw = [ue[:, None, :], ie[:, None, :]]
        if self.plain_cols:
            pi = sparse_indices[:, self.plain_cols]
            pw.append(self._lookup("sparse_embeds_var", pi, sparse_grad))
            lin.append(F.embedding(pi, V["sparse_linear_var"], sparse=sparse_grad).squeeze(-1))
        for off, n, oov in self.fields:
            fi = sparse_indices[:, off:off + n]
            pw.append(self._pooled("sparse_embeds_var", fi, oov, sparse_grad)[:, None, :])
            lin.append(self._pooled("sparse_linear_var", fi, oov, sparse_grad))
        if self.has_dense:
            dv = dense_values.to(ue.dtype)
            pw.append(dv[:, :, None] * V["embedding/dense_embeds_var"][None])
            lin.append(dv * V["embedding/dense_linear_var"][None])
        linear_embed = torch.cat(lin, dim=1)
        pairwise_embed = torch.cat(pw, dim=1)
        linear_term = linear_embed @ V["linear/kernel"] + V["linear/bias"]
        s = pairwise_embed.sum(dim=1)
        pairwise_term = 0.5 * (s * s - (pairwise_embed * pairwise_embed).sum(dim=1))
        deep_term = self.mlp(pairwise_embed.flatten(1), training)
        concat = torch.cat([linear_term, pairwise_term, deep_term], dim=1)
        return (concat @ V["out/kernel"] + V["out/bias"]).squeeze(1)

    def loss(self, users, items, sparse_indices, dense_values, labels, sparse_grad=False):
        logits = self.forward(users, items, sparse_indices, dense_values, training=True, sparse_grad=sparse_grad)
        return F.binary_cross_entropy_with_logits(logits, labels.to(logits.dtype))

    def train_step(self, users, items, sparse_indices, dense_values, labels):
        loss = self.loss(users, items, sparse_indices, dense_values, labels, sparse_grad=True)
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class FMOracle:
    """algorithms/fm.py:140-255: linear + Dense(1, elu)(BN(pairwise))."""

    def __init__(self, weights, use_bn=True, lr=1e-3, epsilon=1e-5, dtype=torch.float32):
        self.V = _Vars(dtype)
        self.use_bn = use_bn
        for k, w in weights.items():
            self.V.add(k, w.reshape(-1, 1) if k.endswith("linear_var") else w, trainable="moving" not in k)
        self.has_sparse = "sparse_embeds_var" in weights
        self.opt = TF1Adam(lr, eps=epsilon)

    def forward(self, users, items, sparse_indices, training=False, sparse_grad=False):
        V, Bf = self.V.v, self.V.buffers
        emb = lambda n, i: F.embedding(i, V[n], sparse=sparse_grad)  # noqa: E731
        lin = [emb("user_linear_var", users).reshape(-1, 1), emb("item_linear_var", items).reshape(-1, 1)]
        pw = [emb("user_embeds_var", users)[:, None, :], emb("item_embeds_var", items)[:, None, :]]
        if self.has_sparse:
            lin.append(F.embedding(sparse_indices, V["sparse_linear_var"], sparse=sparse_grad).squeeze(-1))
            pw.append(emb("sparse_embeds_var", sparse_indices))
        linear_term = torch.cat(lin, dim=1) @ V["linear/kernel"] + V["linear/bias"]
        pe = torch.cat(pw, dim=1)
        s = pe.sum(dim=1)
        pair = 0.5 * (s * s - (pe * pe).sum(dim=1))
        if self.use_bn:
            pair = tf_batch_norm(pair, V["bn/gamma"], V["bn/beta"], Bf["bn/moving_mean"],
                                 Bf["bn/moving_var"], training)
        pt = F.elu(pair @ V["pair/kernel"] + V["pair/bias"])                  # fm.py:168
        return (linear_term + pt).squeeze(1)                                  # fm.py:169

    def train_step(self, users, items, sparse_indices, labels):
        logits = self.forward(users, items, sparse_indices, True, True)
        loss = F.binary_cross_entropy_with_logits(logits, labels.to(logits.dtype))
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


def _prefixed(weights, prefix):
    return {k: v for k, v in weights.items() if k.startswith(prefix + "/")}


class DINOracle:
    """algorithms/din.py:165-250 + layers/attention.py:28-64 + tfops/features.py:151-218
    ("concat" item features) for plain sparse / dense columns.  Item side features enter the
    attention keys through the full `[N+1, K']` feature table exactly as the reference builds it
    (so their table gradients are dense, like TF's)."""

    def __init__(self, weights, hidden_units=(128, 64, 32), use_bn=True, max_seq_len=10,
                 item_sparse_unique=None, item_dense_unique=None, item_dense_cols=(),
                 lr=1e-3, epsilon=1e-5, dtype=torch.float32, use_tf_attention=False):
        self.use_tf_attention = use_tf_attention
        self.V = _Vars(dtype)
        for k in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var", "embedding/dense_embeds_var"):
            if k in weights:
                self.V.add(k, weights[k])
        self.att = DenseNN(self.V, "attention", _prefixed(weights, "attention"), 2, False, torch.sigmoid)
        self.mlp = DenseNN(self.V, "mlp", _prefixed(weights, "mlp"), len(hidden_units), use_bn)
        self.V.add("out/kernel", weights["out/kernel"])
        self.V.add("out/bias", weights["out/bias"])
        self.L = max_seq_len
        self.item_sparse = None if item_sparse_unique is None else torch.as_tensor(item_sparse_unique).long()
        self.item_dense = None if item_dense_unique is None else torch.as_tensor(item_dense_unique).to(dtype)
        self.item_dense_cols = list(item_dense_cols)
        self.opt = TF1Adam(lr, eps=epsilon)
        self.dtype = dtype

    def _item_seq_feats(self):                                  # tfops/features.py:151-218
        v = self.V.v
        parts = [v["item_embeds_var"]]
        if self.item_sparse is not None:
            parts.append(v["sparse_embeds_var"][self.item_sparse].flatten(1))
        if self.item_dense is not None:
            parts.append((self.item_dense[:, :, None] * v["embedding/dense_embeds_var"][self.item_dense_cols][None]).flatten(1))
        return torch.cat(parts, dim=1)

    def _attention(self, q, keys, lens):                        # layers/attention.py:28-64
        L = keys.shape[1]
        if self.use_tf_attention:                               # layers/attention.py:5-25 (keras Attention)
            s = torch.einsum("bk,blk->bl", q, keys)
            mask = torch.arange(L)[None, :] < lens[:, None]
            s = s - 1e9 * (~mask).to(s.dtype)
            return (torch.softmax(s, dim=1)[:, None, :] @ keys).squeeze(1)
        qt = q[:, None, :].expand(-1, L, -1)
        w = self.att(torch.cat([qt, keys, qt - keys, qt * keys], dim=2), False).squeeze(2)
        w = w * (keys.shape[-1] ** -0.5)
        mask = torch.arange(L)[None, :] < lens[:, None]
        w = torch.where(mask, w, torch.full_like(w, -(2.0 ** 32) + 1))
        return (torch.softmax(w, dim=1)[:, None, :] @ keys).squeeze(1)

    def forward(self, users, items, sparse, dense, seqs, lens, training=False):
        v = self.V.v
        concat = [v["user_embeds_var"][users], v["item_embeds_var"][items]]
        if sparse is not None:
            concat.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            concat.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][None]).flatten(1))
        feats = self._item_seq_feats()
        att = self._attention(feats[items], feats[seqs], lens)
        x = self.mlp(torch.cat([*concat, att], dim=1), training)
        return (x @ v["out/kernel"] + v["out/bias"]).reshape(-1)

    def train_step(self, users, items, sparse, dense, seqs, lens, labels):
        logits = self.forward(users, items, sparse, dense, seqs, lens, True)
        loss = F.binary_cross_entropy_with_logits(logits, labels.to(self.dtype))
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class YouTubeRankingOracle:
    """algorithms/youtube_ranking.py:163-246: concat [user, item, seq_embeds_pooling (layers/embedding.py:54-85: pad row
    read as 0, sum over the window / sqrt(seq_len)), sparse, dense] -> dense_nn -> Dense(1).  [UNPINNED: TF]"""

    def __init__(self, weights, n_items, hidden_units=(128, 64, 32), use_bn=True, lr=1e-3, epsilon=1e-5,
                 dtype=torch.float32):
        self.V = _Vars(dtype)
        for k in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var", "embedding/dense_embeds_var"):
            if k in weights:
                self.V.add(k, weights[k])
        self.mlp = DenseNN(self.V, "mlp", _prefixed(weights, "mlp"), len(hidden_units), use_bn)
        self.V.add("out/kernel", weights["out/kernel"])
        self.V.add("out/bias", weights["out/bias"])
        self.n_items, self.opt, self.dtype = n_items, TF1Adam(lr, eps=epsilon), dtype

    def forward(self, users, items, sparse, dense, seqs, lens, training=False):
        v = self.V.v
        rows = v["item_embeds_var"][seqs] * (seqs != self.n_items)[:, :, None].to(self.dtype)   # scatter_update(pad, 0)
        pooled = rows.sum(1) / torch.sqrt(lens.to(self.dtype))[:, None]
        concat = [v["user_embeds_var"][users], v["item_embeds_var"][items], pooled]
        if sparse is not None:
            concat.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            concat.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][None]).flatten(1))
        x = self.mlp(torch.cat(concat, dim=1), training)
        return (x @ v["out/kernel"] + v["out/bias"]).reshape(-1)

    def train_step(self, users, items, sparse, dense, seqs, lens, labels):
        logits = self.forward(users, items, sparse, dense, seqs, lens, True)
        loss = F.binary_cross_entropy_with_logits(logits, labels.to(self.dtype))
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class YouTubeRetrievalOracle:
    """algorithms/youtube_retrieval.py:165-262 + training/tf_trainer.py:133-245.  User vector: dense_nn over
    [safe_embedding_lookup_sparse(seq_embeds_var, history, combiner="sqrtn") ; user sparse ; user dense]; loss:
    `tf.nn.sampled_softmax_loss` / `tf.nn.nce_loss` restated from their documented `_compute_sampled_logits`
    (num_true = 1, remove_accidental_hits, subtract_log_q) with the candidate set GIVEN (the TF sampler's draws cannot
    be reproduced) and expected count = num_sampled / n_items for the unique uniform sampler.  Plain sparse columns
    only.  [UNPINNED: TF]"""

    def __init__(self, weights, n_items, hidden_units=(128, 64, 16), use_bn=True, norm_embed=False,
                 loss_type="sampled_softmax", lr=1e-3, epsilon=1e-5, dtype=torch.float32):
        self.V = _Vars(dtype)
        for k in ("seq_embeds_var", "item_embeds_var", "sparse_embeds_var", "embedding/item_bias_var",
                  "embedding/dense_embeds_var"):
            if k in weights:
                self.V.add(k, weights[k])
        self.mlp = DenseNN(self.V, "mlp", _prefixed(weights, "mlp"), len(hidden_units), use_bn)
        self.n_items, self.norm_embed, self.loss_type = n_items, norm_embed, loss_type
        self.opt, self.dtype = TF1Adam(lr, eps=epsilon), dtype

    def user_embeds(self, seqs, sparse, dense, training=False):
        v = self.V.v
        valid = (seqs != self.n_items).to(self.dtype)                        # pad / pruned entries
        rows = v["seq_embeds_var"][seqs.clamp(max=self.n_items - 1)] * valid[:, :, None]
        n = valid.sum(1, keepdim=True)
        pooled = rows.sum(1) / torch.sqrt(torch.where(n > 0, n, torch.ones_like(n)))   # empty bag -> 0-vector
        concat = [pooled]
        if sparse is not None:
            concat.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            concat.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][None]).flatten(1))
        out = self.mlp(torch.cat(concat, dim=1), training)
        return F.normalize(out, dim=1, eps=1e-12) if self.norm_embed else out

    def loss(self, items, seqs, sparse, dense, sampled):
        v = self.V.v
        ue = self.user_embeds(seqs, sparse, dense, True)
        w = F.normalize(v["item_embeds_var"], dim=1, eps=1e-12) if self.norm_embed else v["item_embeds_var"]
        b = v["embedding/item_bias_var"]
        true_logits = (ue * w[items]).sum(1) + b[items]
        sampled_logits = ue @ w[sampled].T + b[sampled][None, :]
        hits = (sampled[None, :] == items[:, None]).to(self.dtype)
        sampled_logits = sampled_logits + hits * (-float(torch.finfo(torch.float32).max))
        log_q = math.log(min(1.0, len(sampled) / self.n_items))
        logits = torch.cat([(true_logits - log_q)[:, None], sampled_logits - log_q], dim=1)
        labels = torch.zeros_like(logits)
        labels[:, 0] = 1.0
        if self.loss_type == "sampled_softmax":
            return -(labels * torch.log_softmax(logits, dim=1)).sum(1).mean()
        return F.binary_cross_entropy_with_logits(logits, labels, reduction="none").sum(1).mean()

    def train_step(self, items, seqs, sparse, dense, sampled):
        loss = self.loss(items, seqs, sparse, dense, sampled)
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


def export_retrieval_weights(net) -> Dict[str, torch.Tensor]:
    """Weights of a `YouTubeRetrievalNet` under the reference's variable names, on CPU."""
    t, w = net.tables, {}
    for k in ("seq_embeds_var", "item_embeds_var", "sparse_embeds_var"):
        if t.variable(k).shape[0]:
            w[k] = t.variable(k).detach().cpu().clone()
    for name, p in net.P.params.items():
        w[name] = p.detach().cpu().clone()
    st = net.mlp
    if st.bn_in is not None:
        w["mlp/bn_in/moving_mean"], w["mlp/bn_in/moving_var"] = st.bn_in.moving_mean.cpu().clone(), st.bn_in.moving_var.cpu().clone()
    for i, bn in enumerate(st.bns, start=1):
        if bn is not None:
            w[f"mlp/bn{i}/moving_mean"], w[f"mlp/bn{i}/moving_var"] = bn.moving_mean.cpu().clone(), bn.moving_var.cpu().clone()
    return w


class TransformerOracle:
    """algorithms/transformer.py:204-339 with layers/transformer.py:131-180 (positional encoding, ffn),
    layers/attention.py:5-25,67-100 (target attention; multi-head attention in the keras form taken for TF >= 2.10:
    bias-free q/k/v/output projections, query scaled by 1/sqrt(head_dim), masked scores - 1e9),
    layers/normalization.py:9-33, tfops/features.py:151-236 (the full `[N+1, K']` item feature table, "concat" or
    "elementwise").  Plain sparse columns.  [UNPINNED: TF]"""

    def __init__(self, weights, hidden_units=(128, 64, 32), use_bn=True, max_seq_len=10, num_heads=1, num_tfm_layers=1,
                 trainable_pos=True, pos_const=None, use_causal_mask=False, feat_agg_mode="concat",
                 item_sparse_unique=None, item_dense_unique=None, item_dense_cols=(), lr=1e-3, epsilon=1e-5,
                 dtype=torch.float32):
        self.V = _Vars(dtype)
        for k, w in weights.items():
            if k.startswith("mlp/"):
                continue
            self.V.add(k, w)
        self.mlp = DenseNN(self.V, "mlp", _prefixed(weights, "mlp"), len(hidden_units), use_bn,
                           activation=lambda x: x * torch.sigmoid(x))          # swish, layers/activation.py:8-9
        self.L, self.H, self.n_layers, self.causal, self.mode = max_seq_len, num_heads, num_tfm_layers, use_causal_mask, feat_agg_mode
        self.trainable_pos = trainable_pos
        self.pos_const = None if pos_const is None else torch.as_tensor(pos_const).to(dtype)
        self.item_sparse = None if item_sparse_unique is None else torch.as_tensor(item_sparse_unique).long()
        self.item_dense = None if item_dense_unique is None else torch.as_tensor(item_dense_unique).to(dtype)
        self.item_dense_cols = list(item_dense_cols)
        self.opt, self.dtype = TF1Adam(lr, eps=epsilon), dtype

    @staticmethod
    def _ln(x, scale, bias):
        mean = x.mean(-1, keepdim=True)
        return (x - mean) * torch.rsqrt(((x - mean) ** 2).mean(-1, keepdim=True) + 1e-8) * scale + bias

    @staticmethod
    def _rms(x, scale):
        return x * torch.rsqrt((x ** 2).mean(-1, keepdim=True) + 1e-8) * scale

    def _item_table(self):
        v = self.V.v
        item = v["item_embeds_var"]
        sp = v["sparse_embeds_var"][self.item_sparse] if self.item_sparse is not None else None           # [N+1, Fs, K]
        dn = None
        if self.item_dense is not None:
            dn = self.item_dense[:, :, None] * v["embedding/dense_embeds_var"][self.item_dense_cols][None]
        if self.mode == "concat":
            return torch.cat([item] + ([sp.flatten(1)] if sp is not None else []) + ([dn.flatten(1)] if dn is not None else []), dim=1)
        extra = 1.0
        if sp is not None:
            extra = extra + self._ln(sp, v["elementwise_sparse_feats/layer_norm/scale"], v["elementwise_sparse_feats/layer_norm/bias"]).sum(1)
        if dn is not None:
            extra = extra + self._ln(dn, v["elementwise_dense_feats/layer_norm/scale"], v["elementwise_dense_feats/layer_norm/bias"]).sum(1)
        return item * extra

    def _mha(self, x, s, mask):
        v, H = self.V.v, self.H
        B, T, D = x.shape
        hd = D // H
        split = lambda y: y.view(B, T, H, hd).permute(0, 2, 1, 3)  # noqa: E731
        q = split(x @ v[f"{s}/multi_head_attention/query/kernel"]) / math.sqrt(hd)
        k = split(x @ v[f"{s}/multi_head_attention/key/kernel"])
        val = split(x @ v[f"{s}/multi_head_attention/value/kernel"])
        w = q @ k.transpose(-1, -2) + (-1e9) * (1.0 - mask[:, None].to(self.dtype))
        out = (torch.softmax(w, dim=-1) @ val).permute(0, 2, 1, 3).reshape(B, T, D)
        return out @ v[f"{s}/multi_head_attention/attention_output/kernel"]

    def forward(self, users, items, sparse, dense, seqs, lens, training=False):
        v, L = self.V.v, self.L
        concat = [v["user_embeds_var"][users], v["item_embeds_var"][items]]
        if sparse is not None:
            concat.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            concat.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][None]).flatten(1))
        table = self._item_table()
        item_e, seq_e = table[items], table[seqs]
        B, K = len(items), v["item_embeds_var"].shape[1]
        pos = v["transformer/positional_encoding"] if self.trainable_pos else self.pos_const
        x = torch.cat([seq_e, pos[None].expand(B, -1, -1)], dim=2)
        mask = (torch.arange(L)[None, :] < lens[:, None])[:, None, :].expand(-1, L, -1)
        if self.causal:
            mask = mask | torch.tril(torch.ones(L, L, dtype=torch.bool))[None]
        for l in range(1, self.n_layers + 1):
            s = f"transformer_layer{l}"
            att = self._mha(self._rms(x, v[f"{s}/rms_norm_att/scale"]), s, mask) + x
            h = self._rms(att, v[f"{s}/rms_norm_ffn/scale"]) @ v[f"{s}/ffn/dense/kernel"]
            h = 0.5 * h * (1.0 + torch.erf(h / 1.4142135623730951))
            x = att + h @ v[f"{s}/ffn/dense_1/kernel"]
        x = self._rms(x, v["rms_norm_last/scale"])
        q = torch.cat([self._rms(item_e, v["rms_norm_item/scale"]), torch.ones(B, K, dtype=self.dtype)], dim=1)
        sc = torch.einsum("bk,blk->bl", q, x)
        sc = sc - 1e9 * (~(torch.arange(L)[None, :] < lens[:, None])).to(self.dtype)
        seq_out = (torch.softmax(sc, dim=1)[:, None, :] @ x).squeeze(1)
        y = self.mlp(torch.cat([*concat, seq_out], dim=1), training)
        return (y @ v["out/kernel"] + v["out/bias"]).reshape(-1)

    def train_step(self, users, items, sparse, dense, seqs, lens, labels):
        logits = self.forward(users, items, sparse, dense, seqs, lens, True)
        loss = F.binary_cross_entropy_with_logits(logits, labels.to(self.dtype))
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class SIMOracle:
    """algorithms/sim.py:191-345: projected "concat" item feature table (sim.py:195-197); first stage = masked sum of
    the long window + target -> dense_nn -> Dense(1) (:227-245); second stage = top-k inner-product search over the
    long window (:254-276), multi-head target attention over the hits (:278-291, keras form), dot-product attention
    over the short window (:293-296), concat with the other embeddings -> dense_nn -> Dense(1) (:247-252);
    output = alpha * first + beta * second, inference = second.  [UNPINNED: TF]"""

    def __init__(self, weights, hidden_units=(200, 80), use_bn=True, alpha=1.0, beta=1.0, search_topk=10,
                 long_max_len=100, short_max_len=10, num_heads=2, item_sparse_unique=None, item_dense_unique=None,
                 item_dense_cols=(), lr=1e-3, epsilon=1e-5, dtype=torch.float32):
        self.V = _Vars(dtype)
        for k, w in weights.items():
            if not (k.startswith("first_stage_mlp/") or k.startswith("second_stage_mlp/")):
                self.V.add(k, w)
        self.mlp1 = DenseNN(self.V, "first_stage_mlp", _prefixed(weights, "first_stage_mlp"), len(hidden_units), use_bn)
        self.mlp2 = DenseNN(self.V, "second_stage_mlp", _prefixed(weights, "second_stage_mlp"), len(hidden_units), use_bn)
        self.alpha, self.beta, self.topk, self.Lg, self.S, self.H = alpha, beta, search_topk, long_max_len, short_max_len, num_heads
        self.item_sparse = None if item_sparse_unique is None else torch.as_tensor(item_sparse_unique).long()
        self.item_dense = None if item_dense_unique is None else torch.as_tensor(item_dense_unique).to(dtype)
        self.item_dense_cols = list(item_dense_cols)
        self.opt, self.dtype = TF1Adam(lr, eps=epsilon), dtype

    def _seq_table(self):
        v = self.V.v
        parts = [v["item_embeds_var"]]
        if self.item_sparse is not None:
            parts.append(v["sparse_embeds_var"][self.item_sparse].flatten(1))
        if self.item_dense is not None:
            parts.append((self.item_dense[:, :, None] * v["embedding/dense_embeds_var"][self.item_dense_cols][None]).flatten(1))
        return torch.cat(parts, dim=1) @ v["seq_feats_proj/kernel"]

    def stages(self, users, items, sparse, dense, seqs, lens, training):
        v, Lg, H = self.V.v, self.Lg, self.H
        other = [v["user_embeds_var"][users], v["item_embeds_var"][items]]
        if sparse is not None:
            other.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            other.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][None]).flatten(1))
        table = self._seq_table()
        target, long, short = table[items], table[seqs[:, :Lg]], table[seqs[:, Lg:]]
        long_ok = torch.arange(Lg)[None, :] < lens[:, :1]
        pooled = torch.where(long_ok[:, :, None], long, torch.zeros_like(long)).sum(1)
        first = (self.mlp1(torch.cat([target, pooled], dim=1), training) @ v["first_stage_out/kernel"] + v["first_stage_out/bias"]).reshape(-1)
        scores = torch.where(long_ok, (target[:, None, :] @ long.transpose(1, 2)).squeeze(1), torch.full_like(long[:, :, 0], -1e9))
        idx = torch.topk(scores, self.topk, dim=1).indices
        B, K = target.shape
        top = long[torch.arange(B)[:, None], idx]
        top_ok = long_ok[torch.arange(B)[:, None], idx]
        hd = K // H
        q = (target @ v["multi_head_attention/query/kernel"]).view(B, H, 1, hd) / math.sqrt(hd)
        k = (top @ v["multi_head_attention/key/kernel"]).view(B, -1, H, hd).permute(0, 2, 1, 3)
        val = (top @ v["multi_head_attention/value/kernel"]).view(B, -1, H, hd).permute(0, 2, 1, 3)
        w = q @ k.transpose(-1, -2) + (-1e9) * (1.0 - top_ok[:, None, None, :].to(self.dtype))
        long_out = (torch.softmax(w, dim=-1) @ val).permute(0, 2, 1, 3).reshape(B, K) @ v["multi_head_attention/attention_output/kernel"]
        sc = torch.einsum("bk,blk->bl", target, short)
        sc = sc - 1e9 * (~(torch.arange(self.S)[None, :] < lens[:, 1:2])).to(self.dtype)
        short_out = (torch.softmax(sc, dim=1)[:, None, :] @ short).squeeze(1)
        x2 = self.mlp2(torch.cat([long_out, short_out, *other], dim=1), training)
        second = (x2 @ v["second_stage_out/kernel"] + v["second_stage_out/bias"]).reshape(-1)
        return first, second

    def forward(self, users, items, sparse, dense, seqs, lens):
        return self.stages(users, items, sparse, dense, seqs, lens, False)[1]

    def train_step(self, users, items, sparse, dense, seqs, lens, labels):
        first, second = self.stages(users, items, sparse, dense, seqs, lens, True)
        loss = F.binary_cross_entropy_with_logits(self.alpha * first + self.beta * second, labels.to(self.dtype))
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


class TwoTowerOracle:
    """algorithms/two_tower.py:189-410 (towers), 458-479 (adjust_logits), tfops/loss.py:56-75."""

    def __init__(self, weights, hidden_units=(128, 64, 32), use_bn=True, norm_embed=False,
                 user_dense_cols=(), item_dense_cols=(), margin=1.0, temperature=1.0,
                 use_correction=True, remove_accidental_hits=False, lr=1e-3, epsilon=1e-5,
                 dtype=torch.float32):
        self.V = _Vars(dtype)
        for k in ("user_embeds_var", "item_embeds_var", "sparse_embeds_var", "embedding/dense_embeds_var",
                  "temperature_var"):
            if k in weights:
                self.V.add(k, weights[k])
        n = len(hidden_units)
        self.user_tower = DenseNN(self.V, "user_tower", _prefixed(weights, "user_tower"), n, use_bn)
        self.item_tower = DenseNN(self.V, "item_tower", _prefixed(weights, "item_tower"), n, use_bn)
        self.norm_embed, self.margin, self.temperature = norm_embed, margin, temperature
        self.use_correction, self.remove_accidental_hits = use_correction, remove_accidental_hits
        self.ud_cols, self.id_cols = list(user_dense_cols), list(item_dense_cols)
        self.opt = TF1Adam(lr, eps=epsilon)
        self.dtype = dtype

    def _tower(self, tower, id_var, ids, sparse, dense, cols, training):
        v = self.V.v
        parts = [v[id_var][ids]]
        if sparse is not None:
            parts.append(v["sparse_embeds_var"][sparse].flatten(1))
        if dense is not None:
            parts.append((dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][cols][None]).flatten(1))
        out = tower(torch.cat(parts, dim=1), training)
        if self.norm_embed:                                     # tf.linalg.l2_normalize(epsilon=1e-12)
            out = out * torch.rsqrt(torch.clamp((out * out).sum(1, keepdim=True), min=1e-12))
        return out

    def user_embeds(self, users, sparse=None, dense=None, training=False):
        return self._tower(self.user_tower, "user_embeds_var", users, sparse, dense, self.ud_cols, training)

    def item_embeds(self, items, sparse=None, dense=None, training=False):
        return self._tower(self.item_tower, "item_embeds_var", items, sparse, dense, self.id_cols, training)

    def _adjust(self, logits, items, corrections):              # two_tower.py:458-479
        t = self.V.v["temperature_var"] if "temperature_var" in self.V.v else self.temperature
        logits = logits / t
        if self.use_correction and corrections is not None:
            logits = logits - torch.log(torch.clamp(corrections.to(self.dtype), 1e-8, 1.0)).view(1, -1)
        if self.remove_accidental_hits:
            same = (items.view(1, -1) == items.view(-1, 1)) & ~torch.eye(len(items), dtype=torch.bool)
            logits = torch.where(same, torch.full_like(logits, torch.finfo(torch.float32).min), logits)
        return logits

    def _ssl_embeds(self, idx, dense):                          # two_tower.py:295-304,348-353
        v = self.V.v
        table = torch.cat([torch.zeros((1, v["item_embeds_var"].shape[1]), dtype=self.dtype),
                           v["item_embeds_var"], v["sparse_embeds_var"]], dim=0)
        x = table[idx].flatten(1)
        if dense is not None:
            x = torch.cat([x, (dense.to(self.dtype)[:, :, None] * v["embedding/dense_embeds_var"][self.id_cols][None]).flatten(1)], dim=1)
        out = self.item_tower(x, True)
        if self.norm_embed:
            out = out * torch.rsqrt(torch.clamp((out * out).sum(1, keepdim=True), min=1e-12))
        return out

    def loss(self, loss_type, users, items, labels=None, items_neg=None, user_sparse=None, item_sparse=None,
             item_sparse_neg=None, user_dense=None, item_dense=None, item_dense_neg=None, corrections=None,
             ssl_left=None, ssl_right=None, ssl_dense=None, alpha=0.2):
        ue = self.user_embeds(users, user_sparse, user_dense, True)
        ie = self.item_embeds(items, item_sparse, item_dense, True)
        if loss_type == "cross_entropy":
            return F.binary_cross_entropy_with_logits((ue * ie).sum(1), labels.to(self.dtype))
        if loss_type == "max_margin":
            ne = self.item_embeds(items_neg, item_sparse_neg, item_dense_neg, True)
            return F.relu(self.margin + (ue * ne).sum(1) - (ue * ie).sum(1)).mean()
        if loss_type == "softmax":
            logits = self._adjust(ue @ ie.T, items, corrections)
            loss = F.cross_entropy(logits, torch.arange(len(items)))
            if ssl_left is not None:                            # tfops/loss.py:38-47
                sl, sr = self._ssl_embeds(ssl_left, ssl_dense), self._ssl_embeds(ssl_right, ssl_dense)
                t = self.V.v["temperature_var"] if "temperature_var" in self.V.v else self.temperature
                loss = loss + alpha * F.cross_entropy((sl @ sr.T) / t, torch.arange(len(items)))
            return loss
        raise ValueError(loss_type)

    def train_step(self, loss_type, *args, **kw):
        loss = self.loss(loss_type, *args, **kw)
        loss.backward()
        self.opt.step(self.V.trainable())
        return loss.detach()


def export_net_weights(net) -> Dict[str, torch.Tensor]:
    """Weights of a FeatNet / TwoTowerNet under the reference's variable names, on CPU."""
    t = net.tables
    w = {}
    for kind in ("user", "item", "sparse"):
        ev = t.variable(f"{kind}_embeds_var")
        if ev.shape[0]:
            w[f"{kind}_embeds_var"] = ev.detach().cpu().clone()
    for name, p in net.P.params.items():
        w[name] = p.detach().cpu().clone()
    for tower in ("mlp", "user_tower", "item_tower", "first_stage_mlp", "second_stage_mlp"):
        obj = getattr(net, tower, None)
        if obj is None:
            continue
        if obj.bn_in is not None:
            w[f"{tower}/bn_in/moving_mean"] = obj.bn_in.moving_mean.cpu().clone()
            w[f"{tower}/bn_in/moving_var"] = obj.bn_in.moving_var.cpu().clone()
        for i, bn in enumerate(obj.bns, start=1):
            if bn is not None:
                w[f"{tower}/bn{i}/moving_mean"] = bn.moving_mean.cpu().clone()
                w[f"{tower}/bn{i}/moving_var"] = bn.moving_var.cpu().clone()
    return w


def export_fieldnet_weights(net) -> Dict[str, torch.Tensor]:
    """Weights of a librecommender_amd FieldNet (FM/DeepFM) under the reference's variable names,
    on CPU — lets a test start the oracle from the HIP model's exact state."""
    t = net.tables
    w = {}
    for kind in ("user", "item", "sparse"):
        ev = t.variable(f"{kind}_embeds_var")
        if ev.shape[0] == 0:
            continue
        w[f"{kind}_embeds_var"] = ev.detach().cpu().clone()
        lv = t.variable(f"{kind}_linear_var").detach().cpu().clone()
        w[f"{kind}_linear_var"] = lv.reshape(-1) if kind == "sparse" else lv
    for name, p in net.P.params.items():
        w[name] = p.detach().cpu().clone()
    for obj_name, obj in (("mlp", getattr(net, "mlp", None)),):
        if obj is None:
            continue
        if obj.bn_in is not None:
            w["mlp/bn_in/moving_mean"] = obj.bn_in.moving_mean.cpu().clone()
            w["mlp/bn_in/moving_var"] = obj.bn_in.moving_var.cpu().clone()
        for i, bn in enumerate(obj.bns, start=1):
            if bn is not None:
                w[f"mlp/bn{i}/moving_mean"] = bn.moving_mean.cpu().clone()
                w[f"mlp/bn{i}/moving_var"] = bn.moving_var.cpu().clone()
    if getattr(net, "bn", None) is not None:
        w["bn/moving_mean"] = net.bn.moving_mean.cpu().clone()
        w["bn/moving_var"] = net.bn.moving_var.cpu().clone()
    return w


class LightGCNOracle:
    """algorithms/torch_modules/lightgcn_module.py:7-96 + torchops/loss.py (bpr) + training/torch_trainer.py:63-69,
    116-121 (torch.optim.Adam on the two init-embedding tables), restated on torch-CPU sparse ops.  PINNED: loss,
    gradients and the first Adam step against the reference module's own outputs (tests/golden/lightgcn.npz,
    tests/test_oracle_cpu.py::test_lightgcn_oracle_against_reference_step).  The Laplacian comes from the
    interaction list (vectorised: the reference's per-user dok fill is value-identical, lightgcn_module.py:36-61)."""

    def __init__(self, n_users, n_items, embed_size, n_layers, edge_users, edge_items, lr=1e-3, epsilon=1e-8,
                 U0=None, I0=None, seed=42):
        import numpy as np

        self.n_users, self.n_items, self.L = n_users, n_items, n_layers
        n = n_users + n_items
        pairs = np.unique(np.asarray(edge_users, dtype=np.int64) * n_items + np.asarray(edge_items, dtype=np.int64))
        u, i = pairs // n_items, pairs % n_items
        rows = np.concatenate([u, n_users + i])
        cols = np.concatenate([n_users + i, u])
        deg = np.bincount(rows, minlength=n).astype(np.float32)
        with np.errstate(divide="ignore"):
            dinv = np.power(deg, -0.5)
        dinv[np.isinf(dinv)] = 0.0
        vals = (dinv[rows] * dinv[cols]).astype(np.float32)
        self.nnz = len(vals)
        self.lap = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (n, n),
                                           dtype=torch.float32).coalesce()
        g = torch.Generator().manual_seed(seed)
        self.U = torch.nn.Parameter(torch.from_numpy(U0).clone() if U0 is not None
                                    else torch.empty(n_users, embed_size).normal_(0.0, 0.1, generator=g))
        self.I = torch.nn.Parameter(torch.from_numpy(I0).clone() if I0 is not None
                                    else torch.empty(n_items, embed_size).normal_(0.0, 0.1, generator=g))
        self.opt = torch.optim.Adam([self.U, self.I], lr=lr, eps=epsilon)

    def propagate(self):
        cur = torch.cat([self.U, self.I], dim=0)
        embs = [cur]
        for _ in range(self.L):
            cur = torch.sparse.mm(self.lap, cur)
            embs.append(cur)
        out = torch.stack(embs, dim=1).mean(dim=1)
        return out[: self.n_users], out[self.n_users:]

    def train_step(self, users, pos, neg):
        ue, ie = self.propagate()
        u, p, n = (torch.as_tensor(x).long() for x in (users, pos, neg))
        ps, ns = (ue[u] * ie[p]).sum(1), (ue[u] * ie[n]).sum(1)
        loss = -F.logsigmoid(ps - ns).mean()                        # torchops/loss.py: bpr_loss
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()
