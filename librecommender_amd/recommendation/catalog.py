"""Full-catalog scoring of the FM / DeepFM feature models WITHOUT the B x N feature cross product
(SURVEY §8 row f2).

The reference ranks by materialising, per user, one feature row for every item and running the
whole model on it (`recommendation/recommend.py:81-105`, `recommendation/preprocess.py:110-172`):
N x F' embedding gathers and an N x (F'*K) x H1 first-layer GEMM per user.  Every piece of both
models that touches the embeddings is (bi)linear in "user-side fields" + "item-side fields":

  linear term    LIN @ wl           = a_u + b_i
  FM pairwise    0.5((su+si)^2 - qu - qi),   su/si = field sums, qu/qi = sums of squares
  first Dense    BN_eval(x) @ W1    = P_u + Q_i      (x = [user fields | item fields] flattened)

so the item-side quantities (b_i, si, qi, Q_i) are computed ONCE for the whole catalog (and cached
until the tables change), the user-side ones once per user, and a (user, item) pair costs the
MLP tail (H1 -> ... -> 1) plus one K-wide dot product.  The results equal `net.forward` on the
materialised rows up to fp32 re-association (tests: 1e-4)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


def field_sides(data_info, spec=None) -> np.ndarray:
    """is_item[F'] for the field order [user, item, plain sparse cols, pooled fields, dense cols]."""
    d = data_info
    item_sp, item_dn = set(d.item_sparse_col.index), set(d.item_dense_col.index)
    n_sp, n_dn = len(d.sparse_col.name), len(d.dense_col.name)
    if spec is not None:
        plain, pooled_off = spec.plain_cols, spec.field_offset
    else:
        plain, pooled_off = list(range(n_sp)), []
    sides = [False, True]
    sides += [c in item_sp for c in plain]
    sides += [o in item_sp for o in pooled_off]
    if spec is None or spec.n_dense_cols:
        sides += [c in item_dn for c in range(n_dn)]
    return np.asarray(sides, dtype=bool)


class CatalogScorer:
    def __init__(self, model):
        self.model, self.net, self.info = model, model.net, model.data_info
        net = self.net
        self.device = model.device
        spec = getattr(net, "spec", None)
        self.is_item = torch.from_numpy(field_sides(self.info, spec)).to(self.device)
        self.deep = hasattr(net, "mlp")
        self._cache_step = None
        self._item = None

    # ---- model pieces -----------------------------------------------------------------------
    def _embed(self, users, items, sparse, dense):
        net = self.net
        if hasattr(net, "emb"):
            _, E, LIN = net.emb.forward(users, items, sparse, dense, grad=False)
            return E, LIN
        from .. import ops
        e, _, _, lin = ops.fm_embed_fwd(net.tables.embed, net._idx(users, items, sparse), lin=net.tables.lin)
        return e, lin

    def _first_layer(self):
        """(W1', b1') of Dense1(BN_eval(x)) = x @ W1' + b1'."""
        mlp, P = self.net.mlp, self.net.P
        W, b = P[mlp.layers[0].w].detach(), P[mlp.layers[0].b].detach()
        bn = mlp.bn_in
        if bn is None:
            return W, b
        s = P[bn.gamma].detach() * torch.rsqrt(bn.moving_var + bn.eps)
        return W * s[:, None], b + (P[bn.beta].detach() - bn.moving_mean * s) @ W

    def _mlp_tail(self, z1):
        """DenseStack after the first Dense, inference mode (layers/dense.py:12-49)."""
        mlp, P = self.net.mlp, self.net.P
        x = z1
        n = len(mlp.layers)
        for i, (layer, bn) in enumerate(zip(mlp.layers, mlp.bns)):
            if i > 0:
                x = torch.addmm(P[layer.b].detach(), x, P[layer.w].detach())
            if i != n - 1:
                x = mlp.act(x)
                if bn is not None:
                    x = bn(x, False)
        return x

    def _fused_tail(self):
        """(W2', b2', v3, c3) of  deep = relu(relu(z1) @ W2' + b2') @ v3 + c3  for a three-layer relu `dense_nn`
        (layers/dense.py:33-49: activation, then BatchNorm, no activation on the last layer) followed by the output
        layer's deep weights — the inference BatchNorms and the last Dense folded.  None when the stack has
        another shape / activation or the widths are not compiled."""
        from .. import ops
        mlp, net, P = self.net.mlp, self.net, self.net.P
        if not self.deep or len(mlp.layers) != 3 or mlp.act is not F.relu or not self.net.tables.embed.is_cuda:
            return None
        H1, H2 = P[mlp.layers[1].w].shape
        if not ops.pair_mlp_supported(H1, H2):
            return None

        def affine(bn, n):
            if bn is None:
                return torch.ones(n, device=self.device), torch.zeros(n, device=self.device)
            s = P[bn.gamma].detach() * torch.rsqrt(bn.moving_var + bn.eps)
            return s, P[bn.beta].detach() - bn.moving_mean * s

        W2, b2 = P[mlp.layers[1].w].detach(), P[mlp.layers[1].b].detach()
        W3, b3 = P[mlp.layers[2].w].detach(), P[mlp.layers[2].b].detach()
        s0, t0 = affine(mlp.bns[0], H1)
        s1, t1 = affine(mlp.bns[1], H2)
        wo_mlp = P[net.out.w].detach().view(-1)[1 + net.K:]
        W2f = (W2 * s0[:, None]).contiguous()
        b2f = (b2 + t0 @ W2).contiguous()
        v3 = ((W3 * s1[:, None]) @ wo_mlp).contiguous()
        c3 = float((b3 + t1 @ W3) @ wo_mlp)
        return W2f, b2f, v3, c3

    def _pair_weights(self):
        """w[K], c0 with  head(pair) = pair @ w + c0  (before the FM model's elu)."""
        net, P = self.net, self.net.P
        K = net.K
        if self.deep:
            wo = P[net.out.w].detach().view(-1)
            return wo[1:1 + K], None
        w, b = P[net.pair_dense.w].detach().view(-1), P[net.pair_dense.b].detach().view(-1)
        if net.bn is None:
            return w, b
        s = P[net.bn.gamma].detach() * torch.rsqrt(net.bn.moving_var + net.bn.eps)
        return w * s, b + ((P[net.bn.beta].detach() - net.bn.moving_mean * s) * w).sum()

    # ---- item side (cached) -----------------------------------------------------------------
    @torch.no_grad()
    def _item_side(self):
        step = (getattr(self.net, "step", None), getattr(self.info, "feat_version", 0))   # refits and feature refreshes
        if self._item is not None and self._cache_step == step:
            return self._item
        from ..bases.feat_base import merge_user_item_feats
        N, K, dev = self.model.n_items, self.net.K, self.device
        ii = self.is_item
        wl = self.net.P[self.net.linear.w].detach().view(-1)
        wp, _ = self._pair_weights()
        si = torch.empty((N, K), device=dev)
        bi = torch.empty(N, device=dev)      # linear-term part
        ci = torch.empty(N, device=dev)      # 0.5 * sum_k wp_k (si_k^2 - qi_k)
        Q = None
        if self.deep:
            W1, _ = self._first_layer()
            rows = (ii[:, None].expand(-1, K)).reshape(-1)
            W1i = W1[rows]
            Q = torch.empty((N, W1.shape[1]), device=dev)
        chunk = max(1024, (1 << 28) // max(1, ii.numel() * K * 4))      # ~256 MB of E per chunk
        for s in range(0, N, chunk):
            items = np.arange(s, min(N, s + chunk))
            users = np.zeros(len(items), dtype=np.int64)
            sparse, dense = merge_user_item_feats(self.info, users, items)
            E, LIN = self._embed(users, items, sparse, dense)
            EI = E[:, ii]
            s_ = EI.sum(1)
            q_ = (EI * EI).sum(1)
            sl = slice(s, s + len(items))
            si[sl] = s_
            ci[sl] = 0.5 * ((s_ * s_ - q_) @ wp)
            bi[sl] = LIN[:, ii] @ wl[ii]
            if self.deep:
                Q[sl] = EI.flatten(1) @ W1i
        self._item = (si, bi, ci, Q)
        self._cache_step = step
        return self._item

    # ---- scoring ----------------------------------------------------------------------------
    @torch.no_grad()
    def scores(self, user_ids, user_feats: Optional[dict] = None) -> torch.Tensor:
        """[B, n_items] logits, identical (up to fp32 re-association) to the model's forward on the
        materialised (user, item) feature rows."""
        from ..bases.feat_base import merge_user_item_feats
        from ..feature_override import override_dense, override_sparse
        net, P, K, dev = self.net, self.net.P, self.net.K, self.device
        si, bi, ci, Q = self._item_side()
        ii, iu = self.is_item, ~self.is_item
        users = np.asarray(user_ids, dtype=np.int64)
        items0 = np.zeros(len(users), dtype=np.int64)
        sparse, dense = merge_user_item_feats(self.info, users, items0)
        if user_feats is not None:
            sparse = override_sparse(self.info, sparse, user_feats) if sparse is not None else None
            dense = override_dense(self.info, dense, user_feats) if dense is not None else None
        E, LIN = self._embed(users, items0, sparse, dense)
        EU = E[:, iu]
        su, qu = EU.sum(1), (EU * EU).sum(1)
        wl = P[net.linear.w].detach().view(-1)
        bl = P[net.linear.b].detach().view(-1)
        au = LIN[:, iu] @ wl[iu] + bl                                     # [B]
        wp, c0 = self._pair_weights()
        cu = 0.5 * ((su * su - qu) @ wp)                                   # [B]
        B, N = len(users), si.shape[0]
        pair_head = cu[:, None] + ci[None, :] + (su * wp) @ si.T           # [B,N]: head(pair) - c0
        lin_term = au[:, None] + bi[None, :]
        if not self.deep:                                                  # fm.py:168-169
            return lin_term + F.elu(pair_head + c0)
        W1, b1 = self._first_layer()
        rows = (iu[:, None].expand(-1, K)).reshape(-1)
        Pu = EU.flatten(1) @ W1[rows] + b1                                 # [B,H1]
        wo, bo = P[net.out.w].detach().view(-1), P[net.out.b].detach().view(-1)
        wo_mlp = wo[1 + K:]
        out = wo[0] * lin_term + pair_head + bo
        H1 = Pu.shape[1]
        fused = self._fused_tail()
        if fused is not None:            # one MFMA kernel for the MLP tail of every pair (csrc/pair_mlp.hip)
            from .. import ops
            W2f, b2f, v3, c3 = fused
            ops.pair_mlp(Pu.contiguous(), Q, W2f, b2f, v3, c3, out, accumulate=True)
            return out
        chunk = max(256, (1 << 28) // max(1, B * H1 * 4))
        for s in range(0, N, chunk):
            z1 = Pu[:, None, :] + Q[None, s:s + chunk, :]
            h = self._mlp_tail(z1.view(-1, H1))
            out[:, s:s + chunk] += (h @ wo_mlp).view(B, -1)
        return out
