"""FM and DeepFM graphs (algorithms/fm.py:140-170, algorithms/deepfm.py:143-173).

Training step (one pass of the hot path over one batch):
  1. ``lr_fm_embed_fwd_f32``     gather F rows/sample + pairwise term (+ the MLP input e)
  2. ``lr_embed_gather_f32``     linear weights [B,F]
  3. torch (hipBLASLt)           BN / MLP / output layer / loss, autograd for their backward
  4. ``lr_segments_build``       CSR-by-row of the batch's B*F row ids (device radix sort)
  5. ``lr_fm_embed_bwd_adam_f32`` interaction backward + segment sum + row-wise Adam, fused
  6. ``lr_embed_scatter_adam_f32`` linear weights (same segments)
  7. ``lr_adam_dense_f32``        every dense parameter, one launch
Embedding rows use row-wise ("lazy") Adam by default; ``dense_adam=True`` reproduces TF1's dense
update of every row (training/tf_trainer.py:120) for parity runs on small tables.
"""
from __future__ import annotations

from typing import Sequence

import torch

from ..utils.device import to_device
import torch.nn.functional as F

from .. import ops
from ..layers import DenseParams, DenseStack, FieldTables, TFBatchNorm, TFDense
from ..layers.dense import BlockFirstLayer, FoldedL1Kernels, FusedL1IO, fused_l1_backward, fused_l1_forward
from ..layers.tail import DeepFMTail


class _FieldNet:
    def __init__(self, n_users, n_items, sparse_feature_size, n_fields, embed_size, device, seed,
                 lr, epsilon, dense_adam=False, reg=None, tables=None, sparse_offsets=None):
        self.device = device
        self.tables = tables if tables is not None else FieldTables(
            n_users, n_items, sparse_feature_size, embed_size, device, seed, sparse_offsets=sparse_offsets)
        self.F, self.K = int(n_fields), int(embed_size)
        self.P = DenseParams(device, seed)
        self.lr, self.epsilon = lr, epsilon
        self.dense_adam, self.reg = dense_adam, reg or 0.0
        self.step = 0
        self._row_slot = None
        self._bwd_ws = None
        self._seg_stream, self._seg_pending = None, None
        self._want_stats, self._stats_pending = False, None

    def _hp(self):
        return ops.adam_hp(self.lr, self.step, eps=self.epsilon, tf_style=True)

    def _segments_async(self, idx):
        """Start the segment build (radix sort + scan: small, latency-bound kernels that depend on
        the ids only) on a side stream so that it runs under the forward GEMMs; `_embedding_update`
        joins it."""
        if self._seg_stream is None:
            self._seg_stream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._seg_stream.wait_stream(cur)                 # idx is ready, last step's segments are consumed
        with torch.cuda.stream(self._seg_stream):
            self._seg_pending = self.tables.segments(idx)
            self._stats_pending = None
            frs = getattr(self.tables, "field_row_start", None)
            if self._want_stats and frs is not None and frs.numel() == self.F + 1:
                # input-BatchNorm statistics from the runs (distinct rows) instead of from e
                st = ops.fm_field_stats(self.tables.embed, self._seg_pending, frs, idx.shape[0])
                for x in st:
                    x.record_stream(cur)
                self._stats_pending = st

    def _take_stats(self):
        """Join the side stream and return the (mean, var) computed there, if any."""
        st = getattr(self, "_stats_pending", None)
        if st is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._seg_stream)
            self._stats_pending = None
        return st

    def _embedding_update(self, idx, gdeep, gpair, fsum, glin, bn_a=None, bn_c=None):
        t = self.tables
        B = idx.shape[0]
        if self._seg_pending is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._seg_stream)
            seg, self._seg_pending = self._seg_pending, None
        else:
            seg = t.segments(idx)
        hp = self._hp()
        if not self.dense_adam:
            need = ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, self.F)
            if self._bwd_ws is None or self._bwd_ws.numel() < need:
                self._bwd_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ops.fm_embed_bwd_adam(t.embed, t.m, t.v, gdeep, gpair, fsum, B, self.F, seg, hp,
                                  lin=t.lin, lin_m=t.lin_m, lin_v=t.lin_v, glin=glin.contiguous(),
                                  bn_a=bn_a, bn_c=bn_c, ws=self._bwd_ws)
        else:  # TF1 semantics: every row of every table moves every step
            e = ops.embed_gather(t.embed, idx)
            gd = gdeep.clone() if gdeep is not None else None
            if bn_a is not None:  # the folded-BN remainder: dx = G - a - c*x
                gd = gd - bn_a.view(1, self.F, self.K) - bn_c.view(1, self.F, self.K) * e
            ge = ops.fm_pairwise_bwd(e, fsum, gpair, ge=gd)
            if self._row_slot is None:
                self._row_slot = torch.full((t.V,), -1, dtype=torch.int32, device=self.device)
            grows = ops.embed_segment_sum(ge.view(-1, self.K), seg)
            ops.adam_dense(t.embed, t.m, t.v, hp, grows=grows, seg=seg, row_slot=self._row_slot, l2=self.reg)
            lrows = ops.embed_segment_sum(glin.reshape(-1, 1).contiguous(), seg)
            ops.adam_dense(t.lin, t.lin_m, t.lin_v, hp, grows=lrows, seg=seg, row_slot=self._row_slot, l2=self.reg)

    # ---- interface shared with the general nets (feat_nets.py) ---------------------------------
    def _idx(self, users, items, sparse):
        dev = self.device
        u = to_device(users, dev)
        i = to_device(items, dev)
        s = None if sparse is None else to_device(sparse, dev)
        return self.tables.global_idx(u, i, s)

    def assign_oov(self, sparse_oov_rows):
        """OOV rows := mean of the real rows (`bases/tf_base.py:310-353`)."""
        t = self.tables
        with torch.no_grad():
            for tab in (t.embed, t.lin):
                tab[t.user_off + t.n_users] = tab[t.user_off: t.user_off + t.n_users].mean(dim=0)
                tab[t.item_off + t.n_items] = tab[t.item_off: t.item_off + t.n_items].mean(dim=0)
                start = 0
                for oov in (sparse_oov_rows if sparse_oov_rows is not None else []):
                    oov = int(oov)
                    if start >= oov:
                        continue
                    tab[t.sparse_off + oov] = tab[t.sparse_off + start: t.sparse_off + oov].mean(dim=0)
                    start = oov + 1

    @staticmethod
    def loss_fn(logits, labels, loss_type="cross_entropy"):
        if loss_type == "cross_entropy":  # tfops/loss.py:10-17
            return F.binary_cross_entropy_with_logits(logits, labels)
        if loss_type == "focal":  # tfops/loss.py:56-62
            w = labels * 0.25 + (1 - labels) * 0.75
            p = torch.sigmoid(logits)
            p_t = labels * p + (1 - labels) * (1 - p)
            bce = F.binary_cross_entropy_with_logits(logits, labels, reduction="none")
            return (w * (1 - p_t) ** 2.0 * bce).mean()
        if loss_type == "mse":  # rating task, tfops/loss.py:5-8
            return F.mse_loss(logits, labels)
        raise ValueError(f"unknown loss_type: {loss_type}")


class DeepFMNet(_FieldNet):
    """algorithms/deepfm.py:143-173."""

    def __init__(self, n_users, n_items, sparse_feature_size, n_sparse_fields, embed_size=16,
                 hidden_units: Sequence[int] = (128, 64, 32), use_bn=True, dropout_rate=0.0,
                 lr=1e-3, epsilon=1e-5, seed=42, device=None, dense_adam=False, reg=None,
                 mlp_dtype: torch.dtype = torch.float32, tables=None, sparse_offsets=None,
                 bn_stats_from_segments=False, fused_l1=True, hip_tail=True):
        device = device or torch.device("cuda")
        F_ = 2 + int(n_sparse_fields)
        super().__init__(n_users, n_items, sparse_feature_size, F_, embed_size, device, seed, lr,
                         epsilon, dense_adam, reg, tables, sparse_offsets=sparse_offsets)
        self.linear = TFDense(self.P, "linear", F_, 1)                       # deepfm.py:158
        self.mlp = DenseStack(self.P, "mlp", F_ * embed_size, hidden_units, use_bn, dropout_rate)
        self.out = TFDense(self.P, "out", 1 + embed_size + self.mlp.n_out, 1)  # deepfm.py:171-172
        self.P.finalize()
        self.mlp_dtype = mlp_dtype
        # Input-BN statistics from the runs (`lr_fm_field_stats_f32`) read 38 % fewer bytes than the
        # Welford pass over e, but they need the segments BEFORE the first GEMM: measured on the bench
        # workload that puts the (contended) segment build on the critical path and costs more than it
        # saves (3.99 vs 3.86 ms/step) — available, off by default.
        self._want_stats = (bool(bn_stats_from_segments) and bool(use_bn) and mlp_dtype == torch.float32
                            and tables is None)
        # Lookup fused with the first Dense layer on the f32 MFMA pipe (csrc/deepfm_l1.hip): deep_embed
        # [B, F*K] and its gradient are never materialised.  Needs every field's row range (plain
        # sparse columns), a compiled (K, H1) shape, fp32, row-wise Adam.  (Dropout, round 4: a counter-based mask
        # inside the tail kernels — with the hand-written tail; the torch tail applies F.dropout.)
        frs = getattr(self.tables, "field_row_start", None)
        # (dense_adam — TF1's optimiser semantics — rides the same fused step: the row update is replaced by the per-row
        # gradient kernel + ONE streaming pass over every row of the tables, `_rows_update`)
        self.fused_l1 = bool(fused_l1 and tables is None and mlp_dtype == torch.float32
                             and not (reg or 0.0) and len(hidden_units) >= 1
                             and frs is not None and frs.numel() == F_ + 1
                             and getattr(self.tables, "lin", None) is not None
                             and ops.deepfm_l1_supported(embed_size, hidden_units[0]))
        # Widths the fused lookup + first-layer kernels are not compiled for (the reference's default embed_size = 16):
        # the gathered block deep_embed [B, F*K] IS materialised (lr_fm_embed_fwd_f32: 0.2 GB at cfg 2 shapes with
        # K = 16) and handed, re-cut into 32-wide blocks, to the same MFMA first-layer / fold / tail kernels
        # (layers/dense.py:BlockFirstLayer, row-major layout); the row update is lr_fm_embed_bwd_adam_f32.  No
        # autograd, no library GEMM.
        H1_ = hidden_units[0] if len(hidden_units) else 0
        self.block_l1 = bool(fused_l1 and not self.fused_l1 and tables is None and mlp_dtype == torch.float32
                             and not dense_adam and not (reg or 0.0)
                             and len(hidden_units) >= 2 and embed_size == 16 and (F_ * embed_size) % 32 == 0
                             and getattr(self.tables, "lin", None) is not None and hip_tail
                             and BlockFirstLayer.supported(32, H1_) and DeepFMTail.supported(self.mlp))
        self._blk = {}
        self._fseg = self._pack = self._wgrad = self._ge = self._idxT = self._tail = self._fold = self._grows = None
        # tail (layers after the first Dense, output layer, loss, their backward) as hand-written kernels
        self.hip_tail = bool(self.fused_l1 and hip_tail and DeepFMTail.supported(self.mlp))

    def _dense_forward(self, e, pair, lin, training, side=None, stats=None):
        B = e.shape[0]
        linear_term = self.linear(lin)                                      # [B,1]
        deep_in = e.view(B, self.F * self.K)
        if self.mlp_dtype != torch.float32:
            with torch.autocast("cuda", dtype=self.mlp_dtype):
                deep = self.mlp(deep_in, training).float()
        else:
            deep = self.mlp(deep_in, training, side, stats)
        concat = torch.cat([linear_term, pair, deep], dim=1)                # deepfm.py:171
        return self.out(concat).squeeze(1)

    @torch.no_grad()
    def forward(self, idx=None, items=None, sparse=None, **_) -> torch.Tensor:
        if items is not None:                       # (users, items, sparse=...) interface
            idx = self._idx(idx, items, sparse)
        if self.fused_l1:
            io = FusedL1IO(self.tables.embed, self.tables.lin, idx, None, self.F, self.K, pack_bufs=self._pack_bufs())
            return self._fused_tail(self.mlp.fused_first(io, training=False), io, training=False)
        e, pair, _, lin = ops.fm_embed_fwd(self.tables.embed, idx, lin=self.tables.lin)
        return self._dense_forward(e, pair, lin, training=False, side={})

    # ---- fused lookup + first layer path --------------------------------------------------------
    def _pack_bufs(self):
        if self._pack is None:
            H1 = self.P[self.mlp.layers[0].w].shape[1]
            # (the buffers carry the layer's arithmetic — ops.L1_ARITH at this moment: split-bf16 planes where compiled)
            self._pack = ops.deepfm_l1_pack_bufs(self.F, self.K, H1, self.device)
        return self._pack

    @property
    def l1_arith(self) -> str:
        """Arithmetic of the fused first layer's contractions: 'split_bf16' or 'f32_chain' (see ops.L1_ARITH)."""
        return "split_bf16" if self._pack_bufs()[0].dtype == torch.uint8 else "f32_chain"

    def _fused_tail(self, z1, io, training):
        deep = self.mlp.tail(z1, training)
        linear_term = self.linear(io.lin_out)                               # deepfm.py:158
        concat = torch.cat([linear_term, io.pair, deep], dim=1)             # deepfm.py:171
        return self.out(concat).squeeze(1)

    def _fused_core(self, idx, labels, loss_type, hp):
        """One fused training step enqueued on the current stream.  `hp`: `AdamHP` (eager) or an
        `ops.AdamCoefBuffer` (hipGraph capture: nothing step-dependent may be a kernel argument)."""
        t, B, F_, K = self.tables, idx.shape[0], self.F, self.K
        dev = self.device
        if self._fseg is None or self._fseg.B_max < B:
            self._graphs = {}       # graphs captured for smaller batches hold the addresses of the buffers replaced here
            self._fseg = ops.FieldSegmentBuilder(B, F_, t.V, dev)
            self._idxT = torch.empty((F_, B), dtype=torch.int32, device=dev)
            self._ge = torch.empty((B * F_ + 1, K), dtype=torch.float32, device=dev)
            H1 = self.P[self.mlp.layers[0].w].shape[1]
            nch = ops.deepfm_l1_wgrad_chunks(B, F_, K, H1, self.l1_arith)
            self._wgrad = torch.empty((nch, F_ * K, H1), dtype=torch.float32, device=dev)
        same = B == self._fseg.B_max
        idxT = ops.idx_transpose(idx, out=self._idxT if same else None)
        seg = self._fseg.build(idxT, t.field_row_start)
        io = FusedL1IO(t.embed, t.lin, idx, idxT, F_, K, pack_bufs=self._pack_bufs(),
                       wgrad_buf=self._wgrad if same else None)
        if self.hip_tail and loss_type == "cross_entropy":
            return self._fused_core_hip_tail(io, seg, labels, hp, same)
        stats = ops.fm_field_stats(t.embed, seg, t.field_row_start, B) if self.mlp.bn_in is not None else None
        self.P.zero_grad()
        z1 = self.mlp.fused_first(io, training=True, stats=stats)
        logits = self._fused_tail(z1, io, training=True)
        logits.retain_grad()
        loss = self.loss_fn(logits, labels, loss_type)
        loss.backward()
        with torch.no_grad():
            gl = logits.grad.contiguous()                                   # d loss / d logit [B]
            w_out = self.P[self.out.w]                                      # [1 + K + n_out, 1]
            wp = w_out[1:1 + K, 0] + 0.0                                    # weights of the pairwise term (own, 16-byte aligned storage; an elementwise KERNEL: a clone() would be a memcpy node in a captured step)
            lin_scale = w_out[0, 0] * self.P[self.linear.w][:, 0]
            ge = ops.deepfm_l1_dgrad(io.gz, io.WpB, K, F_, seg.slotT, gl=gl, wp=wp, fsum=io.fsum,
                                     out=self._ge if same else None)
            self._rows_update(ge, seg, hp, B, gl, wp, io, lin_scale, same)
            self.P.adam_step(hp)
        self._last_step = (io, gl, wp, seg)     # by-products of the last step (diagnostics; a few MB)
        return loss.detach()

    @torch.no_grad()
    def _fused_core_hip_tail(self, io, seg, labels, hp, same):
        """The step without autograd: first layer (MFMA kernels) + tail (csrc/deepfm_tail.hip) + the BatchNorm
        fold algebra of the first layer (a few elementwise torch ops on [F*K, H1] tensors)."""
        t, P, mlp = self.tables, self.P, self.mlp
        B, F_, K = io.idx.shape[0], self.F, self.K
        bn, l0 = mlp.bn_in, mlp.layers[0]
        if self._fold is None and FoldedL1Kernels.supported(P[l0.w].shape[1]):
            self._fold = FoldedL1Kernels(P, bn, l0, F_, K, self.device)
        if self._tail is None:
            self._tail = DeepFMTail(P, mlp, self.linear, self.out, F_, K, self.device)
            self._tail.on_release = lambda: setattr(self, "_graphs", {})
        if self._fold is not None:              # BatchNorm-fold algebra on the device kernels (csrc/deepfm_fold.hip)
            z1 = self._fold.forward(io, seg, t.field_row_start, B)
            loss, gl, gz1, sgz1 = self._tail.run(z1, io.pair, io.lin_out, labels)
            self._fold.backward(io, gz1, sgz1)
        else:
            if bn is not None:
                mean, var = ops.fm_field_stats(t.embed, seg, t.field_row_start, B)
                bn.moving_mean.mul_(bn.momentum).add_(mean, alpha=1 - bn.momentum)
                bn.moving_var.mul_(bn.momentum).add_(var, alpha=1 - bn.momentum)
                inv = torch.rsqrt(var + bn.eps)
                gamma, beta = P[bn.gamma], P[bn.beta]
            else:
                mean = inv = gamma = beta = None
            z1 = fused_l1_forward(gamma, beta, P[l0.w], P[l0.b], mean, inv, io)
            loss, gl, gz1, sgz1 = self._tail.run(z1, io.pair, io.lin_out, labels)
            dgamma, dbeta, dW, db = fused_l1_backward(gamma, beta, P[l0.w], mean, inv, io, gz1, sgz1)
            P[l0.w].grad.copy_(dW)
            P[l0.b].grad.copy_(db)
            if bn is not None:
                P[bn.gamma].grad.copy_(dgamma)
                P[bn.beta].grad.copy_(dbeta)
        w_out = P[self.out.w]
        wp = w_out[1:1 + K, 0] + 0.0       # own 16-byte aligned storage; a kernel, not a memcpy node (see GraphRunner)
        lin_scale = w_out[0, 0] * P[self.linear.w][:, 0]
        ge = ops.deepfm_l1_dgrad(io.gz, io.WpB, K, F_, seg.slotT, gl=gl, wp=wp, fsum=io.fsum,
                                 out=self._ge if same else None)
        self._rows_update(ge, seg, hp, B, gl, wp, io, lin_scale, same)
        P.adam_step(hp)
        self._last_step = (io, gl, wp, seg)
        return loss

    def _rows_update(self, ge, seg, hp, B, gl, wp, io, lin_scale, same):
        """The embedding / linear rows' optimiser step from the run-ordered per-position gradients `ge`.  Default: row-wise Adam
        on the rows of this batch (one kernel).  `dense_adam`: the reference's own semantics (training/tf_trainer.py:120) — the
        per-row gradients go to compact arrays and ONE streaming pass moves every row of both tables."""
        t, F_ = self.tables, self.F
        need = ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F_)
        if self._bwd_ws is None or self._bwd_ws.numel() < need:
            self._bwd_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if not self.dense_adam:
            ops.fm_rows_adam(t.embed, t.m, t.v, ge, seg, hp, B, F_, gl=gl, wp=wp, lin=t.lin, lin_m=t.lin_m,
                             lin_v=t.lin_v, bn_a=io.bn_a, bn_c=io.bn_c, lin_scale=lin_scale, ws=self._bwd_ws)
            return
        if self._row_slot is None:
            self._row_slot = torch.full((t.V,), -1, dtype=torch.int32, device=self.device)
        if self._grows is None or self._grows[0].shape[0] < B * F_:
            self._graphs = {}
            self._grows = (torch.empty((B * F_, self.K), dtype=torch.float32, device=self.device),
                           torch.empty(B * F_, dtype=torch.float32, device=self.device))
        grows, glin = ops.fm_rows_grad_compact(t.embed, t.lin, ge, seg, B, F_, gl, wp, bn_a=io.bn_a, bn_c=io.bn_c,
                                               lin_scale=lin_scale, ws=self._bwd_ws, out=self._grows)
        ops.adam_dense_rows(t.embed, t.m, t.v, hp, grows, seg, self._row_slot, lin=t.lin, lin_m=t.lin_m, lin_v=t.lin_v,
                            glin_rows=glin)

    @torch.no_grad()
    def _block_step(self, idx, labels):
        """One training step with the first layer on the materialised block (see `block_l1` in `__init__`)."""
        t, P, mlp = self.tables, self.P, self.mlp
        B, F_, K = idx.shape[0], self.F, self.K
        st = self._blk.get(B)
        if st is None:
            Pn = F_ * K // 32
            st = self._blk[B] = dict(
                l1=BlockFirstLayer(P, mlp.bn_in, mlp.layers[0], Pn, 32, B, self.device, layout="rowmajor"),
                tail=DeepFMTail(P, mlp, self.linear, self.out, F_, K, self.device),
                gbuf=torch.empty((B * Pn + 1, 32), dtype=torch.float32, device=self.device),
                ws=torch.empty(ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F_), dtype=torch.uint8, device=self.device))
        e, pair, fsum, lin_out = ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
        z1 = st["l1"].forward(e.view(B, F_ * K))
        loss, gl, gz1, sgz1 = st["tail"].run(z1, pair, lin_out, labels)
        st["l1"].backward(gz1, sgz1, st["gbuf"])
        G = st["gbuf"][: B * (F_ * K // 32)].view(B, F_, K)                 # d loss / d deep_embed (BatchNorm terms included)
        w_out = P[self.out.w]
        gpair = gl[:, None] * w_out[1:1 + K, 0][None, :]                    # deepfm.py:171-172: the FM term feeds one Dense(1)
        glin = gl[:, None] * (w_out[0, 0] * P[self.linear.w][:, 0])[None, :]
        seg = t.segments(idx)
        ops.fm_embed_bwd_adam(t.embed, t.m, t.v, G, gpair.contiguous(), fsum, B, F_, seg, self._hp(), lin=t.lin, lin_m=t.lin_m,
                              lin_v=t.lin_v, glin=glin.contiguous(), ws=st["ws"])
        P.adam_step(self._hp())
        return loss

    def enable_graph(self, flag: bool = True, warm_steps: int = 2) -> None:
        """Replay the fused training step as ONE hipGraph per (batch shape, loss) — the reference runs a
        step as one `sess.run` (training/tf_trainer.py:76-101).  The first `warm_steps` steps of a shape
        run eagerly (lazy initialisation outside the capture), the next one is captured and every step
        from then on is: copy ids / labels into the graph's static inputs, write the step's Adam
        coefficients (`lr_adam_coef_store`), replay — all three on a dedicated non-default stream that is
        event-ordered against the caller's stream (`nets/din_fused.py:GraphRunner`).  The returned loss is the
        graph's static output tensor (overwritten by the next step)."""
        self._use_graph, self._graph_warm = bool(flag), int(warm_steps)
        if not flag:
            self._graphs = {}
            if getattr(self, "_runner", None) is not None:
                self._runner.clear()
                self._graphs = self._runner.graphs

    def _train_step_fused(self, idx, labels, loss_type):
        # (dropout: the tail kernels take the step's mask seed by value — a replayed graph would repeat one mask)
        if not getattr(self, "_use_graph", False) or self.mlp.dropout_rate:
            return self._fused_core(idx, labels, loss_type, self._hp())
        from .din_fused import GraphRunner

        if getattr(self, "_runner", None) is None:
            self._runner = GraphRunner(self.device)
            if getattr(self, "_lazy_scope", None) is not None:     # created inside a trainer's `lazy_join(model=...)` scope
                self._lazy_scope()
        if not hasattr(self, "_graphs") or self._graphs is not self._runner.graphs:
            self._runner.clear()                    # `self._graphs = {}` elsewhere means: forget every captured graph
            self._graphs = self._runner.graphs
        key = (tuple(idx.shape), loss_type)
        st = self._graphs.setdefault(key, {"seen": 0})
        if "graph" not in st:
            # Eager steps (the first steps of a shape; the short last batch of an epoch) run on the caller's stream.
            # Replays run on the runner's dedicated stream, event-ordered against the caller's stream on both sides
            # (`GraphRunner.replay`), so an eager step that re-uses the segment / gradient workspaces of the replays
            # starts after they finished and the next replay starts after it — no device-wide synchronisation.
            st["seen"] = st.get("seen", 0) + 1
            if st["seen"] <= self._graph_warm:
                self._runner.join()
                return self._fused_core(idx, labels, loss_type, self._hp())
            st["idx"], st["labels"] = idx.clone(), labels.clone()
            st["coef"] = ops.AdamCoefBuffer(self.device)
            st["coef"].set(self._hp())
            from .din_fused import GraphNotCapturable

            try:
                self._runner.capture(key, lambda: self._fused_core(st["idx"], st["labels"], loss_type, st["coef"]))
            except GraphNotCapturable as e:       # (a memset / memcpy node in the step): this shape launches eagerly from now on
                import warnings

                warnings.warn(f"DeepFM step of shape {key} is not captured: {e}")
                st = self._graphs.setdefault(key, {})
                st["seen"] = -(1 << 60)
                return self._fused_core(idx, labels, loss_type, self._hp())
            # tensors allocated INSIDE the capture (z1 / pair / fsum / lin_out of the first layer, the output-weight
            # copy) are addressed by the graph's kernel nodes for as long as it is replayed: keep them referenced
            # (`_last_step` is overwritten by the next eager step of another batch shape)
            st["keep"] = self._last_step
            return self._runner.replay(key, lambda: None)

        def feed():
            st["idx"].copy_(idx, non_blocking=True)
            st["labels"].copy_(labels, non_blocking=True)
            st["coef"].set(self._hp())

        return self._runner.replay(key, feed, (idx, labels))

    def train_step(self, idx, labels, labels2=None, loss_type="cross_entropy", sparse=None, **_) -> torch.Tensor:
        if labels2 is not None:                     # (users, items, labels, sparse=...) interface
            idx = self._idx(idx, labels, sparse)
            labels = torch.as_tensor(labels2, device=self.device, dtype=torch.float32)
        self.step += 1
        if self.fused_l1 and idx.shape[0] <= ops.FieldSegmentBuilder.MAX_B:
            return self._train_step_fused(idx, labels, loss_type)
        if self.block_l1 and loss_type == "cross_entropy":
            return self._block_step(idx, labels)
        t = self.tables
        self._segments_async(idx)
        e, pair, fsum, lin = ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
        e.requires_grad_(True)
        pair.requires_grad_(True)
        lin.requires_grad_(True)
        self.P.zero_grad()
        side = {} if self.mlp_dtype == torch.float32 else None
        logits = self._dense_forward(e, pair, lin, training=True, side=side, stats=self._take_stats())
        loss = self.loss_fn(logits, labels, loss_type)
        loss.backward()
        with torch.no_grad():
            side = side or {}
            self._embedding_update(idx, e.grad, pair.grad, fsum, lin.grad, side.get("bn_a"), side.get("bn_c"))
            self.P.adam_step(self._hp())
        return loss.detach()


class FMNet(_FieldNet):
    """algorithms/fm.py:140-170: linear term + Dense(1, elu)(BN(pairwise term))."""

    def __init__(self, n_users, n_items, sparse_feature_size, n_sparse_fields, embed_size=16,
                 use_bn=True, lr=1e-3, epsilon=1e-5, seed=42, device=None, dense_adam=False, reg=None, tables=None):
        device = device or torch.device("cuda")
        F_ = 2 + int(n_sparse_fields)
        super().__init__(n_users, n_items, sparse_feature_size, F_, embed_size, device, seed, lr,
                         epsilon, dense_adam, reg, tables)
        self.linear = TFDense(self.P, "linear", F_, 1)                       # fm.py:155
        self.bn = TFBatchNorm(self.P, "bn", embed_size) if use_bn else None  # fm.py:164-167
        self.pair_dense = TFDense(self.P, "pair", embed_size, 1)             # fm.py:168
        self.P.finalize()

    def _dense_forward(self, pair, lin, training):
        linear_term = self.linear(lin)
        x = self.bn(pair, training) if self.bn is not None else pair
        return (linear_term + F.elu(self.pair_dense(x))).squeeze(1)          # fm.py:168-169

    @torch.no_grad()
    def forward(self, idx=None, items=None, sparse=None, **_):
        if items is not None:
            idx = self._idx(idx, items, sparse)
        _, pair, _, lin = ops.fm_embed_fwd(self.tables.embed, idx, want_e=False, lin=self.tables.lin)
        return self._dense_forward(pair, lin, training=False)

    def train_step(self, idx, labels, labels2=None, loss_type="cross_entropy", sparse=None, **_):
        if labels2 is not None:
            idx = self._idx(idx, labels, sparse)
            labels = torch.as_tensor(labels2, device=self.device, dtype=torch.float32)
        self.step += 1
        t = self.tables
        self._segments_async(idx)
        _, pair, fsum, lin = ops.fm_embed_fwd(t.embed, idx, want_e=False, lin=t.lin)
        pair.requires_grad_(True)
        lin.requires_grad_(True)
        self.P.zero_grad()
        loss = self.loss_fn(self._dense_forward(pair, lin, True), labels, loss_type)
        loss.backward()
        with torch.no_grad():
            self._embedding_update(idx, None, pair.grad, fsum, lin.grad)
            self.P.adam_step(self._hp())
        return loss.detach()


class ShardedDeepFMNet(DeepFMNet):
    """DeepFM with row-sharded tables (one process per GPU).  The batch is
    data-parallel; ``idx`` holds GLOBAL row ids of this rank's samples.  The loss is the mean over
    the global batch (local mean / world); BatchNorm statistics (and its backward sums) are those of the GLOBAL batch
    (small all-reduces at every normalisation: `TFBatchNorm.sync`, the `sync` hooks of the fused kernels), so N ranks
    compute the step one rank would compute on the concatenated batch."""

    def __init__(self, n_rows_global, n_sparse_fields, embed_size=16, hidden_units=(128, 64, 32),
                 use_bn=True, lr=1e-3, epsilon=1e-5, seed=42, device=None, kern=None, group=None,
                 field_row_start=None):
        """`field_row_start` [F + 1] (first GLOBAL row of every field, strictly increasing): with it — and the HIP
        kernels, a compiled (embed_size, first layer) shape, a cross-entropy loss and a batch of at most 16,384
        samples — the step runs the fused kernels of the single-GPU path on the step's row cache (lookup fused
        with the first Dense layer, hand-written tail, run-ordered per-row gradients)."""
        from ..parallel import HipKernels, ShardedFieldTables

        import torch.distributed as dist

        self.kern = kern or HipKernels()
        self.group = group
        self.world = dist.get_world_size(group)
        device = device or torch.device("cuda")
        tables = ShardedFieldTables(n_rows_global, embed_size, device, self.kern, group=group, seed=seed)
        super().__init__(0, 0, 0, n_sparse_fields, embed_size, hidden_units, use_bn, 0.0, lr, epsilon,
                         seed, device, tables=tables)
        self.n_rows_global = int(n_rows_global)
        self.field_row_start = None
        self._sh = None
        from ..parallel import rank_average

        self._sync = rank_average(group)      # BatchNorm over the GLOBAL batch: the N-rank step is the 1-rank step
        self.mlp.set_sync(self._sync)
        if field_row_start is not None and isinstance(self.kern, HipKernels) and device.type == "cuda":
            frs = torch.as_tensor(field_row_start, dtype=torch.int64)
            H1 = self.P[self.mlp.layers[0].w].shape[1]
            if (frs.numel() == self.F + 1 and bool((frs[1:] > frs[:-1]).all()) and ops.deepfm_l1_supported(embed_size, H1)
                    and FoldedL1Kernels.supported(H1) and DeepFMTail.supported(self.mlp)):
                self.field_row_start = frs.to(torch.int32).to(device)
                tables.set_fields(self.field_row_start)       # exchange plans come off the field-wise sort of the ids

    def _train_step_fused_sharded(self, idx, labels, next_idx):
        """The fused single-GPU step (`DeepFMNet._fused_core_hip_tail`) on the row cache of this step's exchange:
        table = cache, ids = cache slots; the per-field runs of the GLOBAL ids give the run order of the row
        gradients, which `lr_fm_rows_grad_f32` sums per row and writes at the rows' cache slots for the
        gradient all-to-all."""
        from ..parallel import allreduce_sum_

        t, P, mlp, dev = self.tables, self.P, self.mlp, self.device
        B, F_, K, W = idx.shape[0], self.F, self.K, self.world
        ctx = t.lookup(idx)
        if next_idx is not None:        # the next plan's few kernels go in FRONT of this step's: its host read never stalls
            t.prefetch(next_idx)
        if self._sh is None or self._sh["B"] != B:
            H1 = P[mlp.layers[0].w].shape[1]
            nch = ops.deepfm_l1_wgrad_chunks(B, F_, K, H1, self.l1_arith)
            self._sh = dict(B=B, ge=torch.empty((B * F_ + 1, K), dtype=torch.float32, device=dev),
                            wgrad=torch.empty((nch, F_ * K, H1), dtype=torch.float32, device=dev))
        sh = self._sh
        if ctx.fseg is not None:        # the plan already holds the field-wise runs and both slot layouts
            seg, slots, slotsT = ctx.fseg, ctx.slots, ctx.slotsT
        else:
            if "fseg" not in sh:
                sh.update(fseg=ops.FieldSegmentBuilder(B, F_, self.n_rows_global, dev),
                          idxT=torch.empty((F_, B), dtype=torch.int32, device=dev),
                          slotsT=torch.empty((F_, B), dtype=torch.int32, device=dev))
            idxT = ops.idx_transpose(idx, out=sh["idxT"])
            seg = sh["fseg"].build(idxT, self.field_row_start)          # per-field runs of the global ids
            slots = ctx.slots.contiguous()                              # [B, F] position -> cache row
            slotsT = ops.idx_transpose(slots, out=sh["slotsT"])
        io = FusedL1IO(ctx.cache, ctx.lin_cache, slots, slotsT, F_, K, pack_bufs=self._pack_bufs(), wgrad_buf=sh["wgrad"])
        if self._fold is None:
            self._fold = FoldedL1Kernels(P, mlp.bn_in, mlp.layers[0], F_, K, dev)
        if self._tail is None:
            self._tail = DeepFMTail(P, mlp, self.linear, self.out, F_, K, dev)
        sync = self._sync if W > 1 else None        # BatchNorm over the global batch (small all-reduces of partial sums)
        z1 = self._fold.forward(io, seg, self.field_row_start, B, cache_slots=slots.view(-1), sync=sync)
        loss, gl, gz1, sgz1 = self._tail.run(z1, io.pair, io.lin_out, labels, sync=sync)
        if W > 1:                       # global-batch mean: every gradient of this rank carries 1 / W
            P.grad.mul_(1.0 / W)        # the tail's parameter gradients (the first layer's are written below)
            gl.mul_(1.0 / W)
            gz1.mul_(1.0 / W)
            sgz1.mul_(1.0 / W)
        self._fold.backward(io, gz1, sgz1, sync=sync)
        w_out = P[self.out.w]
        wp = w_out[1:1 + K, 0] + 0.0       # own 16-byte aligned storage; a kernel, not a memcpy node (see GraphRunner)
        lin_scale = w_out[0, 0] * P[self.linear.w][:, 0]
        ge = ops.deepfm_l1_dgrad(io.gz, io.WpB, K, F_, seg.slotT, gl=gl, wp=wp, fsum=io.fsum, out=sh["ge"])
        need = ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F_)
        if self._bwd_ws is None or self._bwd_ws.numel() < need:
            self._bwd_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        grows, glin_rows = ops.fm_rows_grad(ctx.cache, ctx.lin_cache, ge, seg, B, F_, slots.view(-1), gl, wp,
                                            bn_a=io.bn_a, bn_c=io.bn_c, lin_scale=lin_scale, ws=self._bwd_ws)
        hp = self.kern.adam_hp(self.lr, self.step, self.epsilon)
        t.apply_gradients(ctx, grows, glin_rows, hp)
        allreduce_sum_(P.grad, self.group)
        self.kern.dense_adam(P.flat, P.m, P.v, P.grad, hp)
        return loss

    @torch.no_grad()
    def forward(self, idx=None, items=None, sparse=None, **_):
        """`forward(idx)` with GLOBAL rows, or the feature models' (users, items, sparse=...) interface once the id-space
        layout is known (`tables.set_layout`).  A collective: every rank calls it with its own rows."""
        if items is not None:
            idx = self._idx(idx, items, sparse)
        ctx = self.tables.lookup(idx)
        e, pair, _, lin = self.kern.fm_fwd(ctx.cache, ctx.lin_cache, ctx.slots)
        return self._dense_forward(e, pair, lin, training=False, side={})

    def assign_oov(self, sparse_oov_rows):
        self.tables.assign_oov(sparse_oov_rows)

    def train_step(self, idx, labels, loss_type="cross_entropy", next_idx=None):
        """`next_idx`: the NEXT batch's ids (already resident): its exchange plan (de-duplication + per-peer
        counts, the only host read of a step) is then built beside this step instead of in front of the next."""
        from ..parallel import allreduce_sum_

        self.step += 1
        B = idx.shape[0]
        if self.field_row_start is not None and loss_type == "cross_entropy" and B <= ops.FieldSegmentBuilder.MAX_B:
            with torch.no_grad():
                return self._train_step_fused_sharded(idx, labels, next_idx)
        ctx = self.tables.lookup(idx)
        e, pair, fsum, lin = self.kern.fm_fwd(ctx.cache, ctx.lin_cache, ctx.slots)
        e.requires_grad_(True)
        pair.requires_grad_(True)
        lin.requires_grad_(True)
        self.P.zero_grad()
        side = {}
        logits = self._dense_forward(e, pair, lin, training=True, side=side)
        loss = self.loss_fn(logits, labels, loss_type)
        (loss / self.world).backward()              # global-batch mean
        with torch.no_grad():
            hp = self.kern.adam_hp(self.lr, self.step, self.epsilon)
            grows, glin_rows = self.kern.fm_bwd_rows(ctx.cache, e.grad, pair.grad, fsum, B, self.F, ctx.seg,
                                                     lin.grad.contiguous(), side.get("bn_a"), side.get("bn_c"))
            self.tables.apply_gradients(ctx, grows, glin_rows, hp)
            allreduce_sum_(self.P.grad, self.group)  # each rank holds (1/W) d(local mean loss)
            self.kern.dense_adam(self.P.flat, self.P.m, self.P.v, self.P.grad, hp)
            if next_idx is not None:
                self.tables.prefetch(next_idx)
        return loss.detach()


class ShardedFMNet(FMNet):
    """FM (algorithms/fm.py:140-170: linear term + Dense(1, elu)(BN(pairwise term))) with row-sharded tables, one process
    per GPU (round 4: FM had no multi-GPU net).  The step is `ShardedDeepFMNet`'s exchange without an MLP:
    ids all-to-all -> owners gather -> rows all-to-all (de-duplicated row cache) -> pairwise term + linear weights from the
    cache (`lr_fm_embed_fwd_f32` over cache slots) -> the two small dense layers (replicated) -> loss / W -> per-row
    gradients (`lr_fm_embed_bwd_rows_f32`, no deep-term gradient) -> all-to-all to the owners -> owners sum across peers +
    row-wise Adam; dense gradients: one all-reduce.  BatchNorm over the pairwise term uses the GLOBAL batch statistics
    (`TFBatchNorm.sync`), so N ranks compute the step one rank would compute on the concatenated batch."""

    def __init__(self, n_rows_global, n_sparse_fields, embed_size=16, use_bn=True, lr=1e-3, epsilon=1e-5, seed=42,
                 device=None, kern=None, group=None):
        import torch.distributed as dist

        from ..parallel import HipKernels, ShardedFieldTables, rank_average

        self.kern = kern or HipKernels()
        self.group = group
        self.world = dist.get_world_size(group)
        device = device or torch.device("cuda")
        tables = ShardedFieldTables(n_rows_global, embed_size, device, self.kern, group=group, seed=seed)
        super().__init__(0, 0, 0, n_sparse_fields, embed_size, use_bn, lr, epsilon, seed, device, tables=tables)
        self.n_rows_global = int(n_rows_global)
        if self.bn is not None:
            self.bn.sync = rank_average(group)

    @torch.no_grad()
    def forward(self, idx=None, items=None, sparse=None, **_):
        """`forward(idx)` with GLOBAL rows, or (users, items, sparse=...) once `tables.set_layout` is known.  A collective."""
        if items is not None:
            idx = self._idx(idx, items, sparse)
        ctx = self.tables.lookup(idx)
        _, pair, _, lin = self.kern.fm_fwd(ctx.cache, ctx.lin_cache, ctx.slots)
        return self._dense_forward(pair, lin, training=False)

    def assign_oov(self, sparse_oov_rows):
        self.tables.assign_oov(sparse_oov_rows)

    def train_step(self, idx, labels, loss_type="cross_entropy", next_idx=None):
        from ..parallel import allreduce_sum_

        self.step += 1
        B = idx.shape[0]
        ctx = self.tables.lookup(idx)
        _, pair, fsum, lin = self.kern.fm_fwd(ctx.cache, ctx.lin_cache, ctx.slots)
        pair.requires_grad_(True)
        lin.requires_grad_(True)
        self.P.zero_grad()
        loss = self.loss_fn(self._dense_forward(pair, lin, True), labels, loss_type)
        (loss / self.world).backward()              # global-batch mean
        with torch.no_grad():
            hp = self.kern.adam_hp(self.lr, self.step, self.epsilon)
            grows, glin_rows = self.kern.fm_bwd_rows(ctx.cache, None, pair.grad.contiguous(), fsum, B, self.F, ctx.seg,
                                                     lin.grad.contiguous(), None, None)
            self.tables.apply_gradients(ctx, grows, glin_rows, hp)
            allreduce_sum_(self.P.grad, self.group)
            self.kern.dense_adam(self.P.flat, self.P.m, self.P.v, self.P.grad, hp)
            if next_idx is not None:
                self.tables.prefetch(next_idx)
        return loss.detach()
