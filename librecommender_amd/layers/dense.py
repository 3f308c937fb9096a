"""Dense part of the feature models: ``dense_nn`` / ``tf_dense`` / ``tf.layers.batch_normalization``
(layers/dense.py:12-80 of the reference) on device.

These are plain dense contractions: they run through torch (hipBLASLt) with autograd — they
are not part of the hand-written hot path.  All dense parameters live in ONE flat fp32 buffer
(views per tensor) with one flat gradient buffer, so the TF-style Adam update of every dense
parameter is a single ``lr_adam_dense_f32`` launch.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from .. import _lib, ops


class DenseParams:
    """Flat storage for dense-layer parameters (+ flat grads, Adam moments)."""

    def __init__(self, device: torch.device, seed: int = 42):
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + 1)
        self._specs = []  # (name, shape, init)
        self.params = {}
        self.flat = self.grad = self.m = self.v = None

    def add(self, name: str, shape: Sequence[int], init: str) -> str:
        assert self.flat is None, "finalize() already called"
        self._specs.append((name, tuple(shape), init))
        return name

    def finalize(self) -> None:
        # every tensor starts on a 16-byte boundary of the flat buffer (the kernels read weights 16 bytes at a time)
        total = sum(-(-math.prod(s) // 4) * 4 for _, s, _ in self._specs)
        pad = 0
        self.flat = torch.zeros(total + pad, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        off = 0
        for name, shape, init in self._specs:
            n = math.prod(shape)
            p = self.flat[off:off + n].view(shape)
            if init == "glorot_uniform":
                fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
                limit = math.sqrt(6.0 / (fan_in + fan_out))
                p.uniform_(-limit, limit, generator=self.gen)
            elif init == "ones":
                p.fill_(1.0)
            p.requires_grad_(True)
            p.grad = self.grad[off:off + n].view(shape)  # autograd accumulates in place here
            self.params[name] = p
            off += -(-n // 4) * 4

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.params[name]

    def zero_grad(self) -> None:
        self.grad.zero_()

    def adam_step(self, hp) -> None:
        """`hp`: by-value hyper-parameters or an `ops.AdamCoefBuffer` (graph-captured steps)."""
        with torch.no_grad():
            if isinstance(hp, ops.AdamCoefBuffer):
                ops.adam_dense_dc(self.flat, self.m, self.v, self.grad, hp)
            else:
                ops.adam_dense(self.flat.view(-1, 1), self.m.view(-1, 1), self.v.view(-1, 1), hp,
                               grows=self.grad)


class _blas:
    """Pin the BLAS backend for one GEMM.  Measured on MI355X at B=16384, 12928x128 fp32: the
    forward `x @ W` and the `gz @ W^T` data gradient run 15-25 % faster on rocBLAS' Tensile kernels,
    the `x^T @ gz` weight gradient 2x faster on hipBLASLt (scripts/blas_pref_bench.py)."""

    def __init__(self, lib: str):
        self.lib = lib

    def __enter__(self):
        self.prev = torch.backends.cuda.preferred_blas_library()
        torch.backends.cuda.preferred_blas_library(self.lib)

    def __exit__(self, *exc):
        torch.backends.cuda.preferred_blas_library(self.prev)


def weight_grad_slabs(B: int, n_in: int, n_out: int) -> int:
    """Number of batch slabs for `weight_grad` (1 = one GEMM).  Measured rule (B=16384): outputs up to
    128x128 -> 64 slabs.  Extrapolated rule for global batches of >= 32768 rows (field-parallel first
    layer: 1600..6464 x 128 outputs, 12..50 tiles of 128x128): enough slabs for >= 256 tiles.
    Larger outputs fill the chip on their own."""
    if B >= 4096 and B % 64 == 0 and n_in * n_out <= 128 * 128:
        return 64
    tiles = -(-n_in // 128) * -(-n_out // 128)
    if B >= 32768 and tiles < 64:
        S = 1
        while S * tiles < 256 and S < 64:
            S *= 2
        if B % S == 0:
            return S
    return 1


def weight_grad(x: torch.Tensor, gz: torch.Tensor) -> torch.Tensor:
    """x^T @ gz for a [B, in] x [B, out] pair with B >> in*out.  The output has too few tiles to fill
    256 CUs and the reduction runs over the whole batch, so hipBLASLt's pick crawls (measured at
    B=16384: 129 us for 128x64, 25 us split): reduce in S independent slabs (bmm) and add the slabs
    in a fixed order."""
    B = x.shape[0]
    S = weight_grad_slabs(B, x.shape[1], gz.shape[1])
    if S > 1:
        return torch.bmm(x.view(S, B // S, -1).transpose(1, 2), gz.view(S, B // S, -1)).sum(0)
    with _blas("cublaslt"):
        return x.t() @ gz


class _DenseFn(torch.autograd.Function):
    """addmm with the weight gradient computed by `weight_grad`."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.addmm(b, x, W)

    @staticmethod
    def backward(ctx, gz):
        x, W = ctx.saved_tensors
        gz = gz.contiguous()
        gx = gz @ W.t() if ctx.needs_input_grad[0] else None
        return gx, weight_grad(x.contiguous(), gz), gz.sum(0)


class TFDense:
    """tf.layers.dense / tf.keras.layers.Dense: glorot_uniform kernel, zero bias (dense.py:52-80)."""

    def __init__(self, P: DenseParams, name: str, n_in: int, units: int):
        self.P = P
        self.w = P.add(f"{name}/kernel", (n_in, units), "glorot_uniform")
        self.b = P.add(f"{name}/bias", (units,), "zeros")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled() and not torch.is_autocast_enabled():
            return _DenseFn.apply(x, self.P[self.w], self.P[self.b])
        return torch.addmm(self.P[self.b], x, self.P[self.w])


class _SyncBatchNorm(torch.autograd.Function):
    """Batch-statistics BatchNorm over the GLOBAL batch of a data-parallel step (one process per GPU, equal local batches):
    `avg(t)` all-reduces a small tensor in place and divides by the world size.  Forward averages (E x, E x^2), backward
    averages (E gy, E gy * xhat): the N-rank step is the 1-rank step on the concatenated batch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, avg):
        mean = avg(x.mean(0))                                   # two passes, like tf.nn.moments
        xc = x - mean
        var = avg((xc * xc).mean(0))
        inv = torch.rsqrt(var + eps)
        xhat = xc * inv
        ctx.save_for_backward(xhat, gamma, inv)
        ctx.avg = avg
        ctx.mark_non_differentiable(mean, var)
        return xhat * gamma + beta, mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        xhat, gamma, inv = ctx.saved_tensors
        dbeta, dgamma = gy.sum(0), (gy * xhat).sum(0)            # local sums: the caller's gradient all-reduce adds the ranks
        st = torch.stack([gy.mean(0), (gy * xhat).mean(0)])
        ctx.avg(st)
        gx = gamma * inv * (gy - st[0] - xhat * st[1])
        return gx, dgamma, dbeta, None, None


class TFBatchNorm:
    """tf.layers.batch_normalization(training=is_training): momentum 0.99, epsilon 1e-3, batch
    statistics (biased variance) in training, moving averages otherwise (dense.py:31-41).
    `sync` (a callable averaging a tensor over the ranks in place) makes the batch statistics those of the global batch."""

    sync = None

    def __init__(self, P: DenseParams, name: str, n: int, momentum: float = 0.99, eps: float = 1e-3):
        self.P = P
        self.gamma = P.add(f"{name}/gamma", (n,), "ones")
        self.beta = P.add(f"{name}/beta", (n,), "zeros")
        self.moving_mean = torch.zeros(n, dtype=torch.float32, device=P.device)
        self.moving_var = torch.ones(n, dtype=torch.float32, device=P.device)
        self.momentum, self.eps = momentum, eps

    def __call__(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        g, b = self.P[self.gamma], self.P[self.beta]
        if training and self.sync is not None:
            out, mean, var = _SyncBatchNorm.apply(x, g, b, self.eps, self.sync)
            with torch.no_grad():
                self.moving_mean.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
                self.moving_var.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
            return out
        if training:
            # one fused statistics + normalise kernel pair (and a fused backward) instead of ~30
            # elementwise launches; biased batch variance, like tf.nn.moments
            out, mean, invstd = torch.native_batch_norm(x, g, b, None, None, True, 0.0, self.eps)
            with torch.no_grad():  # UPDATE_OPS (training/tf_trainer.py:122-123)
                var = invstd.pow(-2).sub_(self.eps)
                self.moving_mean.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
                self.moving_var.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
            return out
        mean, var = self.moving_mean, self.moving_var
        return (x - mean) * (g * torch.rsqrt(var + self.eps)) + b


class _FoldedBNDense(torch.autograd.Function):
    """z = Dense(BN_train(x)) without ever forming BN(x) (x is the [B, F*K] embedding block —
    the one large activation of the model).  Forward folds the batch statistics into the kernel:
        z = x @ (diag(gamma/sigma) W) + b + (beta - mu*gamma/sigma) @ W
    Backward returns for x only the GEMM part  G = gz @ W'^T ; the remaining per-feature affine
    terms of the BatchNorm backward,  dx = G - a - c * x , are handed to the caller through `side`
    (``bn_a``, ``bn_c``) and applied inside ``lr_fm_embed_bwd_adam_f32`` where x = table[row]
    is in registers anyway.  All three large GEMMs read the raw x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, W, b, mean, inv, side):
        s = gamma * inv
        Wp = W * s[:, None]
        bp = b + (beta - mean * s) @ W
        ctx.save_for_backward(x, gamma, beta, W, Wp, mean, inv)
        ctx.side = side
        with _blas("cublas"):
            return torch.addmm(bp, x, Wp)

    @staticmethod
    def backward(ctx, gz):
        x, gamma, beta, W, Wp, mean, inv = ctx.saved_tensors
        B = x.shape[0]
        gz = gz.contiguous()
        sgz = gz.sum(0)
        XhG = (weight_grad(x, gz) - mean[:, None] * sgz[None, :]) * inv[:, None]   # x_hat^T gz
        dW = gamma[:, None] * XhG + beta[:, None] * sgz[None, :]
        dgamma = (XhG * W).sum(1)
        dbeta = W @ sgz
        s = gamma * inv
        c = s * inv * (dgamma / B)
        a = s * (dbeta / B) - c * mean
        avg = ctx.side.get("sync")
        if avg is not None:         # global-batch BatchNorm: a, c are linear in (dgamma, dbeta): average them over the ranks
            ac = torch.stack([a, c])
            avg(ac)
            a, c = ac[0], ac[1]
        ctx.side["bn_a"], ctx.side["bn_c"] = a.contiguous(), c.contiguous()
        with _blas("cublas"):
            G = torch.mm(gz, Wp.t())
        return G, dgamma, dbeta, dW, sgz, None, None, None


class DenseStack:
    """dense_nn(net, hidden_units, use_bn, bn_after_activation=True, dropout) — dense.py:12-49:
    optional input BN; Dense -> act -> BN -> dropout per layer; the LAST layer has no
    activation / BN / dropout."""

    def __init__(self, P: DenseParams, name: str, n_in: int, hidden_units: Sequence[int],
                 use_bn: bool = True, dropout_rate: float = 0.0, activation=F.relu):
        self.use_bn, self.dropout_rate, self.act = use_bn, dropout_rate or 0.0, activation
        self.bn_in = TFBatchNorm(P, f"{name}/bn_in", n_in) if use_bn else None
        self.layers: List[TFDense] = []
        self.bns: List[Optional[TFBatchNorm]] = []
        d = n_in
        for i, units in enumerate(hidden_units, start=1):
            self.layers.append(TFDense(P, f"{name}/{name}_layer{i}", d, units))
            last = i == len(hidden_units)
            self.bns.append(TFBatchNorm(P, f"{name}/bn{i}", units) if (use_bn and not last) else None)
            d = units
        self.n_out = d

    def set_sync(self, avg) -> None:
        """Global-batch BatchNorm for data-parallel replicas: `avg(t)` averages a tensor over the ranks in place."""
        for bn in [self.bn_in] + list(self.bns):
            if bn is not None:
                bn.sync = avg

    def _first_folded(self, x: torch.Tensor, training: bool, side: dict, stats=None) -> torch.Tensor:
        bn, layer, P = self.bn_in, self.layers[0], self.bn_in.P
        if training:
            with torch.no_grad():
                if stats is not None:     # batch statistics supplied by the caller (lr_fm_field_stats_f32)
                    mean, var = stats
                elif bn.sync is not None:  # statistics of the GLOBAL batch (one process per GPU)
                    mean = bn.sync(x.mean(0))
                    var = bn.sync(((x - mean) ** 2).mean(0))
                    side["sync"] = bn.sync
                else:
                    var, mean = torch.var_mean(x, dim=0, unbiased=False)
                bn.moving_mean.mul_(bn.momentum).add_(mean, alpha=1 - bn.momentum)
                bn.moving_var.mul_(bn.momentum).add_(var, alpha=1 - bn.momentum)
                inv = torch.rsqrt(var + bn.eps)
            return _FoldedBNDense.apply(x, P[bn.gamma], P[bn.beta], P[layer.w], P[layer.b], mean, inv, side)
        s = P[bn.gamma] * torch.rsqrt(bn.moving_var + bn.eps)
        return torch.addmm(P[layer.b] + (P[bn.beta] - bn.moving_mean * s) @ P[layer.w], x,
                           P[layer.w] * s[:, None])

    def __call__(self, x: torch.Tensor, training: bool, side: Optional[dict] = None, stats=None) -> torch.Tensor:
        """`side` != None enables the folded input-BN + first-layer path; the caller must then
        apply ``dx -= side['bn_a'] + side['bn_c'] * x`` to the gradient it receives for x."""
        fold = side is not None and self.bn_in is not None
        if self.bn_in is not None and not fold:
            x = self.bn_in(x, training)
        x = self._first_folded(x, training, side, stats) if fold else self.layers[0](x)
        return self.tail(x, training)

    def tail(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        """Everything after the first Dense (whose output `x` is): activation / BN / dropout of layer
        1 and the remaining layers (dense.py:33-49)."""
        for i, bn in enumerate(self.bns):
            if i > 0:
                x = self.layers[i](x)
            if i != len(self.layers) - 1:
                x = self.act(x)
                if bn is not None:
                    x = bn(x, training)
                if self.dropout_rate and training:
                    x = F.dropout(x, self.dropout_rate, training=True)
        return x

    def fused_first(self, io: "FusedL1IO", training: bool, stats=None) -> torch.Tensor:
        """First Dense (with the input BatchNorm folded in) computed straight from the embedding
        tables by `lr_deepfm_l1_fwd_f32`: deep_embed is never formed.  `stats` = (mean, biased var) of
        the virtual [B, F*K] block in training (from `lr_fm_field_stats_f32`)."""
        P, layer = self.layers[0].P, self.layers[0]
        bn = self.bn_in
        if bn is None:
            return _FusedL1.apply(None, None, P[layer.w], P[layer.b], None, None, io)
        if training:
            with torch.no_grad():
                mean, var = stats
                bn.moving_mean.mul_(bn.momentum).add_(mean, alpha=1 - bn.momentum)
                bn.moving_var.mul_(bn.momentum).add_(var, alpha=1 - bn.momentum)
                inv = torch.rsqrt(var + bn.eps)
        else:
            mean, inv = bn.moving_mean, torch.rsqrt(bn.moving_var + bn.eps)
        return _FusedL1.apply(P[bn.gamma], P[bn.beta], P[layer.w], P[layer.b], mean, inv, io)


class FusedL1IO:
    """Inputs / by-products of the fused lookup + first layer (one per step)."""

    def __init__(self, table, lin, idx, idxT, F, K, pack_bufs=None, wgrad_buf=None, fwd_bufs=None):
        self.table, self.lin, self.idx, self.idxT, self.F, self.K = table, lin, idx, idxT, F, K
        self.pack_bufs, self.wgrad_buf, self.fwd_bufs = pack_bufs, wgrad_buf, fwd_bufs
        # arithmetic of the layer's contractions (ops.L1_ARITH): carried by the packed-kernel buffers a net allocated
        self.arith = None if pack_bufs is None else ("split_bf16" if pack_bufs[0].dtype == torch.uint8 else "f32_chain")
        self.pair = self.fsum = self.lin_out = self.WpB = self.gz = None
        self.bn_a = self.bn_c = None


def fused_l1_forward(gamma, beta, W, b, mean, inv, io: "FusedL1IO") -> torch.Tensor:
    """z1 = gather(table, idx) @ Wp + bp with Wp = diag(gamma * inv) W, bp = b + (beta - mean * gamma * inv) @ W
    (plain W, b without BatchNorm); by-products (pair, fsum, lin_out, packed kernel) are left in `io`."""
    if gamma is not None:
        s = gamma * inv
        bp = b + (beta - mean * s) @ W
        WpA, WpB = ops.deepfm_l1_pack(W.contiguous(), io.F, io.K, out=io.pack_bufs, scale=s.contiguous())
    else:
        bp = b
        WpA, WpB = ops.deepfm_l1_pack(W.contiguous(), io.F, io.K, out=io.pack_bufs)
    z1, io.pair, io.fsum, io.lin_out = ops.deepfm_l1_fwd(io.table, io.idx, WpA, bp.contiguous(), W.shape[1], lin=io.lin)
    io.WpB = WpB
    return z1


def fused_l1_backward(gamma, beta, W, mean, inv, io: "FusedL1IO", gz: torch.Tensor, sgz: Optional[torch.Tensor] = None):
    """(dgamma, dbeta, dW, db) from gz = d loss / d z1: the weight-side gradients come from
    `lr_deepfm_l1_wgrad_f32` (gather^T @ gz) — same algebra as `_FoldedBNDense.backward`; the row-side
    gradient is NOT formed: `gz` and the BatchNorm remainder terms (``bn_a``, ``bn_c``: dx = G - a - c * x)
    are left in `io` for `lr_deepfm_l1_dgrad_f32` + `lr_fm_rows_adam_f32`."""
    gz = gz.contiguous()
    io.gz = gz
    B = gz.shape[0]
    part = ops.deepfm_l1_wgrad(io.table, io.idxT, gz, out=io.wgrad_buf, arith=io.arith)
    dWraw = part[0] if part.shape[0] == 1 else part.sum(0)          # gather^T @ gz, fixed order
    if sgz is None:
        sgz = gz.sum(0)
    if gamma is None:
        return None, None, dWraw, sgz
    XhG = (dWraw - mean[:, None] * sgz[None, :]) * inv[:, None]     # x_hat^T gz
    dW = gamma[:, None] * XhG + beta[:, None] * sgz[None, :]
    dgamma = (XhG * W).sum(1)
    dbeta = W @ sgz
    s = gamma * inv
    c = s * inv * (dgamma / B)
    a = s * (dbeta / B) - c * mean
    io.bn_a, io.bn_c = a.contiguous(), c.contiguous()
    return dgamma, dbeta, dW, sgz


class FoldedL1Kernels:
    """`fused_l1_forward` / `fused_l1_backward` with the BatchNorm-fold algebra on the device kernels of
    csrc/deepfm_fold.hip (no autograd, persistent buffers: the whole step is capturable in a hipGraph).
    `forward` starts from the per-field partial sums of `lr_fm_field_stats_f32`; `backward` writes the
    gradients straight into the flat gradient buffer of `DenseParams`."""

    STAT_CHUNKS = 8

    def __init__(self, P, bn, layer, F: int, K: int, device, stat_chunks: Optional[int] = None):
        self.P, self.bn, self.layer, self.F, self.K, self.device = P, bn, layer, int(F), int(K), device
        n, H1 = self.F * self.K, P[layer.w].shape[1]
        f32 = dict(dtype=torch.float32, device=device)
        self.H1 = H1
        if stat_chunks is not None:
            self.STAT_CHUNKS = int(stat_chunks)
        self.stat_partial = torch.empty((self.F, self.STAT_CHUNKS, 2, self.K), **f32)
        self.mean, self.inv, self.s, self.t = (torch.empty(n, **f32) for _ in range(4))
        self.n_slabs = _lib.load().lr_deepfm_l1_fold_bias_slabs(n)
        self.bias_partial = torch.empty((self.n_slabs, H1), **f32)
        self.bp = torch.empty(H1, **f32)
        self.bn_ac = torch.empty(2 * n, **f32)                 # (one buffer: one all-reduce under `sync`)
        self.bn_a, self.bn_c = self.bn_ac[:n], self.bn_ac[n:]

    @staticmethod
    def supported(H1: int) -> bool:
        return H1 in (64, 128, 256)

    def forward(self, io: "FusedL1IO", seg, field_row_start, B: int, cache_slots: Optional[torch.Tensor] = None,
                sync=None) -> torch.Tensor:
        """`sync` (data-parallel replicas with equal local batches: a callable averaging a tensor over the ranks in place):
        the per-field partial sums are averaged before they are finalised with the LOCAL batch size, i.e. the statistics
        are those of the global batch.
        `cache_slots` (row-sharded tables): `io.table` is the step's row cache, `io.idx` = `cache_slots` viewed
        [B, F] and `seg` the per-field runs of the GLOBAL ids; the statistics then read a run's row through the
        position -> cache-row map."""
        P, bn, l0, st = self.P, self.bn, self.layer, ops._stream()
        W, b = P[l0.w], P[l0.b]
        n = self.F * self.K
        if bn is not None:
            if seg is None:         # a materialised block: statistics straight from the rows (csrc/dense_block.hip)
                ops.table_colstats(io.table, io.idx, self.STAT_CHUNKS, out=self.stat_partial)
            elif cache_slots is None:
                ops._call("lr_fm_field_stats_f32", ops._ptr(io.table), self.K, ops._ptr(seg.rows), ops._ptr(seg.start),
                          ops._ptr(seg.n_seg), ops._ptr(field_row_start), self.F, self.STAT_CHUNKS, ops._ptr(self.stat_partial), st)
            else:
                ops._call("lr_fm_field_stats_slots_f32", ops._ptr(io.table), self.K, ops._ptr(seg.rows), ops._ptr(seg.start),
                          ops._ptr(seg.n_seg), ops._ptr(field_row_start), self.F, self.STAT_CHUNKS, ops._ptr(self.stat_partial),
                          ops._ptr(seg.pos), ops._ptr(cache_slots), st)
            if sync is not None:
                sync(self.stat_partial)
            if os.environ.get("LIBRECO_FOLD_CHAIN", "merged") == "merged" and self.H1 <= 256 and 256 % self.H1 == 0:
                # two launches (round 6): statistics + bias partials of every 64-row slab, then the weight pack with the
                # partials' reduction as extra workgroups — the four-launch chain below, bit for bit
                ops._call("lr_deepfm_l1_fold_stats_bias_f32", ops._ptr(self.stat_partial), self.F, self.STAT_CHUNKS, self.K, B,
                          float(bn.eps), float(bn.momentum), ops._ptr(P[bn.gamma]), ops._ptr(P[bn.beta]),
                          ops._ptr(bn.moving_mean), ops._ptr(bn.moving_var), ops._ptr(self.mean), ops._ptr(self.inv),
                          ops._ptr(self.s), ops._ptr(self.t), ops._ptr(W), ops._ptr(b), self.H1, ops._ptr(self.bias_partial), st)
                WpA, WpB = ops.deepfm_l1_pack(W, self.F, self.K, out=io.pack_bufs, scale=self.s,
                                              reduce=(self.bias_partial, self.bp))
            else:
                ops._call("lr_deepfm_l1_fold_stats_f32", ops._ptr(self.stat_partial), self.F, self.STAT_CHUNKS, self.K, B,
                          float(bn.eps), float(bn.momentum), ops._ptr(P[bn.gamma]), ops._ptr(P[bn.beta]),
                          ops._ptr(bn.moving_mean), ops._ptr(bn.moving_var), ops._ptr(self.mean), ops._ptr(self.inv),
                          ops._ptr(self.s), ops._ptr(self.t), st)
                WpA, WpB = ops.deepfm_l1_pack(W, self.F, self.K, out=io.pack_bufs, scale=self.s)
                ops._call("lr_deepfm_l1_fold_bias_f32", ops._ptr(self.t), ops._ptr(W), ops._ptr(b), n, self.H1,
                          ops._ptr(self.bias_partial), st)
                ops._call("lr_reduce_partials_f32", ops._ptr(self.bias_partial), self.n_slabs, self.H1, self.H1,
                          ops._ptr(self.bp), st)
            bias = self.bp
        else:
            WpA, WpB = ops.deepfm_l1_pack(W, self.F, self.K, out=io.pack_bufs)
            bias = b
        z1, io.pair, io.fsum, io.lin_out = ops.deepfm_l1_fwd(io.table, io.idx, WpA, bias, self.H1, lin=io.lin,
                                                             out=getattr(io, "fwd_bufs", None))
        io.WpB = WpB
        return z1

    def backward(self, io: "FusedL1IO", gz: torch.Tensor, sgz: torch.Tensor, sync=None) -> None:
        """`sync`: the BatchNorm-backward remainder coefficients (linear in the batch sums d gamma / d beta) are averaged
        over the ranks: those of the global batch."""
        P, bn, l0 = self.P, self.bn, self.layer
        io.gz = gz
        B = gz.shape[0]
        part = ops.deepfm_l1_wgrad(io.table, io.idxT, gz, out=io.wgrad_buf, arith=io.arith)
        W = P[l0.w]
        has = bn is not None
        ops._call("lr_deepfm_l1_fold_bwd_f32", ops._ptr(part), part.shape[0], self.F * self.K, self.H1, B, ops._ptr(sgz),
                  ops._ptr(W), ops._ptr(P[bn.gamma]) if has else 0, ops._ptr(P[bn.beta]) if has else 0,
                  ops._ptr(self.mean) if has else 0, ops._ptr(self.inv) if has else 0, ops._ptr(W.grad),
                  ops._ptr(P[bn.gamma].grad) if has else 0, ops._ptr(P[bn.beta].grad) if has else 0, ops._ptr(P[l0.b].grad),
                  ops._ptr(self.bn_a) if has else 0, ops._ptr(self.bn_c) if has else 0, ops._stream())
        if has and sync is not None:
            sync(self.bn_ac)
        io.bn_a, io.bn_c = (self.bn_a, self.bn_c) if has else (None, None)


class _FusedL1(torch.autograd.Function):
    """Autograd wrapper of `fused_l1_forward` / `fused_l1_backward` (torch tail)."""

    @staticmethod
    def forward(ctx, gamma, beta, W, b, mean, inv, io):
        ctx.io = io
        ctx.save_for_backward(gamma, beta, W, mean, inv)
        return fused_l1_forward(gamma, beta, W, b, mean, inv, io)

    @staticmethod
    def backward(ctx, gz):
        gamma, beta, W, mean, inv = ctx.saved_tensors
        dgamma, dbeta, dW, db = fused_l1_backward(gamma, beta, W, mean, inv, ctx.io, gz)
        return dgamma, dbeta, dW, db, None, None, None


class BlockFirstLayer:
    """First Dense layer of `dense_nn` (layers/dense.py:12-49 of the reference: input BatchNorm -> Dense) over a
    MATERIALISED input stored plane by plane — `xbuf` [P, B, K]: plane p = field p of every sample, i.e. the reference's
    concat([field 0, field 1, ...]) row of sample b is (xbuf[0, b], xbuf[1, b], ...).  The block is handed to the fused
    lookup + first-layer kernels as a table of P * B rows with the identity id map idx[b, p] = p * B + b, so forward
    (MFMA), weight gradient, row gradient and the BatchNorm-fold algebra are the kernels of the DeepFM step; the
    BatchNorm-backward remainder (dx = G - a - c * x) is one elementwise pass.  Persistent buffers, no autograd, no
    host reads: capturable in a hipGraph.

    `forward(xbuf)` -> z1 [B, H1];  `backward(gz1, sgz1, gbuf)` writes d loss / d xbuf into `gbuf` (first P * B rows, same
    planar layout; `gbuf` must hold at least P * B + 1 rows) and the parameter gradients into `DenseParams`' flat
    gradient buffer."""

    def __init__(self, P, bn, layer, planes: int, K: int, B: int, device, layout: str = "planar"):
        """`layout="rowmajor"`: the block is the reference's own [B, planes * K] matrix (sample-major) instead of planes
        of B rows — e.g. a materialised deep_embed whose K-wide field blocks are re-cut into `planes` blocks of a width the
        MFMA kernels are compiled for."""
        self.Pn, self.K, self.B, self.device = int(planes), int(K), int(B), device
        self.rowmajor = layout == "rowmajor"
        H1 = P[layer.w].shape[1]
        self.H1 = H1
        chunks = max(1, min(64, -(-512 // self.Pn), -(-B // 64)))
        self.fold = FoldedL1Kernels(P, bn, layer, self.Pn, self.K, device, stat_chunks=chunks)
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        if self.rowmajor:
            self.idx = torch.arange(self.Pn * B, **i32).view(B, self.Pn).contiguous()      # [B, P]: row of (sample, block)
            self.idxT = self.idx.t().contiguous()
        else:
            self.idxT = torch.arange(self.Pn * B, **i32).view(self.Pn, B).contiguous()     # [P, B]: row of (plane, sample)
            self.idx = self.idxT.t().contiguous()                                          # [B, P]
        n = self.Pn * self.K
        self.pack = ops.deepfm_l1_pack_bufs(self.Pn, self.K, H1, device)
        arith = "split_bf16" if self.pack[0].dtype == torch.uint8 else "f32_chain"
        nch = ops.deepfm_l1_wgrad_chunks(B, self.Pn, self.K, H1, arith)
        self.wgrad = torch.empty((nch, n, H1), **f32)
        self.fwd_bufs = (torch.empty((B, H1), **f32), torch.empty((B, self.K), **f32), torch.empty((B, self.K), **f32))
        self.io = None

    @staticmethod
    def supported(K: int, H1: int) -> bool:
        return bool(ops.deepfm_l1_supported(K, H1)) and FoldedL1Kernels.supported(H1)

    def forward(self, xbuf: torch.Tensor) -> torch.Tensor:
        table = xbuf.reshape(self.Pn * self.B, self.K)
        self.io = FusedL1IO(table, None, self.idx, self.idxT, self.Pn, self.K, pack_bufs=self.pack, wgrad_buf=self.wgrad,
                            fwd_bufs=self.fwd_bufs)
        return self.fold.forward(self.io, None, None, self.B)

    def backward(self, gz1: torch.Tensor, sgz1: torch.Tensor, gbuf: torch.Tensor) -> None:
        io = self.io
        self.fold.backward(io, gz1, sgz1)
        n_rows = self.Pn * self.B
        ops.deepfm_l1_dgrad(io.gz, io.WpB, self.K, self.Pn, self.idxT, out=gbuf)
        if io.bn_a is not None:
            ops.bn_remainder_(gbuf[:n_rows], io.table, io.bn_a, io.bn_c, self.B, period=self.Pn if self.rowmajor else 0)
