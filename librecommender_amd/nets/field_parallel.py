"""DeepFM over several GPUs, partitioned by FIELD (the alternative to the row-sharded
exchange of `ShardedDeepFMNet` when the model has many fields).

Why: with row-sharded tables every step moves the batch's distinct embedding rows and their
gradients across xGMI — on the benchmark workload ~2.1 M rows x 260 B each way per GPU, about 1 GB
per step, i.e. more time on the point-to-point links than the whole single-GPU step.  Partitioning
the *fields* instead keeps every table access local and moves only [batch, width] activations:

  rank r owns the fields [f_lo, f_hi) — their table rows, Adam moments, input-BatchNorm
  parameters, the matching rows of the first MLP kernel and of the linear-term kernel.
  1. all-to-all of the id columns: every rank receives the ids of ITS fields for the GLOBAL batch
     (B_global x F_r int32).
  2. local `lr_fm_embed_fwd_f32` over B_global x F_r positions: e_r, the partial field sum
     fsum_r and partial pairwise term pair_r.  Global pairwise term:
         pair = sum_r (pair_r - fsum_r^2 / 2) + (sum_r fsum_r)^2 / 2
     so one all-reduce of [B_global, 2K+1] (fsum_r | pair_r - fsum_r^2/2 | partial linear term).
  3. first MLP layer tensor-parallel: z1_r = Dense_r(BN_r(e_r)) is a partial sum over the rank's
     features (BatchNorm is per feature, every rank sees the global batch of its features: exact
     global-batch statistics); reduce-scatter -> each rank keeps z1 of its own B samples.
  4. rest of the MLP, output layer and loss data-parallel on the own samples, hidden BatchNorm with
     all-reduced statistics; backward retraces the collectives (all-gather <-> reduce-scatter).
  5. local `lr_fm_embed_bwd_adam_f32` — the same fused backward + row-wise Adam as on one GPU,
     no gradient exchange for tables, first-layer kernel or linear kernel; one small all-reduce
     for the replicated dense parameters.

Per step and rank ~ (2 x 128 + 3K + 2) x 4 bytes per global sample through ring collectives instead
of the rows themselves.  The N-rank step equals the 1-rank step on the concatenated batch (same
global-batch BatchNorm, same lazy Adam); tests/test_field_parallel_cpu.py checks exactly that under
gloo with the oracle kernels, and the first step against the reference-graph oracle.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from ..layers.dense import DenseParams, TFDense, _FoldedBNDense
from ..parallel import _a2a_single, _all_gather_into, _reduce_scatter_sum, allreduce_sum_
from .fm_nets import _FieldNet


class _OwnRowsOfSum(torch.autograd.Function):
    """This rank's B rows of  S = sum over ranks of S_r  ([W*B, C]).  The all-reduce itself is started
    by `FieldParallelDeepFMNet._start_sum` (asynchronously under RCCL, so that it runs under the
    first-layer GEMM) and only awaited here.  The gradient of a row lives on the rank that owns the
    sample: backward only hands the own-row gradient to the net, which all-gathers it later
    (`_start_grad_gather`, in flight during the first-layer backward GEMMs) and applies the three
    small products to fsum / q / the linear kernel by hand (`_finish_stat_grads`)."""

    @staticmethod
    def forward(ctx, anchor, net):
        ctx.net = net
        full = net._finish_sum()
        B = full.shape[0] // net.world
        return full[net.rank * B:(net.rank + 1) * B].clone()

    @staticmethod
    def backward(ctx, g_own):
        ctx.net._dstats_own = g_own.contiguous()
        return None, None


class _ReduceScatterRows(torch.autograd.Function):
    """[W*B, C] partial sums -> this rank's [B, C] rows of the total; backward all-gathers."""

    @staticmethod
    def forward(ctx, part, net):
        ctx.net = net
        out = torch.empty((part.shape[0] // net.world, part.shape[1]), dtype=part.dtype, device=part.device)
        _reduce_scatter_sum(out, part.contiguous(), net.group)
        return out

    @staticmethod
    def backward(ctx, g_own):
        net = ctx.net
        g = torch.empty((g_own.shape[0] * net.world, g_own.shape[1]), dtype=g_own.dtype, device=g_own.device)
        _all_gather_into(g, g_own.contiguous(), group=net.group)
        net._start_grad_gather()          # queued behind this gather, overlaps the GEMMs that consume `g`
        return g, None


class _GlobalBatchNorm(torch.autograd.Function):
    """Training-mode BatchNorm over the GLOBAL batch (biased variance, `tf.nn.moments`): the
    per-feature sums are all-reduced in the forward, the two reduction terms of the input gradient
    in the backward.  gamma / beta receive their LOCAL gradient (the dense-parameter all-reduce
    adds the ranks up)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, net):
        n = x.shape[0] * net.world
        stats = torch.stack([x.double().sum(0), (x.double() * x.double()).sum(0)])
        allreduce_sum_(stats, net.group)
        mean = stats[0] / n
        var = (stats[1] / n - mean * mean).clamp_(min=0.0)
        inv = torch.rsqrt(var + eps)
        xhat = ((x.double() - mean) * inv).float()
        ctx.save_for_backward(xhat, gamma, inv.float())
        ctx.net, ctx.n = net, n
        mean_f, var_f = mean.float(), var.float()
        ctx.mark_non_differentiable(mean_f, var_f)
        return xhat * gamma + beta, mean_f, var_f

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        xhat, gamma, inv = ctx.saved_tensors
        dbeta, dgamma = g.sum(0), (g * xhat).sum(0)
        red = torch.stack([dbeta, dgamma]).double()
        allreduce_sum_(red, ctx.net.group)
        red = (red / ctx.n).float()
        dx = gamma * inv * (g - red[0] - xhat * red[1])
        return dx, dgamma, dbeta, None, None


class FieldParallelDeepFMNet:
    """`idx` of `train_step / forward` holds GLOBAL table rows, [B, F], this rank's samples; every
    rank must pass the same B.  `field_row_start[f] .. field_row_start[f+1]` are the rows of field f."""

    momentum, bn_eps = 0.99, 1e-3                       # tf.layers.batch_normalization defaults

    def __init__(self, field_row_start, embed_size=16, hidden_units=(128, 64, 32), use_bn=True, lr=1e-3,
                 epsilon=1e-5, seed=42, device=None, kern=None, group=None):
        from ..parallel import HipKernels

        self.kern = kern or HipKernels()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device or torch.device("cuda")
        frs = np.asarray(field_row_start, dtype=np.int64)
        self.F, self.K = len(frs) - 1, int(embed_size)
        if self.F < self.world:
            raise ValueError(f"{self.F} fields cannot be partitioned over {self.world} ranks; use ShardedDeepFMNet")
        self.bounds = [(self.F * r) // self.world for r in range(self.world + 1)]      # contiguous field blocks
        self.f_lo, self.f_hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.Fr = self.f_hi - self.f_lo
        self.row_base, self.V_local = int(frs[self.f_lo]), int(frs[self.f_hi] - frs[self.f_lo])
        self.hidden, self.use_bn = list(hidden_units), bool(use_bn)
        self.lr, self.epsilon, self.step = lr, epsilon, 0
        dev, K, H1 = self.device, self.K, self.hidden[0]
        gen = torch.Generator(device=dev)

        def uniform(shape, limit, stream):               # per-field streams: the same global model for any W
            gen.manual_seed((int(seed) * 1_000_003 + stream) & ((1 << 62) - 1))
            return (torch.rand(shape, generator=gen, device=dev) * 2.0 - 1.0) * limit

        # ---- this rank's tables -----------------------------------------------------------------
        self.embed = torch.empty((self.V_local, K), dtype=torch.float32, device=dev)
        self.lin = torch.empty((self.V_local, 1), dtype=torch.float32, device=dev)
        for f in range(self.f_lo, self.f_hi):
            a, b = int(frs[f]) - self.row_base, int(frs[f + 1]) - self.row_base
            self.embed[a:b] = uniform((b - a, K), 0.01, 4 * f)
            self.lin[a:b] = uniform((b - a, 1), 0.01, 4 * f + 1)
        self.m, self.v = torch.zeros_like(self.embed), torch.zeros_like(self.embed)
        self.lin_m, self.lin_v = torch.zeros_like(self.lin), torch.zeros_like(self.lin)
        # ---- field-sharded dense parameters (no gradient exchange) ------------------------------
        PL = self.PL = DenseParams(dev, seed)
        PL.add("linear/kernel", (self.Fr, 1), "zeros")
        PL.add("mlp/mlp_layer1/kernel", (self.Fr * K, H1), "zeros")
        if self.use_bn:
            PL.add("mlp/bn_in/gamma", (self.Fr * K,), "ones")
            PL.add("mlp/bn_in/beta", (self.Fr * K,), "zeros")
        PL.finalize()
        with torch.no_grad():                             # glorot limits of the FULL layers
            for j, f in enumerate(range(self.f_lo, self.f_hi)):
                PL["linear/kernel"][j] = uniform((1,), math.sqrt(6.0 / (self.F + 1)), 4 * f + 2)
                PL["mlp/mlp_layer1/kernel"][j * K:(j + 1) * K] = uniform((K, H1), math.sqrt(6.0 / (self.F * K + H1)), 4 * f + 3)
        self.bn_in_mean = torch.zeros(self.Fr * K, dtype=torch.float32, device=dev)
        self.bn_in_var = torch.ones(self.Fr * K, dtype=torch.float32, device=dev)
        # ---- replicated dense parameters (same seed on every rank) --------------------------------
        P = self.P = DenseParams(dev, seed)
        P.add("linear/bias", (1,), "zeros")
        P.add("mlp/mlp_layer1/bias", (H1,), "zeros")
        self.layers, self.bns, d = [], [], H1
        for i, units in enumerate(self.hidden[1:], start=2):
            if self.use_bn:
                self.bns.append((P.add(f"mlp/bn{i - 1}/gamma", (d,), "ones"), P.add(f"mlp/bn{i - 1}/beta", (d,), "zeros"),
                                 torch.zeros(d, device=dev), torch.ones(d, device=dev)))
            self.layers.append(TFDense(P, f"mlp/mlp_layer{i}", d, units))
            d = units
        self.out = TFDense(P, "out", 1 + K + d, 1)
        P.finalize()
        self._zero_b1 = torch.zeros(H1, dtype=torch.float32, device=dev)
        self._stats_full = self._stats_work = None
        self._dstats_own = self._dstats_full = self._dstats_work = None
        self._bwd_ws = None
        self._seg_stream = None

    # ---- pieces -----------------------------------------------------------------------------------
    def _exchange_ids(self, idx):
        """[B, F] global rows of the own samples -> [W*B, F_r] local rows of the own fields."""
        B, W = idx.shape[0], self.world
        cols = idx.to(torch.int32).t().contiguous()                       # [F, B]: a rank's fields are contiguous
        send = [(self.bounds[r + 1] - self.bounds[r]) * B for r in range(W)]
        recv = torch.empty(W * self.Fr * B, dtype=torch.int32, device=idx.device)
        _a2a_single(recv, cols.view(-1), [self.Fr * B] * W, send, group=self.group)
        loc = recv.view(W, self.Fr, B).permute(0, 2, 1).reshape(W * B, self.Fr)
        return (loc - self.row_base).contiguous()

    def _hp(self):
        return self.kern.adam_hp(self.lr, self.step, self.epsilon)

    def _start_sum(self, part):
        full = part.detach().clone()
        self._stats_full, self._stats_work = full, None
        if full.is_cuda and dist.get_backend(self.group) != "gloo":
            self._stats_work = dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            allreduce_sum_(full, self.group)

    def _start_grad_gather(self):
        own = self._dstats_own
        if own is None:
            return
        full = torch.empty((own.shape[0] * self.world, own.shape[1]), dtype=own.dtype, device=own.device)
        self._dstats_full, self._dstats_work, self._dstats_own = full, None, None
        if own.is_cuda and dist.get_backend(self.group) != "gloo":
            self._dstats_work = dist.all_gather_into_tensor(full, own, group=self.group, async_op=True)
        else:
            _all_gather_into(full, own, group=self.group)

    def _finish_stat_grads(self, lin_r):
        """(gpair [Bg,K], glin [Bg,F_r]) for the fused backward; adds the linear kernel's gradient."""
        self._start_grad_gather()                      # no-op unless the reduce-scatter backward did not run
        if self._dstats_work is not None:
            self._dstats_work.wait()
            self._dstats_work = None
        d, K = self._dstats_full, self.K
        self._dstats_full = None
        Wl = self.PL["linear/kernel"]
        dlin_sum = d[:, 2 * K:]                                          # [Bg, 1]
        Wl.grad.add_(lin_r.detach().t() @ dlin_sum)                       # d(lin_r @ Wl) / d Wl
        return d[:, K:2 * K].contiguous(), (dlin_sum @ Wl.detach().t()).contiguous()

    def _finish_sum(self):
        if self._stats_work is not None:
            self._stats_work.wait()
            self._stats_work = None
        return self._stats_full

    def _hidden(self, z1, training):
        x = z1 + self.P["mlp/mlp_layer1/bias"]
        for i, layer in enumerate(self.layers):
            x = F.relu(x)
            if self.use_bn:
                g, b, mm, mv = self.bns[i]
                if training:
                    x, mean, var = _GlobalBatchNorm.apply(x, self.P[g], self.P[b], self.bn_eps, self)
                    with torch.no_grad():
                        mm.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
                        mv.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
                else:
                    x = (x - mm) * (self.P[g] * torch.rsqrt(mv + self.bn_eps)) + self.P[b]
            x = layer(x)
        return x

    def _first_layer(self, e_flat, training, side):
        PL = self.PL
        W1 = PL["mlp/mlp_layer1/kernel"]
        if not self.use_bn:
            return e_flat @ W1
        g, b = PL["mlp/bn_in/gamma"], PL["mlp/bn_in/beta"]
        if training:
            with torch.no_grad():
                var, mean = torch.var_mean(e_flat, dim=0, unbiased=False)
                self.bn_in_mean.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
                self.bn_in_var.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
                inv = torch.rsqrt(var + self.bn_eps)
            return _FoldedBNDense.apply(e_flat, g, b, W1, self._zero_b1, mean, inv, side)
        s = g * torch.rsqrt(self.bn_in_var + self.bn_eps)
        return torch.addmm((b - self.bn_in_mean * s) @ W1, e_flat, W1 * s[:, None])

    def _logits(self, e, pair_r, fsum_r, lin_r, training, side):
        Bg, K = e.shape[0], self.K
        with torch.no_grad():
            part = torch.cat([fsum_r, pair_r - 0.5 * fsum_r * fsum_r, lin_r @ self.PL["linear/kernel"]], dim=1)
        self._start_sum(part)                                 # in flight during the first-layer GEMM
        z1 = _ReduceScatterRows.apply(self._first_layer(e.view(Bg, self.Fr * K), training, side), self)
        deep = self._hidden(z1, training)
        stats = _OwnRowsOfSum.apply(z1, self)                 # `z1` only anchors the node in the graph
        fsum, q, lin_sum = stats[:, :K], stats[:, K:2 * K], stats[:, 2 * K:]
        pair = q + 0.5 * fsum.detach() * fsum.detach()        # d pair / d e_f = fsum - e_f is applied by the kernel
        concat = torch.cat([lin_sum + self.P["linear/bias"], pair, deep], dim=1)          # deepfm.py:171
        return self.out(concat).squeeze(1)

    # ---- public -----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, idx):
        loc = self._exchange_ids(idx)
        e, pair_r, fsum_r, lin_r = self.kern.fm_fwd(self.embed, self.lin, loc)
        return self._logits(e, pair_r, fsum_r, lin_r, False, None)

    def train_step(self, idx, labels, loss_type="cross_entropy"):
        self.step += 1
        Bg = idx.shape[0] * self.world
        loc = self._exchange_ids(idx)
        # the segment build (radix sort + scans, latency-bound) depends on the ids only: run it on a
        # side stream under the forward GEMMs, join before the fused backward (as in DeepFMNet)
        side_stream = None
        if loc.is_cuda:
            if self._seg_stream is None:
                self._seg_stream = torch.cuda.Stream(device=self.device)
            side_stream, cur = self._seg_stream, torch.cuda.current_stream(self.device)
            side_stream.wait_stream(cur)
            with torch.cuda.stream(side_stream):
                seg = self.kern.segments(loc.reshape(-1), self.V_local, tag="field")
        else:
            seg = self.kern.segments(loc.reshape(-1), self.V_local, tag="field")
        e, pair_r, fsum_r, lin_r = self.kern.fm_fwd(self.embed, self.lin, loc)
        e.requires_grad_(True)
        self.P.zero_grad()
        self.PL.zero_grad()
        side = {}
        logits = self._logits(e, pair_r, fsum_r, lin_r, True, side)
        loss = _FieldNet.loss_fn(logits, labels, loss_type)
        (loss / self.world).backward()                              # mean over the global batch
        with torch.no_grad():
            if side_stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(side_stream)
            hp = self._hp()
            fsum = self._stats_full[:, : self.K].contiguous()       # global field sum of every sample
            gpair, glin = self._finish_stat_grads(lin_r)
            self.kern.fm_bwd_adam(self, e.grad, gpair, fsum, Bg, self.Fr, seg, glin,
                                  side.get("bn_a"), side.get("bn_c"), hp)
            self.kern.dense_adam(self.PL.flat, self.PL.m, self.PL.v, self.PL.grad, hp)    # sharded: local gradient is complete
            allreduce_sum_(self.P.grad, self.group)
            self.kern.dense_adam(self.P.flat, self.P.m, self.P.v, self.P.grad, hp)
        return loss.detach()

    # ---- tests / export ---------------------------------------------------------------------------
    def gather_full(self):
        """(embed [V, K], lin [V, 1]) of the whole model on every rank."""
        parts = [None] * self.world
        dist.all_gather_object(parts, (self.embed.cpu(), self.lin.cpu()), group=self.group)
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

    def gather_sharded_dense(self):
        """The field-sharded dense parameters assembled in field order (name -> full tensor)."""
        parts = [None] * self.world
        dist.all_gather_object(parts, {k: p.detach().cpu() for k, p in self.PL.params.items()}, group=self.group)
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
