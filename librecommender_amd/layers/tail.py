"""DeepFM tail on the device without autograd: the layers of `dense_nn` after the first Dense
(layers/dense.py:33-49 of the reference), the output layer (algorithms/deepfm.py:158, 171-172), the
sigmoid cross-entropy loss (tfops/loss.py:14-16) and their backward — csrc/deepfm_tail.hip.

`run` consumes z1 (the first Dense layer's output), the pairwise term and the gathered linear weights (both None
with `F = K = 0`: the plain output layer of DIN / YouTubeRanking, algorithms/din.py:190-192) and
returns (loss, gl, gz1, sgz1): the loss, d loss / d logit, d loss / d z1 and its column sums.  The
gradients of every parameter it touches are written into the flat gradient buffer of `DenseParams`
(`P[name].grad` views), the BatchNorm moving averages are updated in place."""
from __future__ import annotations

import os
import weakref
from typing import List

import torch

from .. import _lib, ops

_ptr = ops._ptr


def _call(name, *args):
    ops._call(name, *args)


_LIVE: "weakref.WeakSet[DeepFMTail]" = weakref.WeakSet()


class TailBarrierError(RuntimeError):
    """The one-launch tail's grid barrier did not complete: the step's BatchNorm statistics did not see the whole batch."""


def check_all() -> None:
    """Raise if any live tail's one-launch form gave up on a grid barrier (`DeepFMTail.check`).  One device read per tail that
    ran fused: called where the loss is read back anyway (training/trainer.py once per epoch, bench.py after the timed region)."""
    for t in list(_LIVE):
        t.check()


def _multi_rank() -> bool:
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class DeepFMTail:
    TS = 64          # samples per workgroup of the tail kernels
    SYNC_WORDS, STICKY, LIMIT = 24, 18, 19      # layout of `sync_words`: include/libreco_hip.h, lr_mlp_tail3_args

    def __init__(self, P, mlp, linear, out, F: int, K: int, device: torch.device):
        self.P, self.mlp, self.linear, self.out = P, mlp, linear, out
        self.F, self.K, self.device = int(F), int(K), device
        self.widths: List[int] = [P[l.w].shape[1] for l in mlp.layers]
        self._B = 0
        self._jobs, self._jobs_dev, self._jobs_max_n = [], None, 0
        # One buffer set per batch size, kept alive: a hipGraph captured for one batch shape holds the raw addresses
        # of its set (activations, partial sums, the device-resident reduction job table) while steps of another
        # shape run in between — the last, shorter batch of every epoch.  `on_release` is called when a set has to
        # be dropped (more than MAX_SETS shapes): the owner must forget the graphs it captured.
        self._sets = {}
        self.on_release = None
        # dropout after every hidden layer's BatchNorm (layers/dense.py:44-47): keep probability and a per-step seed passed BY
        # VALUE — a captured graph would freeze it, so the owners of this tail do not capture steps of a net with dropout
        self.keep = 1.0 - float(getattr(mlp, "dropout_rate", 0.0) or 0.0)
        self.drop_seed = 0
        # the whole chain as ONE persistent launch where the shape is compiled (three layers 128 -> 64 -> 32: the reference's
        # default hidden_units) — same per-tile arithmetic, bit-identical results (csrc/deepfm_tail.hip: mlp_tail3_kernel)
        self.fused = bool(len(self.widths) == 3 and os.environ.get("LIBRECO_TAIL", "fused") != "chain"
                          and _lib.load().lr_mlp_tail3_supported(*self.widths, self.K, self.F))
        # The one launch holds a hand-rolled grid barrier: all of its workgroups must be resident at once.  The launcher sizes
        # the grid from the device's occupancy and a barrier that cannot complete ends in NaN losses + a sticky error word
        # (`check`), but a collective's kernels running beside it take CUs away in a way no one-GPU test can vet — under a
        # process group with more than one rank the 13-launch chain runs unless LIBRECO_TAIL_MULTI_RANK=fused says otherwise.
        self.fused_multi_rank = os.environ.get("LIBRECO_TAIL_MULTI_RANK", "chain") == "fused"
        self.spin_limit = 0                      # polls per barrier before giving up (0: the library's default, ~2 s)
        self._ran_fused = False
        _LIVE.add(self)

    @staticmethod
    def supported(mlp, loss_type: str = "cross_entropy") -> bool:
        import torch.nn.functional as Fn

        if loss_type != "cross_entropy" or mlp.act is not Fn.relu:      # (round 4: dropout is a counter-based mask in the kernels)
            return False
        lib = _lib.load()
        w = [mlp.layers[0].P[l.w].shape[1] for l in mlp.layers]
        if any(x % 16 or x > 256 or x < 16 for x in w):
            return False
        return all(lib.lr_mlp_tail_supported(a, b) for a, b in zip(w[:-1], w[1:]))

    MAX_SETS = 8
    _SET_ATTRS = ("nblk", "z", "gh", "stat_partial", "mean", "inv", "bn_partial", "dW_partial", "db_partial", "G",
                  "head_partial", "loss_sum", "gl", "gz1", "sgz_partial", "sgz1", "_jobs", "_jobs_dev", "_jobs_max_n", "sync_words")

    def _alloc(self, B: int) -> None:
        if self._B:                                 # park the set in use
            self._sets[self._B] = {k: getattr(self, k) for k in self._SET_ATTRS}
        if B in self._sets:
            for k, v in self._sets.pop(B).items():
                setattr(self, k, v)
            self._B = B
            return
        if len(self._sets) >= self.MAX_SETS:
            self._sets.pop(next(iter(self._sets)))  # oldest parked set
            if self.on_release is not None:
                self.on_release()
        dev, w, n = self.device, self.widths, len(self.widths)
        nblk = -(-B // self.TS)
        f32 = dict(dtype=torch.float32, device=dev)
        self.nblk = nblk
        self.z = [None] + [torch.empty((B, w[i]), **f32) for i in range(1, n)]         # z[0] is the caller's z1
        self.gh = [torch.empty((B, w[i]), **f32) for i in range(n - 1)]
        self.stat_partial = [torch.empty((nblk, 2, w[i]), **f32) for i in range(n - 1)]
        self.mean = [torch.empty(w[i], **f32) for i in range(n - 1)]
        self.inv = [torch.empty(w[i], **f32) for i in range(n - 1)]
        self.bn_partial = [torch.empty((nblk, 2, w[i]), **f32) for i in range(n - 1)]
        self.dW_partial = [torch.empty((nblk, w[i] * w[i + 1]), **f32) for i in range(n - 1)]
        self.db_partial = [torch.empty((nblk, w[i + 1]), **f32) for i in range(n - 1)]
        off = 1 if self.F > 0 else 0                        # plain form (no linear term): see lr_mlp_head_f32
        self.G = off + self.K + w[-1] + 1 + self.F + off
        self.head_partial = torch.empty((nblk, self.G + 1), **f32)
        self.loss_sum = torch.empty(1, **f32)
        self.gl = torch.empty(B, **f32)
        self.gz1 = torch.empty((B, w[0]), **f32)
        self.sgz_partial = torch.empty((nblk, w[0]), **f32)
        self.sgz1 = torch.empty(w[0], **f32)
        self.sync_words = torch.zeros(self.SYNC_WORDS, dtype=torch.int32, device=dev)   # arrival counter, error words, phase marks, poll bound
        self._B = B
        self._jobs, self._jobs_dev, self._jobs_max_n = [], None, 0     # the job table holds pointers into these buffers

    def _reduce(self, partial: torch.Tensor, offset: int, n: int, out: torch.Tensor, defer: bool = False, div: float = 0.0) -> None:
        """out[c] = sum_k partial[k][offset + c] (fixed order).  `defer`: the result is only read by the optimiser —
        the job joins the ONE multi-job launch at the end of `run` (the job table is built once per buffer set:
        every pointer is persistent)."""
        stride = partial.numel() // partial.shape[0]
        if defer:
            if self._jobs_dev is None:
                self._jobs.append((partial.data_ptr() + 4 * offset, out.data_ptr(), n, stride, partial.shape[0], float(div)))
            return
        assert div == 0.0
        _call("lr_reduce_partials_f32", partial.data_ptr() + 4 * offset, partial.shape[0], n, stride, _ptr(out), ops._stream())

    def _flush_deferred(self) -> None:
        if self._jobs_dev is None:
            import numpy as np

            assert _lib.load().lr_reduce_job_bytes() == 40
            rec = np.zeros(len(self._jobs), dtype=[("p", "<u8"), ("o", "<u8"), ("n", "<i8"), ("s", "<i8"), ("k", "<i4"), ("div", "<f4")])
            for i, (p_, o_, n_, s_, k_, d_) in enumerate(self._jobs):
                rec[i] = (p_, o_, n_, s_, k_, d_)
            self._jobs_dev = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
            self._jobs_max_n = max(j[2] for j in self._jobs)
        _call("lr_reduce_partials_multi_f32", _ptr(self._jobs_dev), len(self._jobs), self._jobs_max_n, ops._stream())

    def _bn(self, i: int):
        return self.mlp.bns[i]

    def run(self, z1: torch.Tensor, pair: torch.Tensor, lin_out: torch.Tensor, labels: torch.Tensor, sync=None, drop_seed=None):
        """`sync` (data-parallel replicas with equal local batches: a callable averaging a tensor over the ranks in place):
        every BatchNorm's partial sums — forward statistics and the backward sums — are averaged over the ranks before
        they are reduced with the LOCAL batch size: BatchNorm over the global batch, the kernels unchanged.  (The
        parameter gradients written from the averaged backward sums are in the same "local mean" units as every other
        gradient of the replica, so the caller's usual 1/W scaling + all-reduce applies to them too.)"""
        P, mlp, w, n = self.P, self.mlp, self.widths, len(self.widths)
        B = z1.shape[0]
        if B != self._B:
            self._alloc(B)
        s, nblk = ops._stream(), self.nblk
        z = self.z
        z[0] = z1
        if drop_seed is None:
            self.drop_seed = (self.drop_seed + 1) & 0x7FFFFFFF
            drop_seed = self.drop_seed
        keep = float(self.keep)
        mode = "fused" if (self.fused and sync is None and (self.fused_multi_rank or not _multi_rank())) else "chain"
        if getattr(self, "_jobs_mode", mode) != mode:       # (the two forms defer different reductions: one job table each)
            self._jobs, self._jobs_dev, self._jobs_max_n = [], None, 0
        self._jobs_mode = mode
        if mode == "fused":
            return self._run_fused(z1, pair, lin_out, labels, int(drop_seed), keep)
        # ---- forward --------------------------------------------------------------------------
        for i in range(n - 1):
            bn = self._bn(i)
            if bn is not None:
                if i == 0:
                    _call("lr_mlp_colstats_f32", _ptr(z[0]), B, w[0], _ptr(self.stat_partial[0]), s)
                if sync is not None:
                    sync(self.stat_partial[i])
                _call("lr_mlp_bn_finalize_f32", _ptr(self.stat_partial[i]), nblk, w[i], B, float(bn.eps), float(bn.momentum),
                      _ptr(bn.moving_mean), _ptr(bn.moving_var), _ptr(self.mean[i]), _ptr(self.inv[i]), s)
            lay = mlp.layers[i + 1]
            nxt = self._bn(i + 1) if i + 1 < n - 1 else None
            _call("lr_mlp_layer_fwd_f32", _ptr(z[i]), B, w[i],
                  _ptr(self.mean[i]) if bn is not None else 0, _ptr(self.inv[i]) if bn is not None else 0,
                  _ptr(P[bn.gamma]) if bn is not None else 0, _ptr(P[bn.beta]) if bn is not None else 0,
                  _ptr(P[lay.w]), _ptr(P[lay.b]), w[i + 1], _ptr(z[i + 1]),
                  _ptr(self.stat_partial[i + 1]) if nxt is not None else 0, int(drop_seed), keep, i, s)
        wo, bo = P[self.out.w], P[self.out.b]
        K, F, dn = self.K, self.F, w[-1]
        off = 1 if F > 0 else 0
        wl, bl = (P[self.linear.w], P[self.linear.b]) if off else (None, None)
        _call("lr_mlp_head_f32", _ptr(z[n - 1]), dn, _ptr(pair) if K > 0 else 0, K, _ptr(lin_out) if off else 0, F,
              _ptr(labels), _ptr(wl), _ptr(bl), _ptr(wo), _ptr(bo), B, 0, _ptr(self.gl), _ptr(self.head_partial), s)
        hp = self.head_partial
        self._reduce(hp, 0, off + K + dn, wo.grad, defer=True)
        self._reduce(hp, off + K + dn, 1, bo.grad, defer=True)
        if off:
            self._reduce(hp, 2 + K + dn, F, wl.grad, defer=True)
            self._reduce(hp, 2 + K + dn + F, 1, bl.grad, defer=True)
        self._reduce(hp, self.G, 1, self.loss_sum, defer=True, div=B)     # the mean: divided in the reduction's own launch
        # ---- backward -------------------------------------------------------------------------
        wd = wo[off + K:, 0]                     # the deep term's output weights (contiguous view)
        for i in range(n - 2, -1, -1):
            bn = self._bn(i)
            lay = mlp.layers[i + 1]
            last = i + 1 == n - 1
            up = None if last else self._bn(i + 1)
            args = [0 if last else 1, _ptr(self.gl) if last else 0, _ptr(wd) if last else 0,
                    0 if last else _ptr(self.gh[i + 1]), 0 if last else _ptr(z[i + 1])]
            if up is not None:
                args += [_ptr(self.mean[i + 1]), _ptr(self.inv[i + 1]), _ptr(P[up.gamma]), _ptr(P[up.gamma].grad), _ptr(P[up.beta].grad)]
            else:
                args += [0, 0, 0, 0, 0]
            args += [_ptr(z[i])]
            if bn is not None:
                args += [_ptr(self.mean[i]), _ptr(self.inv[i]), _ptr(P[bn.gamma]), _ptr(P[bn.beta])]
            else:
                args += [0, 0, 0, 0]
            args += [_ptr(P[lay.w]), w[i], w[i + 1], B, _ptr(self.gh[i]), _ptr(self.dW_partial[i]), _ptr(self.db_partial[i]),
                     _ptr(self.bn_partial[i]) if bn is not None else 0, int(drop_seed), keep, i, s]
            _call("lr_mlp_layer_bwd_f32", *args)
            self._reduce(self.dW_partial[i], 0, w[i] * w[i + 1], P[lay.w].grad, defer=True)
            self._reduce(self.db_partial[i], 0, w[i + 1], P[lay.b].grad, defer=True)
            if bn is not None:
                if sync is not None:
                    sync(self.bn_partial[i])
                self._reduce(self.bn_partial[i], 0, w[i], P[bn.beta].grad)       # sum gh        = d beta
                self._reduce(self.bn_partial[i], w[i], w[i], P[bn.gamma].grad)   # sum gh * xhat = d gamma
        if n >= 2:
            bn = self._bn(0)
            _call("lr_mlp_first_bwd_f32", _ptr(self.gh[0]), _ptr(z[0]),
                  _ptr(self.mean[0]) if bn is not None else 0, _ptr(self.inv[0]) if bn is not None else 0,
                  _ptr(P[bn.gamma]) if bn is not None else 0, _ptr(P[bn.gamma].grad) if bn is not None else 0,
                  _ptr(P[bn.beta].grad) if bn is not None else 0, B, w[0], _ptr(self.gz1), _ptr(self.sgz_partial), s)
            self._reduce(self.sgz_partial, 0, w[0], self.sgz1)
            gz1, sgz1 = self.gz1, self.sgz1
        else:                                     # the first Dense is the last layer: gz1 = gl (x) wd
            gz1 = torch.outer(self.gl, wd)
            sgz1 = gz1.sum(0)
        self._flush_deferred()
        return self.loss_sum[0], self.gl, gz1, sgz1

    def check(self) -> None:
        """Raise `TailBarrierError` if a one-launch step on any of this tail's buffer sets gave up on a grid barrier (the
        sticky word sync[18]; the losses of that step and of every later one are NaN).  Reads the device: call it where the
        loss is read back."""
        if not self._ran_fused:
            return
        sets = [getattr(self, "sync_words", None)] + [d.get("sync_words") for d in self._sets.values()]
        for w in sets:
            if w is not None and int(w[self.STICKY]) != 0:
                raise TailBarrierError(
                    "the fused DeepFM tail (lr_mlp_tail3_f32) gave up on a grid barrier: its workgroups were not all resident "
                    "(other kernels held compute units).  The parameters updated since then are not to be trusted; rerun with "
                    "LIBRECO_TAIL=chain.")

    def _run_fused(self, z1, pair, lin_out, labels, drop_seed: int, keep: float):
        """`run` as one persistent launch (`lr_mlp_tail3_f32`) + the one multi-job reduction of the weight / bias / head
        partials; the gradients of both BatchNorms, the batch statistics and the moving averages are written by the launch."""
        P, mlp, w = self.P, self.mlp, self.widths
        B = z1.shape[0]
        bn0, bn1 = self._bn(0), self._bn(1)
        l1, l2 = mlp.layers[1], mlp.layers[2]
        K, F, dn = self.K, self.F, w[-1]
        off = 1 if F > 0 else 0
        wo, bo = P[self.out.w], P[self.out.b]
        wl, bl = (P[self.linear.w], P[self.linear.b]) if off else (None, None)
        a = _lib.MlpTail3Args()
        a.B, a.K, a.F = B, K, F
        a.z0, a.pair, a.lin_out, a.labels = _ptr(z1), (_ptr(pair) if K > 0 else 0), (_ptr(lin_out) if off else 0), _ptr(labels)
        for i, bn in ((0, bn0), (1, bn1)):
            if bn is not None:
                vals = dict(eps=float(bn.eps), mom=float(bn.momentum), mm=_ptr(bn.moving_mean), mv=_ptr(bn.moving_var),
                            gamma=_ptr(P[bn.gamma]), beta=_ptr(P[bn.beta]), dgamma=_ptr(P[bn.gamma].grad), dbeta=_ptr(P[bn.beta].grad))
                for k, v in vals.items():
                    setattr(a, f"{k}{i}", v)
                setattr(a, f"stat{i}", _ptr(self.stat_partial[i]))
                setattr(a, f"bnp{i}", _ptr(self.bn_partial[i]))
                setattr(a, f"mean{i}", _ptr(self.mean[i]))
                setattr(a, f"inv{i}", _ptr(self.inv[i]))
        a.W1, a.b1, a.W2, a.b2 = _ptr(P[l1.w]), _ptr(P[l1.b]), _ptr(P[l2.w]), _ptr(P[l2.b])
        a.wl, a.bl, a.wo, a.bo = _ptr(wl), _ptr(bl), _ptr(wo), _ptr(bo)
        a.z1, a.z2, a.gh0, a.gh1 = _ptr(self.z[1]), _ptr(self.z[2]), _ptr(self.gh[0]), _ptr(self.gh[1])
        a.dW1p, a.db1p, a.dW2p, a.db2p = (_ptr(self.dW_partial[0]), _ptr(self.db_partial[0]), _ptr(self.dW_partial[1]),
                                          _ptr(self.db_partial[1]))
        a.headp, a.gl, a.gz0, a.sgzp = _ptr(self.head_partial), _ptr(self.gl), _ptr(self.gz1), _ptr(self.sgz_partial)
        a.drop_seed, a.keep, a.sync = drop_seed, keep, _ptr(self.sync_words)
        if int(self.spin_limit) != getattr(self, "_limit_set", 0):
            self.sync_words[self.LIMIT] = int(self.spin_limit)
            self._limit_set = int(self.spin_limit)
        self._ran_fused = True
        import ctypes as C

        _call("lr_mlp_tail3_f32", C.byref(a), ops._stream())
        hp = self.head_partial
        self._reduce(hp, 0, off + K + dn, wo.grad, defer=True)
        self._reduce(hp, off + K + dn, 1, bo.grad, defer=True)
        if off:
            self._reduce(hp, 2 + K + dn, F, wl.grad, defer=True)
            self._reduce(hp, 2 + K + dn + F, 1, bl.grad, defer=True)
        self._reduce(hp, self.G, 1, self.loss_sum, defer=True, div=B)     # the mean: divided in the reduction's own launch
        for i, lay in ((1, l2), (0, l1)):
            self._reduce(self.dW_partial[i], 0, w[i] * w[i + 1], P[lay.w].grad, defer=True)
            self._reduce(self.db_partial[i], 0, w[i + 1], P[lay.b].grad, defer=True)
        self._reduce(self.sgz_partial, 0, w[0], self.sgz1, defer=True)
        self._flush_deferred()
        return self.loss_sum[0], self.gl, self.gz1, self.sgz1
