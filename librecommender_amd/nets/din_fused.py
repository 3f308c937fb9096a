"""DIN training step (algorithms/din.py:165-250) as ONE chain of hand-written kernels, replayed as one hipGraph —
the reference runs the step as one `sess.run` (training/tf_trainer.py:76-101).

For the plain-id case ([user, item, plain sparse columns] fields, pure-id items, `din_attention`, relu MLP with or
without BatchNorm, cross-entropy loss, row-wise Adam) no torch autograd and no library GEMM is left in the step:

  ids (torch, int)      global rows of the field planes [Fp, B]
  lr_embed_gather_f32   field rows -> planes 0..Fp-1 of the MLP-input block  x [Fp + 1, B, K]
  lr_din_attn_pool_fwd  attention-pooled history -> plane Fp of the block (its `out` pointer)
  BlockFirstLayer       input BatchNorm folded into the first Dense, on the f32 MFMA kernels of the DeepFM step
  DeepFMTail            remaining layers, plain output layer, sigmoid cross-entropy, their backward
  BlockFirstLayer.bwd   dW1 / BatchNorm gradients; d loss / d x written plane by plane into ONE gradient buffer
                        [user | item | sparse... | attention-out | query | keys] whose leading planes ARE the row
                        gradients of the field rows
  lr_din_attn_pool_bwd  attention backward: query / key gradients behind them, parameter gradients into the flat buffer
  lr_segments_build     + lr_embed_scatter_adam_dc_f32: de-duplicated row-wise Adam of every touched row
  lr_adam_dense_dc_f32  all dense parameters

Adam's step-dependent coefficients live in device memory (`ops.AdamCoefBuffer`), so the captured graph is replayed
unchanged every step.  Replays run on a dedicated non-default stream, ordered against the caller's stream by events
(`GraphRunner`)."""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Dict, Optional

import torch

from .. import ops
from ..layers.dense import BlockFirstLayer
from ..layers.tail import DeepFMTail


class GraphNotCapturable(RuntimeError):
    """A captured step holds memset / memcpy nodes (`GraphRunner.check_kernel_nodes_only`): the owner launches such steps eagerly."""


class GraphRunner:
    """Capture-once / replay-many helper shared by the fused steps.

    Replays are launched on ONE dedicated non-default stream per runner and ordered against the caller's current
    stream with events on both sides (`wait_stream`), so that (i) the copy of the step's inputs into the graph's
    static buffers, the coefficient store and the replay are stream-ordered among themselves, and (ii) eager work
    the caller enqueues afterwards (a device-side loader producing the next batch, a step of another batch shape that
    shares workspaces) starts only after the replay has finished.

    A captured step must consist of KERNEL nodes only: with memset / memcpy nodes in the graph (hipMemsetAsync of a
    counter, a `clone()` / contiguous `copy_`) replays raced against their neighbouring kernel nodes as soon as another
    stream kept the device busy (device-side loader) — memory faults at varying addresses in `DeepFM.fit`, gone with
    AMD_SERIALIZE_KERNEL=3 (profiles/r03_graph_fault.md).  The C-ABI zeroes its counters with a kernel
    (`zero_words_async`), the nets use elementwise kernels instead of copies inside a step."""

    # Inside `lazy_join()` (the trainer's epoch loop over a device-side loader) a replay does NOT make the caller's
    # stream wait for it: the loader kernels of the next batch — enqueued on the caller's stream — run beside the
    # replayed step instead of behind it.  Everything that reads model state on the caller's stream must `join_all()`
    # first (eager steps do; leaving `lazy_join()` does).
    LAZY = False            # default of runners without a setting of their own (`lazy_join()` without a model)
    _live: "weakref.WeakSet" = None

    def __init__(self, device: torch.device):
        self.device = device
        self.lazy: Optional[bool] = None        # per-runner setting (`lazy_join(model=...)`): None = follow the class default
        self.stream: Optional[torch.cuda.Stream] = None
        self.graphs: Dict[tuple, dict] = {}
        self.pending = False
        if GraphRunner._live is None:
            GraphRunner._live = weakref.WeakSet()
        GraphRunner._live.add(self)

    def clear(self) -> None:
        self.join()
        self.graphs = {}

    def join(self) -> None:
        if self.pending and self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.pending = False

    @classmethod
    def join_all(cls) -> None:
        for r in list(cls._live or ()):
            r.join()

    def _stream(self) -> torch.cuda.Stream:
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        return self.stream

    def capture(self, key, build):
        """`build()` enqueues the step on the current stream and returns its static outputs."""
        self.join()
        st = self.graphs.setdefault(key, {})
        side = self._stream()
        side.wait_stream(torch.cuda.current_stream(self.device))
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph(keep_graph=True)          # the hipGraph_t stays around: its nodes are checked before use
        with torch.cuda.graph(g, stream=side):
            st["out"] = build()
        torch.cuda.current_stream(self.device).wait_stream(side)
        try:
            self.check_kernel_nodes_only(g)
        except GraphNotCapturable:
            self.graphs.pop(key, None)
            raise
        g.instantiate()
        st["graph"] = g
        return st

    @staticmethod
    def check_kernel_nodes_only(g) -> int:
        """the fix for the replay faults rests on captured steps holding kernel nodes only — enforced here: a
        memset / memcpy node (a `copy_` / `clone()` inside a step, a library call that clears its scratch with hipMemsetAsync)
        raises instead of replaying.  Returns the node count."""
        import ctypes as C

        from .. import _lib

        n = C.c_int(0)
        foreign = _lib.load().lr_graph_foreign_nodes(C.c_void_p(g.raw_cuda_graph()), C.byref(n))
        if foreign < 0:
            raise RuntimeError(f"hipGraphGetNodes failed: {_lib.load().lr_strerror(-foreign).decode()}")
        if foreign > 0:
            raise GraphNotCapturable(f"the captured step holds {foreign} memset / memcpy node(s) among {n.value}: a captured step "
                               f"must consist of kernel launches only (use elementwise kernels instead of copy_ / clone, and "
                               f"launch steps whose library calls clear scratch with memsets eagerly)")
        return n.value

    def replay(self, key, feed, tensors=()):
        """`feed()` enqueues the input copies / coefficient store; `tensors`: caller tensors read by `feed` (kept alive
        for the side stream by `record_stream`).  Returns the graph's static output (overwritten by the next replay) —
        inside `lazy_join()` a copy of it made on the replay stream (attribute `_lr_own`), valid after a `join()`."""
        st = self.graphs[key]
        cur = torch.cuda.current_stream(self.device)
        side = self._stream()
        side.wait_stream(cur)
        lazy = GraphRunner.LAZY if self.lazy is None else self.lazy
        with torch.cuda.stream(side):
            feed()
            st["graph"].replay()
            out = st["out"]
            if lazy and isinstance(out, torch.Tensor):
                out = out + 0.0
                out._lr_own = True
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(side)
        if lazy:
            self.pending = True
        else:
            cur.wait_stream(side)
            self.pending = False
        return out


def _runners_of(model):
    net = getattr(model, "net", model)
    out = []
    for r in (getattr(net, "_runner", None), getattr(getattr(net, "_fstep", None), "runner", None), getattr(net, "runner", None)):
        if isinstance(r, GraphRunner):
            out.append(r)
    return out


@contextlib.contextmanager
def lazy_join(enabled: bool = True, model=None):
    """Scope in which graph replays are not joined to the caller's stream one by one (see `GraphRunner.LAZY`).  With `model`
    (the trainer passes its own) only THAT model's runners are switched — the setting is per trainer, not process-global
    (ADVICE r03); runners created inside the scope (the first captured step) pick it up through the scope's own default."""
    if model is None:
        prev = GraphRunner.LAZY
        GraphRunner.LAZY = bool(enabled)
        try:
            yield
        finally:
            GraphRunner.LAZY = prev
            if not prev:
                GraphRunner.join_all()
        return
    seen = {}

    def apply():
        for r in _runners_of(model):
            if id(r) not in seen:
                seen[id(r)] = (r, r.lazy)
                r.lazy = bool(enabled)

    apply()
    net = getattr(model, "net", model)
    prev_hook = getattr(net, "_lazy_scope", None)
    net._lazy_scope = apply            # nets call it after creating a runner lazily (first fused step)
    try:
        yield
    finally:
        net._lazy_scope = prev_hook
        for r, prev in seen.values():
            r.lazy = prev
            r.join()


class _Bufs:
    pass


class FusedDINStep:
    """Per (B, L) buffer set + captured graph of the fused DIN step for one `FeatDINNet`."""

    def __init__(self, net):
        self.net = net
        self.sets: Dict[tuple, _Bufs] = {}
        self.runner = GraphRunner(net.device)
        self.warm = 1          # eager steps of a shape before it is captured (lazy initialisation outside the capture)

    GRAPH_MAX_IDS = 1 << 22     # id-stream length up to which the step is captured: the library's own radix sort (kernels only)

    # ---- eligibility ------------------------------------------------------------------------
    @staticmethod
    def supported(net) -> bool:
        import torch.nn.functional as Fn

        s = net.spec
        if not net.fused or s.pooled or s.n_dense_cols or net.dense_adam or net.reg:
            return False
        mlp = net.mlp
        if mlp.dropout_rate or mlp.act is not Fn.relu or len(mlp.layers) < 2:
            return False
        H1 = net.P[mlp.layers[0].w].shape[1]
        return BlockFirstLayer.supported(net.K, H1) and DeepFMTail.supported(mlp)

    # ---- buffers ----------------------------------------------------------------------------
    def _set(self, B: int, L: int) -> _Bufs:
        key = (B, L)
        if key in self.sets:
            return self.sets[key]
        net, dev = self.net, self.net.device
        K, Fp = net.K, net.spec.n_fields
        Pn = Fp + 1
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        b = _Bufs()
        b.B, b.L, b.Fp, b.Pn = B, L, Fp, Pn
        b.xbuf = torch.empty((Pn, B, K), **f32)
        b.attn = torch.empty((B, L), **f32)
        n_pos = Pn * B + B + B * L                   # [field planes | attention-out (dropped) | query | keys]
        b.gbuf = torch.empty((n_pos, K), **f32)
        b.ids = torch.empty(n_pos, **i32)
        b.idsP = b.ids[:Fp * B].view(Fp, B)          # the field planes' rows double as the head of the id stream
        b.ids[Fp * B:Pn * B] = -1                    # attention-output plane: not a table row
        b.l1 = BlockFirstLayer(net.P, net.mlp.bn_in, net.mlp.layers[0], Pn, K, B, dev)
        b.tail = DeepFMTail(net.P, net.mlp, None, net.out, 0, 0, dev)
        b.seg = ops.SegmentBuilder(n_pos, net.tables.V, dev)
        lib = ops._lib.load()
        b.att_ws = torch.empty(max(lib.lr_din_attn_ws_bytes(B, L, K, 16), 8), dtype=torch.uint8, device=dev)
        # the attention MLP's hidden activations, forward -> backward (26 MB at cfg 3): the backward does not recompute them
        mfma_shape = K in (16, 32, 64, 128) and L <= 2048
        # the samples in the order the backward attention kernels' waves take them (written by the forward's launch: balanced by length)
        b.order = torch.empty(B, **i32) if mfma_shape and os.environ.get("LIBRECO_DIN_ORDER", "1") != "0" else None
        b.att_hid = (torch.empty(ops.din_hid_floats(B, L, K, b.order is not None), **f32)
                     if mfma_shape and os.environ.get("LIBRECO_DIN_SAVED_H", "1") != "0" else None)
        plain = net.spec.plain_cols
        b.plain_all = plain == list(range(net.spec.n_sparse_cols))
        b.plain_idx = torch.tensor(plain, dtype=torch.int64, device=dev) if plain else None
        b.plain_cols32 = torch.tensor(plain, dtype=torch.int32, device=dev) if plain else None
        self.sets[key] = b
        return b

    # ---- the step -----------------------------------------------------------------------------
    @torch.no_grad()
    def _core(self, b: _Bufs, users, items, sparse, seqs, lens, labels, hp):
        """Enqueue one training step on the current stream.  All arguments are int32 / fp32 device tensors."""
        net, t, P = self.net, self.net.tables, self.net.P
        B, L, Fp, Pn, K = b.B, b.L, b.Fp, b.Pn, net.K
        # ---- the step's id stream (field planes plane-major | -1 | query rows | window rows, pads -1): one launch ----
        ops.din_build_ids(users, items, sparse if b.plain_idx is not None else None,
                          None if (b.plain_idx is None or b.plain_all) else b.plain_cols32, seqs, lens,
                          t.user_off, t.item_off, t.sparse_off, b.ids)
        # ---- id stream of the table update + its segment build (radix sort + scan: a dozen small latency-bound
        # launches that depend on the ids only) on a side stream, beside the forward / backward kernels; joined in
        # front of the scatter.  Inside a capture this is a fork / join of the graph.
        cur = torch.cuda.current_stream(net.device)
        if getattr(b, "side", None) is None:
            b.side = torch.cuda.Stream(device=net.device)
        n0 = Pn * B
        fork = os.environ.get("LIBRECO_DIN_FORK", "1") != "0"       # 0: the segment build in line (profiling switch)
        if fork:
            b.side.wait_stream(cur)
            with torch.cuda.stream(b.side):
                seg = b.seg.build(b.ids)
        else:
            seg = b.seg.build(b.ids)
        # ---- forward --------------------------------------------------------------------------------
        x2 = b.xbuf.view(Pn * B, K)
        ops.embed_gather(t.embed, b.idsP.reshape(-1), out=x2[:Fp * B])
        W1, b1, W2, b2 = net._att_params()
        item_tab = t.variable("item_embeds_var")
        ops.din_attn_pool_fwd(item_tab, items, seqs, lens, W1, b1, W2, b2, out=b.xbuf[Fp], attn=b.attn, hid=b.att_hid, order_out=b.order)
        z1 = b.l1.forward(b.xbuf)
        loss, gl, gz1, sgz1 = b.tail.run(z1, None, None, labels)
        # ---- backward -------------------------------------------------------------------------------
        b.l1.backward(gz1, sgz1, b.gbuf)
        # (pad positions carry id -1 in `b.ids`, the table update never reads their gradient rows: not zeroed.  The
        # parameter half of the backward re-reads the key rows, so it cannot run beside the table update that moves them.)
        ops.din_attn_pool_bwd(item_tab, items, seqs, lens, W1, b1, W2, b2, b.attn, b.gbuf[Fp * B:n0],
                              gq_out=b.gbuf[n0:n0 + B], gkey_out=b.gbuf[n0 + B:].view(B, L, K),
                              param_out=(W1.grad, b1.grad, W2.grad, b2.grad), ws=b.att_ws, keep_pad_rows=True, hid=b.att_hid, order=b.order)
        if fork:
            cur.wait_stream(b.side)
        ops.embed_scatter_adam(t.embed, t.m, t.v, b.gbuf, seg, hp)
        P.adam_step(hp)
        return loss

    def train_step(self, users, items, sparse, seqs, lens, labels, use_graph: bool):
        net = self.net
        B, L = seqs.shape
        b = self._set(B, L)
        # Above 4 M ids lr_segments_build hands the sort to rocPRIM, whose onesweep form clears its histograms with memset
        # calls: a captured step must hold kernel nodes only (GraphRunner checks it) — such steps launch eagerly.
        if b.ids.numel() > self.GRAPH_MAX_IDS:
            use_graph = False
        if not use_graph:
            self.runner.join()
            return self._core(b, users, items, sparse, seqs, lens, labels, net._hp())
        key = (B, L)
        st = self.runner.graphs.get(key)
        if st is None or "graph" not in st:
            seen = getattr(b, "seen", 0) + 1
            b.seen = seen
            if seen <= self.warm:
                self.runner.join()
                return self._core(b, users, items, sparse, seqs, lens, labels, net._hp())
            b.s_in = [x.clone() if x is not None else None for x in (users, items, sparse, seqs, lens, labels)]
            b.coef = ops.AdamCoefBuffer(net.device)
            b.coef.set(net._hp())
            try:
                self.runner.capture(key, lambda: self._core(b, *b.s_in, b.coef))     # records; runs nothing
            except GraphNotCapturable as e:       # this shape launches eagerly from now on
                import warnings

                warnings.warn(f"DIN step of shape {key} is not captured: {e}")
                b.seen = -(1 << 60)
                return self._core(b, users, items, sparse, seqs, lens, labels, net._hp())
            return self.runner.replay(key, lambda: None)
        ins = (users, items, sparse, seqs, lens, labels)

        def feed():
            for dst, src in zip(b.s_in, ins):
                if dst is not None:
                    dst.copy_(src, non_blocking=True)
            b.coef.set(net._hp())

        return self.runner.replay(key, feed, ins)
