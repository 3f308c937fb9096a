"""Tensor-level wrappers over the C-ABI (``include/libreco_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below
validates its arguments, passes raw device pointers + the current stream to the shared
library, and raises if the tensors are not on a HIP device (no CPU path exists).
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import AdamHP, COMBINERS, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of selected C-ABI launches on the stream they are enqueued on
    (bench.py's roofline leg).  Disabled by default: zero overhead on the product path."""

    def __init__(self):
        self.names = set()
        self.events = {}

    def enable(self, *names):
        self.names = set(names)
        self.events = {n: [] for n in names}

    def disable(self):
        self.names = set()

    def summary(self):
        """name -> (launches, mean_ms); call after torch.cuda.synchronize()."""
        out = {}
        for n, evs in self.events.items():
            if evs:
                ms = [a.elapsed_time(b) for a, b in evs]
                out[n] = (len(ms), sum(ms) / len(ms))
        return out


TIMER = KernelTimer()


def _call(name: str, *args) -> None:
    fn = getattr(_lib.load(), name)
    if name in TIMER.names:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        TIMER.events[name].append((a, b))
    else:
        rc = fn(*args)
    check(rc, name)


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype: torch.dtype, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the libreco HIP kernels only run on an MI355X device "
            "tensor (there is no CPU fallback)"
        )
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-d, got shape {tuple(t.shape)}")
    return t


def adam_hp(lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-5,
            weight_decay: float = 0.0, tf_style: bool = True) -> AdamHP:
    if step < 1:
        raise ValueError("Adam step counts from 1")
    return AdamHP(lr, beta1, beta2, eps, weight_decay, int(step), 1 if tf_style else 0)


# --------------------------------------------------------------------------------------
# gather / pooling
# --------------------------------------------------------------------------------------
def embed_gather(table: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[..., :] = table[idx[...], :]`` (layers/embedding.py:23)."""
    _req(table, torch.float32, "table")
    _req(idx, torch.int32, "idx")
    t2 = table if table.dim() == 2 else table.reshape(table.shape[0], -1)
    V, K = t2.shape
    n = idx.numel()
    if out is None:
        out = torch.empty((*idx.shape, K), dtype=torch.float32, device=table.device)
    else:
        _req(out, torch.float32, "out")
        if out.numel() != n * K:
            raise ValueError("out has the wrong size")
    _call("lr_embed_gather_f32", _ptr(t2), V, K, _ptr(idx), n, _ptr(out), _stream())
    return out


def embed_bag_pool(table: torch.Tensor, idx: torch.Tensor, combiner: str, oov: int) -> torch.Tensor:
    """Fixed-length bag pooling with OOV->0 (tfops/features.py:90-118)."""
    _req(table, torch.float32, "table", 2)
    _req(idx, torch.int32, "idx", 2)
    V, K = table.shape
    nbags, bag_len = idx.shape
    out = torch.empty((nbags, K), dtype=torch.float32, device=table.device)
    _call("lr_embed_bag_pool_f32", _ptr(table), V, K, _ptr(idx), nbags, bag_len,
                                            COMBINERS[combiner], int(oov), _ptr(out), _stream())
    return out


def embed_bag_pool_bwd(gout: torch.Tensor, idx: torch.Tensor, V: int, combiner: str, oov: int) -> torch.Tensor:
    _req(gout, torch.float32, "gout", 2)
    _req(idx, torch.int32, "idx", 2)
    nbags, bag_len = idx.shape
    K = gout.shape[1]
    gentry = torch.empty((nbags * bag_len, K), dtype=torch.float32, device=gout.device)
    _call("lr_embed_bag_pool_bwd_f32", _ptr(gout), K, _ptr(idx), V, nbags, bag_len,
                                                COMBINERS[combiner], int(oov), _ptr(gentry),
                                                _stream())
    return gentry


def pair_dot(U: torch.Tensor, I: torch.Tensor, user: torch.Tensor, item: torch.Tensor) -> torch.Tensor:
    """``out[i] = <U[user[i]], I[item[i]]>`` (prediction/predict.py:36-40)."""
    _req(U, torch.float32, "U", 2)
    _req(I, torch.float32, "I", 2)
    _req(user, torch.int32, "user", 1)
    _req(item, torch.int32, "item", 1)
    if U.shape[1] != I.shape[1] or user.numel() != item.numel():
        raise ValueError("shape mismatch")
    out = torch.empty(user.numel(), dtype=torch.float32, device=U.device)
    _call("lr_pair_dot_f32", _ptr(U), U.shape[0], _ptr(I), I.shape[0], U.shape[1],
                                      _ptr(user), _ptr(item), user.numel(), _ptr(out), _stream())
    return out


# --------------------------------------------------------------------------------------
# segments ("CSR by touched row") + gradient scatter
# --------------------------------------------------------------------------------------
@dataclass
class Segments:
    """Device-side grouping of batch positions by table row (see lr_segments_build)."""

    pos: torch.Tensor      # int32 [n]
    rows: torch.Tensor     # int32 [n]      (first n_seg valid)
    start: torch.Tensor    # int32 [n + 1]  (first n_seg + 1 valid)
    n_seg: torch.Tensor    # int32 [1], device
    n: int
    V: int
    slots: Optional[torch.Tensor] = None   # int32 [n]: run number of every position (on request)
    slotT: Optional[torch.Tensor] = None   # int32 [F,B]: index of position (b,f) in `pos` (FieldSegmentBuilder)
    runT: Optional[torch.Tensor] = None    # int32 [F,B]: run number of position (b,f) (FieldSegmentBuilder(want_runs=True))
    owner: Optional[object] = None         # the builder: keeps the long-run workspaces of the scatter kernels

    def long_ws(self, K: int) -> Optional[torch.Tensor]:
        """Persistent workspace for the long-run (Zipf head) path of the scatter kernels (see lr_embed_scatter_ws_bytes)."""
        return self.owner.long_ws(K) if self.owner is not None and hasattr(self.owner, "long_ws") else None

    def count(self) -> int:
        """Host sync — for tests and logging only."""
        return int(self.n_seg.item())


class SegmentBuilder:
    """Reusable workspace for ``lr_segments_build`` (no per-step allocation)."""

    def __init__(self, n_max: int, V: int, device: torch.device):
        self.n_max, self.V, self.device = int(n_max), int(V), device
        nbytes = _lib.load().lr_segments_ws_bytes(self.n_max, self.V)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.pos = torch.empty(self.n_max, dtype=torch.int32, device=device)
        self.rows = torch.empty(self.n_max, dtype=torch.int32, device=device)
        self.start = torch.empty(self.n_max + 1, dtype=torch.int32, device=device)
        self.n_seg = torch.zeros(1, dtype=torch.int32, device=device)
        self.slots = None
        self._long_ws = {}

    def long_ws(self, K: int) -> torch.Tensor:
        ws = self._long_ws.get(K)
        if ws is None:
            ws = self._long_ws[K] = torch.empty(max(_lib.load().lr_embed_scatter_ws_bytes(self.n_max, K), 256),
                                                dtype=torch.uint8, device=self.device)
        return ws

    def build(self, idx: torch.Tensor, want_slots: bool = False) -> Segments:
        """``want_slots``: also emit the run number of every position (``seg.slots``)."""
        _req(idx, torch.int32, "idx")
        n = idx.numel()
        if n > self.n_max:
            raise ValueError(f"idx has {n} entries, builder was sized for {self.n_max}")
        if want_slots and self.slots is None:
            self.slots = torch.empty(self.n_max, dtype=torch.int32, device=self.device)
        _call("lr_segments_build", _ptr(idx), n, self.V, _ptr(self.pos), _ptr(self.rows),
                                            _ptr(self.start), _ptr(self.n_seg),
                                            _ptr(self.slots) if want_slots else None, _ptr(self.ws),
                                            self.ws.numel(), _stream())
        seg = Segments(self.pos, self.rows, self.start, self.n_seg, n, self.V, owner=self)
        seg.slots = self.slots[:n] if want_slots else None
        return seg


def build_segments(idx: torch.Tensor, V: int) -> Segments:
    return SegmentBuilder(max(idx.numel(), 1), V, idx.device).build(idx)


class FieldSegmentBuilder:
    """Reusable buffers for ``lr_segments_build_fields`` (idx [B,F] whose column f only holds rows of
    field f): same `Segments` as `SegmentBuilder.build(idx.reshape(-1))`, plus ``seg.slotT`` [F,B] —
    the index of position (b,f) in ``seg.pos`` (-1 for dropped entries)."""

    MAX_B = 16384

    def __init__(self, B_max: int, F: int, V: int, device: torch.device, want_runs: bool = False):
        self.B_max, self.F, self.V, self.device = int(B_max), int(F), int(V), device
        n = self.B_max * self.F
        self.ws = torch.empty(_lib.load().lr_segments_fields_ws_bytes(self.B_max, self.F), dtype=torch.uint8, device=device)
        self.pos = torch.empty(n, dtype=torch.int32, device=device)
        self.rows = torch.empty(n, dtype=torch.int32, device=device)
        self.start = torch.empty(n + 1, dtype=torch.int32, device=device)
        self.n_seg = torch.zeros(1, dtype=torch.int32, device=device)
        self.slotT = torch.empty((self.F, self.B_max), dtype=torch.int32, device=device)
        self.runT = torch.empty((self.F, self.B_max), dtype=torch.int32, device=device) if want_runs else None

    def build(self, idxT: torch.Tensor, field_row_start: torch.Tensor) -> Segments:
        """``want_runs`` builders also set ``seg.runT`` [F,B]: the run number of every position (-1 if dropped)."""
        _req(idxT, torch.int32, "idxT", 2)
        _req(field_row_start, torch.int32, "field_row_start", 1)
        F, B = idxT.shape
        if F != self.F or B > self.B_max or field_row_start.numel() != F + 1:
            raise ValueError("idxT does not match this builder")
        cut = (lambda t: t if B == self.B_max else t.view(-1)[: F * B].view(F, B))
        slotT = cut(self.slotT)
        seg = Segments(self.pos, self.rows, self.start, self.n_seg, B * F, self.V)
        if self.runT is None:
            _call("lr_segments_build_fields", _ptr(idxT), B, F, _ptr(field_row_start), _ptr(self.pos), _ptr(self.rows),
                  _ptr(self.start), _ptr(self.n_seg), _ptr(slotT), _ptr(self.ws), self.ws.numel(), _stream())
        else:
            seg.runT = cut(self.runT)
            _call("lr_segments_build_fields_runs", _ptr(idxT), B, F, _ptr(field_row_start), _ptr(self.pos), _ptr(self.rows),
                  _ptr(self.start), _ptr(self.n_seg), _ptr(slotT), _ptr(seg.runT), _ptr(self.ws), self.ws.numel(), _stream())
        seg.slotT = slotT
        return seg


class OwnerPartition:
    """Reusable buffers for ``lr_owner_partition_i32``: the stable owner-major order of a batch's distinct rows under
    round-robin row sharding (owner = row % W).  `run(seg)` -> (perm [n_max] int32: place of run r, send_ids [n_max] int32:
    local rows in that order, counts [W + 1] int64 on the device: per owner, then the total)."""

    def __init__(self, n_max: int, W: int, device: torch.device):
        self.n_max, self.W = int(n_max), int(W)
        self.ws = torch.empty(max(_lib.load().lr_owner_partition_ws_bytes(self.n_max, self.W), 256), dtype=torch.uint8, device=device)
        self.perm = torch.empty(self.n_max, dtype=torch.int32, device=device)
        self.send_ids = torch.empty(self.n_max, dtype=torch.int32, device=device)
        self.counts = torch.zeros(self.W + 1, dtype=torch.int64, device=device)

    def run(self, rows: torch.Tensor, n_seg: torch.Tensor):
        _req(rows, torch.int32, "rows", 1)
        _req(n_seg, torch.int32, "n_seg", 1)
        if rows.numel() > self.n_max:
            raise ValueError("rows exceed this partition's capacity")
        _call("lr_owner_partition_i32", _ptr(rows), _ptr(n_seg), rows.numel(), self.W, _ptr(self.perm), _ptr(self.send_ids),
              _ptr(self.counts), _ptr(self.ws), self.ws.numel(), _stream())
        return self.perm, self.send_ids, self.counts


def idx_transpose(idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,F] int32 -> [F,B] (the field-major id layout of the per-field kernels)."""
    _req(idx, torch.int32, "idx", 2)
    B, F = idx.shape
    if out is None:
        out = torch.empty((F, B), dtype=torch.int32, device=idx.device)
    _call("lr_idx_transpose_i32", _ptr(idx), B, F, _ptr(out), _stream())
    return out


def din_build_ids(users, items, sparse, cols, seqs, lens, user_off: int, item_off: int, sparse_off: int,
                  out: torch.Tensor) -> torch.Tensor:
    """The id stream of the fused DIN step in one launch (see lr_din_build_ids_i32).  `sparse` [B, n_sp] int32 or None, `cols`
    int32 [n_plain] (the plain sparse columns) or None = all of them."""
    for t_, n_ in ((users, "users"), (items, "items"), (seqs, "seqs"), (lens, "lens"), (out, "out")):
        _req(t_, torch.int32, n_)
    B, L = seqs.shape
    n_plain = 0
    if sparse is not None:
        _req(sparse, torch.int32, "sparse", 2)
        n_plain = int(cols.numel()) if cols is not None else sparse.shape[1]
        if cols is not None:
            _req(cols, torch.int32, "cols", 1)
    if out.numel() != (2 + n_plain + 2 + L) * B or not seqs.is_contiguous() or (sparse is not None and not sparse.is_contiguous()):
        raise ValueError("`out` must hold (2 + n_plain + 2 + L) * B ids; seqs / sparse contiguous")
    _call("lr_din_build_ids_i32", _ptr(users), _ptr(items), _ptr(sparse), sparse.shape[1] if sparse is not None else 0,
          _ptr(cols), n_plain, _ptr(seqs), _ptr(lens), B, L, int(user_off), int(item_off), int(sparse_off), _ptr(out), _stream())
    return out


def _din_order(order, B):
    if order is not None and (order.dtype != torch.int32 or not order.is_contiguous() or order.numel() != B):
        raise ValueError("order must be a contiguous int32 tensor of B elements")
    return order


def embed_segment_sum(grad: torch.Tensor, seg: Segments) -> torch.Tensor:
    _req(grad, torch.float32, "grad")
    K = grad.shape[-1]
    if grad.numel() != seg.n * K:
        raise ValueError("grad rows must match the segmented index count")
    grows = torch.zeros((max(seg.n, 1), K), dtype=torch.float32, device=grad.device)
    ws = seg.long_ws(K)
    _call("lr_embed_segment_sum_f32", _ptr(grad), K, _ptr(seg.pos), _ptr(seg.start), _ptr(seg.n_seg), seg.n, _ptr(grows),
          _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return grows


def embed_scatter_add(table: torch.Tensor, grad: torch.Tensor, seg: Segments, alpha: float = 1.0) -> None:
    _req(table, torch.float32, "table", 2)
    _req(grad, torch.float32, "grad")
    V, K = table.shape
    if grad.numel() != seg.n * K or V != seg.V:
        raise ValueError("shape mismatch")
    ws = seg.long_ws(K)
    _call("lr_embed_scatter_add_f32", _ptr(table), V, K, _ptr(grad), _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start),
          _ptr(seg.n_seg), seg.n, float(alpha), _ptr(ws), 0 if ws is None else ws.numel(), _stream())


def embed_scatter_adam(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor,
                       seg: Segments, hp) -> None:
    """`hp`: an `AdamHP` (by value) or an `AdamCoefBuffer` (device-resident coefficients, graph-capturable).
    `grad` may hold more rows than `seg.n` (a prefix of a larger buffer)."""
    _req(table, torch.float32, "table", 2)
    _req(m, torch.float32, "m", 2)
    _req(v, torch.float32, "v", 2)
    _req(grad, torch.float32, "grad")
    V, K = table.shape
    if grad.numel() < seg.n * K or V != seg.V or m.shape != table.shape or v.shape != table.shape:
        raise ValueError("shape mismatch")
    dc = isinstance(hp, AdamCoefBuffer)
    ws = seg.long_ws(K)
    _call("lr_embed_scatter_adam_dc_f32" if dc else "lr_embed_scatter_adam_f32", _ptr(table), _ptr(m), _ptr(v), V, K,
          _ptr(grad), _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg), seg.n,
          _ptr(hp.dev) if dc else hp, _ptr(ws), 0 if ws is None else ws.numel(), _stream())


def embed_scatter_adam_lin(table, m, v, grad, lin, lin_m, lin_v, glin, seg: Segments, hp: AdamHP) -> None:
    """`embed_scatter_adam` on a table and on its per-row linear weight ([V,1]) in one pass over the segments."""
    for t_, n_ in ((table, "table"), (m, "m"), (v, "v"), (grad, "grad"), (lin, "lin"), (lin_m, "lin_m"), (lin_v, "lin_v"),
                   (glin, "glin")):
        _req(t_, torch.float32, n_)
    V, K = table.shape
    if grad.numel() != seg.n * K or glin.numel() != seg.n or V != seg.V or lin.numel() != V:
        raise ValueError("shape mismatch")
    dc = isinstance(hp, AdamCoefBuffer)
    _call("lr_embed_scatter_adam_lin_dc_f32" if dc else "lr_embed_scatter_adam_lin_f32", _ptr(table), _ptr(m), _ptr(v),
          V, K, _ptr(grad), _ptr(lin), _ptr(lin_m), _ptr(lin_v), _ptr(glin), _ptr(seg.pos), _ptr(seg.rows),
          _ptr(seg.start), _ptr(seg.n_seg), seg.n, _ptr(hp.dev) if dc else hp, _stream())


def embed_peer_adam(table, m, v, grad, ids, peer_counts, hp: AdamHP, lin=None, lin_m=None, lin_v=None, glin=None,
                    peer_tab: Optional[torch.Tensor] = None) -> None:
    """The owner-side update of the row-sharded tables from the peers' de-duplicated lists (`lr_embed_peer_adam_f32`):
    `ids` int32 [n] local rows (the W lists back to back, `peer_counts` a host list of W lengths), `grad` [n, K] and
    `glin` [n] their gradients, `peer_tab` int32 [V * W] zeros (W > 1)."""
    import ctypes as C

    for t_, n_ in ((table, "table"), (m, "m"), (v, "v"), (grad, "grad")):
        _req(t_, torch.float32, n_)
    _req(ids, torch.int32, "ids", 1)
    V, K = table.shape
    W = len(peer_counts)
    n = int(sum(peer_counts))
    if ids.numel() != n or grad.numel() != n * K or (glin is not None and glin.numel() != n):
        raise ValueError("shape mismatch")
    if W > 1 and (peer_tab is None or peer_tab.numel() < V * W or peer_tab.dtype != torch.int32):
        raise ValueError("peer_tab must be int32 [V * W]")
    counts = (C.c_int64 * W)(*[int(c) for c in peer_counts])
    _call("lr_embed_peer_adam_f32", _ptr(table), _ptr(m), _ptr(v), V, K, _ptr(grad), _ptr(lin), _ptr(lin_m), _ptr(lin_v),
          _ptr(glin), _ptr(ids), C.cast(counts, C.c_void_p), W, _ptr(peer_tab), hp, _stream())


def adam_dense(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, hp: AdamHP,
               grows: Optional[torch.Tensor] = None, seg: Optional[Segments] = None,
               row_slot: Optional[torch.Tensor] = None, l2: float = 0.0,
               vmax: Optional[torch.Tensor] = None) -> None:
    """Dense Adam over every row (TF1: training/tf_trainer.py:120; torch: torch_trainer.py:63-69).
    ``vmax`` (same shape as ``v``) switches on AMSGrad."""
    if vmax is not None:
        _req(vmax, torch.float32, "vmax")
    _req(table, torch.float32, "table")
    t2 = table.reshape(table.shape[0], -1) if table.dim() != 2 else table
    V, K = t2.shape
    n_max = 0
    if seg is None and grows is not None:  # full dense gradient (MLP / BN parameters)
        _req(grows, torch.float32, "grad")
        if grows.numel() != t2.numel():
            raise ValueError("dense gradient must have the parameter's size")
        _call("lr_adam_dense_f32", _ptr(t2), _ptr(m), _ptr(v), _ptr(vmax), V, K, _ptr(grows), 0, 0, 0,
                                            0, float(l2), hp, _stream())
        return
    if seg is not None and grows is not None and seg.n > 0:
        n_max = seg.n
        if row_slot is None:
            row_slot = torch.full((V,), -1, dtype=torch.int32, device=table.device)
        _req(row_slot, torch.int32, "row_slot", 1)
    _call("lr_adam_dense_f32", _ptr(t2), _ptr(m), _ptr(v), _ptr(vmax), V, K, _ptr(grows),
                                        _ptr(seg.rows) if n_max else 0,
                                        _ptr(seg.n_seg) if n_max else 0, n_max,
                                        _ptr(row_slot) if n_max else 0, float(l2), hp, _stream())


# --------------------------------------------------------------------------------------
# FM
# --------------------------------------------------------------------------------------
def fm_pairwise_fwd(e: torch.Tensor, want_sum: bool = True):
    _req(e, torch.float32, "e", 3)
    B, F, K = e.shape
    pair = torch.empty((B, K), dtype=torch.float32, device=e.device)
    fsum = torch.empty((B, K), dtype=torch.float32, device=e.device) if want_sum else None
    _call("lr_fm_pairwise_fwd_f32", _ptr(e), B, F, K, _ptr(pair), _ptr(fsum), _stream())
    return pair, fsum


def fm_pairwise_bwd(e: torch.Tensor, fsum: torch.Tensor, gpair: torch.Tensor,
                    ge: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(e, torch.float32, "e", 3)
    _req(fsum, torch.float32, "fsum", 2)
    _req(gpair, torch.float32, "gpair", 2)
    B, F, K = e.shape
    acc = 1
    if ge is None:
        ge = torch.empty_like(e)
        acc = 0
    else:
        _req(ge, torch.float32, "ge", 3)
    _call("lr_fm_pairwise_bwd_f32", _ptr(e), _ptr(fsum), _ptr(gpair), B, F, K, _ptr(ge),
                                             acc, _stream())
    return ge


def fm_embed_fwd(table: torch.Tensor, idx: torch.Tensor, want_e: bool = True,
                 lin: Optional[torch.Tensor] = None):
    """Fused gather + pairwise interaction (deepfm.py:155-163 over tfops/features.py:40).
    Returns (e, pair, fsum) or, with `lin` (the linear-weight table), (e, pair, fsum, lin_out)."""
    _req(table, torch.float32, "table", 2)
    _req(idx, torch.int32, "idx", 2)
    V, K = table.shape
    B, F = idx.shape
    dev = table.device
    e = torch.empty((B, F, K), dtype=torch.float32, device=dev) if want_e else None
    pair = torch.empty((B, K), dtype=torch.float32, device=dev)
    fsum = torch.empty((B, K), dtype=torch.float32, device=dev)
    lin_out = None
    if lin is not None:
        _req(lin, torch.float32, "lin")
        if lin.numel() != V:
            raise ValueError("lin must hold one weight per table row")
        lin_out = torch.empty((B, F), dtype=torch.float32, device=dev)
    _call("lr_fm_embed_fwd_f32", _ptr(table), _ptr(lin), V, K, _ptr(idx), B, F, _ptr(e), _ptr(pair),
          _ptr(fsum), _ptr(lin_out), _stream())
    return (e, pair, fsum) if lin is None else (e, pair, fsum, lin_out)


def fm_embed_bwd_adam(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                      gdeep: Optional[torch.Tensor], gpair: torch.Tensor, fsum: torch.Tensor,
                      B: int, F: int, seg: Segments, hp: AdamHP, lin=None, lin_m=None, lin_v=None,
                      glin: Optional[torch.Tensor] = None, bn_a: Optional[torch.Tensor] = None,
                      bn_c: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> None:
    _req(table, torch.float32, "table", 2)
    _req(m, torch.float32, "m", 2)
    _req(v, torch.float32, "v", 2)
    _req(gpair, torch.float32, "gpair", 2)
    _req(fsum, torch.float32, "fsum", 2)
    if glin is not None:          # the C ABI takes the linear gradient field-major [F,B]
        glin = glin.reshape(B, F).t().contiguous()
    for t_, n_ in ((gdeep, "gdeep"), (glin, "glin"), (bn_a, "bn_a"), (bn_c, "bn_c"), (lin, "lin"),
                   (lin_m, "lin_m"), (lin_v, "lin_v")):
        if t_ is not None:
            _req(t_, torch.float32, n_)
    V, K = table.shape
    if seg.n != B * F or seg.V != V:
        raise ValueError("segments were not built over idx[B*F] of this table")
    need = _lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=table.device)
    _call("lr_fm_embed_bwd_adam_f32", _ptr(table), _ptr(m), _ptr(v), _ptr(lin), _ptr(lin_m),
          _ptr(lin_v), V, K, _ptr(gdeep), _ptr(gpair), _ptr(fsum), _ptr(glin), _ptr(bn_a), _ptr(bn_c),
          B, F, _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg), hp, _ptr(ws),
          ws.numel(), _stream())


def fm_embed_bwd_rows(row_cache: torch.Tensor, gdeep: Optional[torch.Tensor], gpair: torch.Tensor,
                      fsum: torch.Tensor, B: int, F: int, seg: Segments,
                      glin: Optional[torch.Tensor] = None, bn_a: Optional[torch.Tensor] = None,
                      bn_c: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None):
    """Per-distinct-row gradients (run order) for row-sharded tables; see the header."""
    _req(row_cache, torch.float32, "row_cache", 2)
    _req(gpair, torch.float32, "gpair", 2)
    _req(fsum, torch.float32, "fsum", 2)
    U, K = row_cache.shape
    dev = row_cache.device
    grows = torch.empty((U, K), dtype=torch.float32, device=dev)
    if glin is not None:          # field-major [F,B] for the C ABI
        glin = glin.reshape(B, F).t().contiguous()
    glin_rows = torch.empty((U,), dtype=torch.float32, device=dev) if glin is not None else None
    need = _lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _call("lr_fm_embed_bwd_rows_f32", _ptr(row_cache), K, _ptr(gdeep), _ptr(gpair), _ptr(fsum),
          _ptr(glin), _ptr(bn_a), _ptr(bn_c), B, F, _ptr(seg.pos), _ptr(seg.start), _ptr(seg.n_seg),
          _ptr(grows), _ptr(glin_rows), _ptr(ws), ws.numel(), _stream())
    return grows, glin_rows


class AdamCoefBuffer:
    """Step-dependent Adam coefficients in device memory (see lr_adam_coef_store): what the `_dc`
    kernels of a hipGraph-captured training step read."""

    def __init__(self, device: torch.device):
        self.dev = torch.zeros(max(64, _lib.load().lr_adam_coef_bytes()), dtype=torch.uint8, device=device)

    def set(self, hp: AdamHP) -> None:
        """Enqueue (current stream) the write of step `hp.step`'s coefficients."""
        _call("lr_adam_coef_store", hp, _ptr(self.dev), _stream())


def adam_dense_dc(flat: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor, coef: AdamCoefBuffer) -> None:
    for t_, n_ in ((flat, "flat"), (m, "m"), (v, "v"), (grad, "grad")):
        _req(t_, torch.float32, n_)
    if not (flat.numel() == m.numel() == v.numel() == grad.numel()):
        raise ValueError("shape mismatch")
    _call("lr_adam_dense_dc_f32", _ptr(flat), _ptr(m), _ptr(v), flat.numel(), _ptr(grad), _ptr(coef.dev), _stream())


# --------------------------------------------------------------------------------------
# DeepFM: lookup fused with the first Dense layer (f32 MFMA) — see include/libreco_hip.h
# --------------------------------------------------------------------------------------
def deepfm_l1_supported(K: int, H1: int) -> bool:
    return bool(_lib.load().lr_deepfm_l1_supported(int(K), int(H1)))


# Arithmetic of the fused first layer's three contractions (csrc/deepfm_l1_sb.hip vs csrc/deepfm_l1.hip):
#   "split_bf16"  six-term split-bf16 MFMA products, f32 accumulation — the default where the shape is compiled (K = 64, H1 = 128);
#                 as close to f64 as the f32 chain (tests/test_l1_split_bf16_gpu.py), not bit-identical to it
#   "f32_chain"   v_mfma_f32_32x32x2_f32: the exact k-ordered f32 fma chain (every other shape; selectable everywhere)
# The packed-weight buffers carry the choice (uint8 planes vs float32 fragments): a net that allocated its buffers under one
# setting keeps it.  `LIBRECO_L1_ARITH` in the environment sets the initial value.
L1_ARITH = os.environ.get("LIBRECO_L1_ARITH", "split_bf16")
if L1_ARITH not in ("split_bf16", "f32_chain"):
    raise ValueError("LIBRECO_L1_ARITH must be split_bf16 or f32_chain")


def set_l1_arith(mode: str) -> str:
    """Select the arithmetic of first-layer buffers allocated from now on; returns the previous setting."""
    global L1_ARITH
    if mode not in ("split_bf16", "f32_chain"):
        raise ValueError("mode must be 'split_bf16' or 'f32_chain'")
    prev, L1_ARITH = L1_ARITH, mode
    return prev


def deepfm_l1_sb_supported(K: int, H1: int) -> bool:
    return bool(_lib.load().lr_deepfm_l1_sb_supported(int(K), int(H1)))


def deepfm_l1_use_sb(K: int, H1: int) -> bool:
    return L1_ARITH == "split_bf16" and deepfm_l1_sb_supported(K, H1)


def deepfm_l1_pack_bufs(F: int, K: int, H1: int, device, arith: Optional[str] = None):
    """Persistent (forward, row-gradient) packed-kernel buffers of a first layer [F*K, H1] under the current arithmetic."""
    sb = deepfm_l1_use_sb(K, H1) if arith is None else (arith == "split_bf16" and deepfm_l1_sb_supported(K, H1))
    if sb:
        n = _lib.load().lr_deepfm_l1_sb_pack_bytes(F, K, H1)
        return (torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device))
    return (torch.empty((F * K, H1), dtype=torch.float32, device=device),
            torch.empty((F * K, H1), dtype=torch.float32, device=device))


def _is_sb(buf: torch.Tensor) -> bool:
    return buf.dtype == torch.uint8


def deepfm_l1_pack(Wp: torch.Tensor, F: int, K: int, out=None, scale: Optional[torch.Tensor] = None, reduce=None):
    """First kernel Wp [F*K, H1] (optionally row-scaled: the BatchNorm fold, `diag(scale) Wp`) -> (WpA, WpB) in MFMA fragment
    order: f32 fragments for the f32 chain, three bf16 planes each (uint8 buffers) for the split-bf16 kernels — `out` from
    `deepfm_l1_pack_bufs` decides, the current `L1_ARITH` when `out` is None.  `reduce` = (partial [n, H1], out [H1]) with a
    scale: the launch also sums the folded bias's slab partials (no reduction launch of its own)."""
    red = (0, 0, 0)
    if reduce is not None:
        if scale is None:
            raise ValueError("`reduce` rides the scaled pack")
        rp, ro = reduce
        _req(rp, torch.float32, "reduce partial", 2)
        _req(ro, torch.float32, "reduce out", 1)
        if rp.shape[1] != Wp.shape[1] or ro.numel() != Wp.shape[1] or not rp.is_contiguous():
            raise ValueError("reduce = (partial [n, H1] contiguous, out [H1])")
        red = (_ptr(rp), rp.shape[0], _ptr(ro))
    _req(Wp, torch.float32, "Wp", 2)
    H1 = Wp.shape[1]
    if Wp.shape[0] != F * K:
        raise ValueError("Wp must be [F*K, H1]")
    if scale is not None:
        _req(scale, torch.float32, "scale", 1)
        if scale.numel() != F * K:
            raise ValueError("scale must be [F*K]")
    if out is None:
        out = deepfm_l1_pack_bufs(F, K, H1, Wp.device)
    if _is_sb(out[0]):
        n = _lib.load().lr_deepfm_l1_sb_pack_bytes(F, K, H1)
        if not deepfm_l1_sb_supported(K, H1) or out[0].numel() < n or out[1].numel() < n:
            raise ValueError(f"split-bf16 pack buffers do not fit K={K} H1={H1}")
        _call("lr_deepfm_l1_sb_pack", _ptr(Wp), _ptr(scale), F, K, H1, _ptr(out[0]), _ptr(out[1]), *red, _stream())
    elif scale is not None:
        _call("lr_deepfm_l1_pack_scaled_f32", _ptr(Wp), _ptr(scale), F, K, H1, _ptr(out[0]), _ptr(out[1]), *red, _stream())
    else:
        _call("lr_deepfm_l1_pack_f32", _ptr(Wp), F, K, H1, _ptr(out[0]), _ptr(out[1]), _stream())
    return out


_L1_WS: dict = {}     # persistent scratch of the split-bf16 kernels per (device, stream, purpose): addressed by captured graphs


def _l1_ws(device, key, nbytes: int) -> torch.Tensor:
    """One buffer per (device, stream, purpose, size), never released or replaced: a captured step keeps addressing it.  The
    STREAM is part of the key (round-5 advisor finding): launches on one stream are ordered and may share scratch, two nets whose
    captured steps replay on their own streams (`GraphRunner` has one per net) must not."""
    dev_i = device.index if device.index is not None else torch.cuda.current_device()
    k = (dev_i, int(torch.cuda.current_stream(dev_i).cuda_stream), key, int(nbytes))
    t = _L1_WS.get(k)
    if t is None:
        t = _L1_WS[k] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return t


def deepfm_l1_fwd(table: torch.Tensor, idx: torch.Tensor, WpA: torch.Tensor, bias: Optional[torch.Tensor],
                  H1: int, lin: Optional[torch.Tensor] = None, out=None):
    """(z1 [B,H1], pair [B,K], fsum [B,K], lin_out [B,F] | None).  `out` = (z1, pair, fsum) persistent buffers.
    `WpA` from `deepfm_l1_pack`: its kind selects the kernel (`lr_deepfm_l1_fwd_f32` / `lr_deepfm_l1_fwd_sb_f32`)."""
    _req(table, torch.float32, "table", 2)
    _req(idx, torch.int32, "idx", 2)
    V, K = table.shape
    B, F = idx.shape
    sb = _is_sb(WpA)
    if sb:
        if not deepfm_l1_sb_supported(K, H1) or WpA.numel() < _lib.load().lr_deepfm_l1_sb_pack_bytes(F, K, H1):
            raise ValueError("WpA must come from deepfm_l1_pack for this shape")
    else:
        _req(WpA, torch.float32, "WpA")
        if WpA.numel() != F * K * H1:
            raise ValueError("WpA has the wrong size")
    dev = table.device
    if out is not None:
        z1, pair, fsum = out
        if z1.shape != (B, H1) or pair.shape != (B, K) or fsum.shape != (B, K):
            raise ValueError("out buffers have the wrong shape")
    else:
        z1 = torch.empty((B, H1), dtype=torch.float32, device=dev)
        pair = torch.empty((B, K), dtype=torch.float32, device=dev)
        fsum = torch.empty((B, K), dtype=torch.float32, device=dev)
    lin_out = None
    if lin is not None:
        _req(lin, torch.float32, "lin")
        lin_out = torch.empty((B, F), dtype=torch.float32, device=dev)
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
    if sb:
        need = _lib.load().lr_deepfm_l1_fwd_sb_ws_bytes(B, F)
        ws = _l1_ws(dev, "fwd", need)
        _call("lr_deepfm_l1_fwd_sb_f32", _ptr(table), _ptr(lin), V, K, _ptr(idx), B, F, _ptr(WpA), _ptr(bias), H1,
              _ptr(z1), _ptr(pair), _ptr(fsum), _ptr(lin_out), _ptr(ws), ws.numel(), _stream())
    else:
        _call("lr_deepfm_l1_fwd_f32", _ptr(table), _ptr(lin), V, K, _ptr(idx), B, F, _ptr(WpA), _ptr(bias), H1,
              _ptr(z1), _ptr(pair), _ptr(fsum), _ptr(lin_out), _stream())
    return z1, pair, fsum, lin_out


def deepfm_l1_wgrad_chunks(B: int, F: int, K: int, H1: int, arith: Optional[str] = None) -> int:
    """Batch chunks of the weight-gradient kernel the current arithmetic would use for this shape."""
    sb = deepfm_l1_use_sb(K, H1) if arith is None else (arith == "split_bf16" and deepfm_l1_sb_supported(K, H1))
    return int(_lib.load().lr_deepfm_l1_wgrad_sb_chunks(B, F) if sb else _lib.load().lr_deepfm_l1_wgrad_chunks(B, F))


def deepfm_l1_wgrad(table: torch.Tensor, idxT: torch.Tensor, gz: torch.Tensor, n_chunks: Optional[int] = None,
                    out: Optional[torch.Tensor] = None, arith: Optional[str] = None) -> torch.Tensor:
    """partial [n_chunks, F*K, H1]; gather(table, idx)^T @ gz = partial.sum(0).  `arith` (default: `L1_ARITH`) selects
    `lr_deepfm_l1_wgrad_sb_f32` (gz split into bf16 planes first: `lr_deepfm_l1_sb_gz_pack`) or `lr_deepfm_l1_wgrad_f32`;
    an `out` buffer fixes the number of chunks (any value is valid for both kernels)."""
    _req(table, torch.float32, "table", 2)
    _req(idxT, torch.int32, "idxT", 2)
    _req(gz, torch.float32, "gz", 2)
    V, K = table.shape
    F, B = idxT.shape
    H1 = gz.shape[1]
    if gz.shape[0] != B:
        raise ValueError("gz must be [B, H1]")
    sb = deepfm_l1_use_sb(K, H1) if arith is None else (arith == "split_bf16" and deepfm_l1_sb_supported(K, H1))
    if out is not None:
        if out.numel() % (F * K * H1) != 0 or out.numel() == 0:
            raise ValueError("out has the wrong size")
        if n_chunks is None:
            n_chunks = out.numel() // (F * K * H1)
        elif out.numel() != n_chunks * F * K * H1:
            raise ValueError("out has the wrong size")
    if n_chunks is None:
        n_chunks = deepfm_l1_wgrad_chunks(B, F, K, H1, "split_bf16" if sb else "f32_chain")
    if out is None:
        out = torch.empty((n_chunks, F * K, H1), dtype=torch.float32, device=table.device)
    if B == 0:
        return out.zero_()
    if sb:
        gzp = _l1_ws(table.device, "gzp", _lib.load().lr_deepfm_l1_sb_gz_pack_bytes(B, H1))
        _call("lr_deepfm_l1_sb_gz_pack", _ptr(gz), B, H1, _ptr(gzp), _stream())
        _call("lr_deepfm_l1_wgrad_sb_f32", _ptr(table), V, K, _ptr(idxT), B, F, _ptr(gzp), H1, n_chunks, _ptr(out), _stream())
    else:
        _call("lr_deepfm_l1_wgrad_f32", _ptr(table), V, K, _ptr(idxT), B, F, _ptr(gz), H1, n_chunks, _ptr(out), _stream())
    return out


def deepfm_l1_dgrad(gz: torch.Tensor, WpB: torch.Tensor, K: int, F: int, slotT: torch.Tensor,
                    gl: Optional[torch.Tensor] = None, wp: Optional[torch.Tensor] = None,
                    fsum: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ge [B*F + 1, K]: per-position row gradients in run order; dropped positions land in the spare last row.
    `WpB` from `deepfm_l1_pack`: its kind selects the kernel (`lr_deepfm_l1_dgrad_f32` / `lr_deepfm_l1_dgrad_sb_f32`)."""
    _req(gz, torch.float32, "gz", 2)
    _req(slotT, torch.int32, "slotT", 2)
    B, H1 = gz.shape
    sb = _is_sb(WpB)
    if slotT.shape != (F, B):
        raise ValueError("shape mismatch")
    if sb:
        if not deepfm_l1_sb_supported(K, H1) or WpB.numel() < _lib.load().lr_deepfm_l1_sb_pack_bytes(F, K, H1):
            raise ValueError("WpB must come from deepfm_l1_pack for this shape")
    else:
        _req(WpB, torch.float32, "WpB")
        if WpB.numel() != F * K * H1:
            raise ValueError("shape mismatch")
    for t_, n_ in ((gl, "gl"), (wp, "wp"), (fsum, "fsum")):
        if t_ is not None:
            _req(t_, torch.float32, n_)
    if out is None:
        out = torch.empty((B * F + 1, K), dtype=torch.float32, device=gz.device)    # + the spare row of dropped positions
    elif out.numel() < (B * F + 1) * K:
        raise ValueError("ge must hold B*F + 1 rows")
    _call("lr_deepfm_l1_dgrad_sb_f32" if sb else "lr_deepfm_l1_dgrad_f32", _ptr(gz), H1, _ptr(WpB), K, F, B, _ptr(gl),
          _ptr(wp), _ptr(fsum), _ptr(slotT), _ptr(out), _stream())
    return out


def fm_rows_adam(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, ge: torch.Tensor, seg: Segments,
                 hp, B: int, F: int, gl=None, wp=None, lin=None, lin_m=None, lin_v=None, bn_a=None,
                 bn_c=None, lin_scale=None, ws: Optional[torch.Tensor] = None) -> None:
    """Adam over run-ordered per-position gradients (see lr_fm_rows_adam_f32).  `hp`: an `AdamHP`
    (by value) or an `AdamCoefBuffer` (device-resident coefficients, graph-capturable)."""
    _req(table, torch.float32, "table", 2)
    _req(m, torch.float32, "m", 2)
    _req(v, torch.float32, "v", 2)
    _req(ge, torch.float32, "ge", 2)
    for t_, n_ in ((gl, "gl"), (wp, "wp"), (lin, "lin"), (lin_m, "lin_m"), (lin_v, "lin_v"), (bn_a, "bn_a"),
                   (bn_c, "bn_c"), (lin_scale, "lin_scale")):
        if t_ is not None:
            _req(t_, torch.float32, n_)
    V, K = table.shape
    if seg.n != B * F or seg.V != V or ge.dim() != 2 or ge.shape[1] != K or ge.shape[0] < B * F:
        raise ValueError("segments / ge were not built over idx[B*F] of this table")
    need = _lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=table.device)
    dc = isinstance(hp, AdamCoefBuffer)
    _call("lr_fm_rows_adam_dc_f32" if dc else "lr_fm_rows_adam_f32", _ptr(table), _ptr(m), _ptr(v), _ptr(lin),
          _ptr(lin_m), _ptr(lin_v), V, K, _ptr(ge), _ptr(gl), _ptr(wp), _ptr(bn_a), _ptr(bn_c), _ptr(lin_scale), B, F,
          _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg), _ptr(hp.dev) if dc else hp, _ptr(ws),
          ws.numel(), _stream())


def fm_rows_grad(cache: torch.Tensor, lin_cache: Optional[torch.Tensor], ge: torch.Tensor, seg: Segments, B: int,
                 F: int, slots: torch.Tensor, gl, wp, bn_a=None, bn_c=None, lin_scale=None,
                 ws: Optional[torch.Tensor] = None):
    """Row-sharded tables: per-row gradients of the run-ordered per-position gradients `ge` (the arithmetic of
    `fm_rows_adam` without the update), written at the rows' cache slots: returns (grows [n_cache, K],
    glin_rows [n_cache] or None).  `seg`: per-field runs of the GLOBAL ids, `slots` [B*F]: position -> cache row."""
    _req(cache, torch.float32, "cache", 2)
    _req(ge, torch.float32, "ge", 2)
    _req(slots, torch.int32, "slots")
    n_cache, K = cache.shape
    if seg.n != B * F or slots.numel() != B * F or ge.shape[1] != K or ge.shape[0] < B * F:
        raise ValueError("segments / slots / ge were not built over idx[B*F]")
    grows = torch.empty((n_cache, K), dtype=torch.float32, device=cache.device)
    glin = torch.empty(n_cache, dtype=torch.float32, device=cache.device) if lin_cache is not None else None
    need = _lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=cache.device)
    _call("lr_fm_rows_grad_f32", _ptr(cache), _ptr(lin_cache), n_cache, K, _ptr(ge), _ptr(gl), _ptr(wp), _ptr(bn_a),
          _ptr(bn_c), _ptr(lin_scale), B, F, _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg),
          _ptr(slots), _ptr(grows), _ptr(glin), _ptr(ws), ws.numel(), _stream())
    return grows, glin


def fm_rows_grad_compact(table: torch.Tensor, lin: Optional[torch.Tensor], ge: torch.Tensor, seg: Segments, B: int, F: int,
                         gl, wp, bn_a=None, bn_c=None, lin_scale=None, ws: Optional[torch.Tensor] = None, out=None):
    """The per-row gradients of `fm_rows_adam` WITHOUT the update, rows read from the tables themselves, run s writing
    grows[s] / glin_rows[s]: the operand of the dense (TF1) table pass `adam_dense_rows`.  `out`: persistent
    (grows [>= B*F, K], glin_rows [>= B*F] or None) buffers of a captured step."""
    _req(table, torch.float32, "table", 2)
    _req(ge, torch.float32, "ge", 2)
    V, K = table.shape
    if seg.n != B * F or seg.V != V or ge.shape[1] != K or ge.shape[0] < B * F:
        raise ValueError("segments / ge were not built over idx[B*F] of this table")
    if out is None:
        out = (torch.empty((B * F, K), dtype=torch.float32, device=table.device),
               torch.empty(B * F, dtype=torch.float32, device=table.device) if lin is not None else None)
    grows, glin = out
    if grows.shape[0] < B * F or grows.shape[1] != K or (glin is None) != (lin is None) or (glin is not None and glin.numel() < B * F):
        raise ValueError("`out` must hold one row per position")
    need = _lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=table.device)
    _call("lr_fm_rows_grad_compact_f32", _ptr(table), _ptr(lin), V, K, _ptr(ge), _ptr(gl), _ptr(wp), _ptr(bn_a), _ptr(bn_c),
          _ptr(lin_scale), B, F, _ptr(seg.pos), _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg), _ptr(grows), _ptr(glin),
          _ptr(ws), ws.numel(), _stream())
    return grows, glin


def adam_dense_rows(table, m, v, hp, grows, seg: Segments, row_slot: torch.Tensor, lin=None, lin_m=None, lin_v=None,
                    glin_rows=None) -> None:
    """TF1's dense Adam over EVERY row of `table` [V, K] (and of `lin` [V]) in one streaming launch; the rows of
    `seg.rows[:n_seg]` take `grows[s]` / `glin_rows[s]`.  `row_slot`: int32 [V], all -1 (restored on return).
    `hp`: an `AdamHP` (by value) or an `AdamCoefBuffer` (graph-capturable)."""
    for t_, n_ in ((table, "table"), (m, "m"), (v, "v"), (grows, "grows")):
        _req(t_, torch.float32, n_, 2)
    _req(row_slot, torch.int32, "row_slot", 1)
    V, K = table.shape
    if row_slot.numel() != V or seg.V != V or grows.shape[1] != K or grows.shape[0] < seg.n:
        raise ValueError("row_slot / segments / grows do not belong to this table")
    dc = isinstance(hp, AdamCoefBuffer)
    _call("lr_adam_dense_rows_dc_f32" if dc else "lr_adam_dense_rows_f32", _ptr(table), _ptr(m), _ptr(v), _ptr(lin), _ptr(lin_m),
          _ptr(lin_v), V, K, _ptr(grows), _ptr(glin_rows), _ptr(seg.rows), _ptr(seg.n_seg), seg.n, _ptr(row_slot),
          _ptr(hp.dev) if dc else hp, _stream())


# --------------------------------------------------------------------------------------
# full-catalog scoring + top-k
# --------------------------------------------------------------------------------------
# Arithmetic of the score contraction: "f32_chain" = the exact k-ordered f32 fma chain on the f32 MFMA pipe; "split_bf16" = six
# bf16 MFMA products per f32 product with f32 accumulation (as close to fp64 as the chain, not bit-identical to it; the item
# planes are split on the fly).  An explicit argument of every call (no library-side state); `LIBRECO_TOPK_ARITH` only sets the
# default the Python callers pass.
# "filter" (the default) / "filter_f32_chain": lr_score_topk_filter_f32 — a one-product bf16 pass keeps k' > k candidates per
# user (ranked by an upper bound of the exact score), f32 scores of those, a PROOF per user that nothing outside the k' can be in
# the top k (its k-th exact score exceeds the k'-th bound), and the exact kernel (split-bf16 /
# f32 chain) for the users without a proof and for the shapes the filter does not take (below 2^20 items, k > 100, reduction
# widths outside 33..128).  44.5 ms per 1,024 users x 100 M items pass against 144 ms (split_bf16) and 209 ms (f32_chain); ids
# equal to the fp64 ranking wherever fp64 scores are separated by more than f32 rounding (tests/test_fullsize_parity_gpu.py, all
# three forms).  TOPK_FILTER_FORCE takes the filter below 2^20 items too (tests).
_TOPK_ARITHS = ("split_bf16", "f32_chain", "filter", "filter_f32_chain")
TOPK_ARITH = os.environ.get("LIBRECO_TOPK_ARITH", "filter")
if TOPK_ARITH not in _TOPK_ARITHS:
    raise ValueError(f"LIBRECO_TOPK_ARITH must be one of {_TOPK_ARITHS}")
TOPK_FILTER_FORCE = False


def score_topk(users: torch.Tensor, items: torch.Tensor, k: int,
               consumed_ptr: Optional[torch.Tensor] = None,
               consumed_idx: Optional[torch.Tensor] = None,
               filter_flag: Optional[torch.Tensor] = None, item_base: int = 0,
               ws: Optional[torch.Tensor] = None, arith: Optional[str] = None,
               failed_out: Optional[torch.Tensor] = None):
    """``users @ items.T`` + per-user top-k (recommendation/recommend.py:66-68 + ranking.py).  `failed_out` ([B] uint8, the
    filtered forms only): 1 where a user was not certified by the filter and was ranked by the exact kernel."""
    arith = TOPK_ARITH if arith is None else arith
    if arith not in _TOPK_ARITHS:
        raise ValueError(f"arith must be one of {_TOPK_ARITHS}")
    filt = arith.startswith("filter")
    _req(users, torch.float32, "users", 2)
    _req(items, torch.float32, "items", 2)
    B, D = users.shape
    N = items.shape[0]
    if items.shape[1] != D:
        raise ValueError("users and items must share the embedding width")
    if k > N:
        raise ValueError(f"`n_rec` {k} exceeds num of items {N}")
    if D % 4 != 0:  # the MFMA path needs 16-byte rows: pad the (tiny) reduction dim with zeros
        pad = 4 - D % 4
        users = torch.nn.functional.pad(users, (0, pad)).contiguous()
        items = torch.nn.functional.pad(items, (0, pad)).contiguous()
        D += pad
    lib = _lib.load()
    need = (lib.lr_score_topk_filter_ws_bytes if filt else lib.lr_score_topk_ws_bytes)(B, N, D, k)
    if need == 0 and B > 0 and N > 0:
        raise ValueError(f"unsupported score_topk shape B={B} N={N} D={D} k={k}")
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 8), dtype=torch.uint8, device=users.device)
    if consumed_ptr is not None:
        _req(consumed_ptr, torch.int64, "consumed_ptr", 1)
        _req(consumed_idx, torch.int32, "consumed_idx", 1)
    if filter_flag is not None:
        _req(filter_flag, torch.uint8, "filter_flag", 1)
    out_s = torch.empty((B, k), dtype=torch.float32, device=users.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=users.device)
    if filt:
        if failed_out is not None:
            _req(failed_out, torch.uint8, "failed_out", 1)
            if failed_out.numel() < B:
                raise ValueError("failed_out holds one byte per user")
        _call("lr_score_topk_filter_f32", _ptr(users), B, _ptr(items), N, D, _ptr(consumed_ptr), _ptr(consumed_idx),
              _ptr(filter_flag), k, item_base, _ptr(out_s), _ptr(out_i), _ptr(ws), ws.numel(),
              0 if arith == "filter_f32_chain" else 1, 1 if TOPK_FILTER_FORCE else 0, _ptr(failed_out), _stream())
        return out_s, out_i
    if failed_out is not None:
        raise ValueError("failed_out belongs to the filtered forms")
    _call("lr_score_topk_sb_f32" if arith == "split_bf16" else "lr_score_topk_f32", _ptr(users), B, _ptr(items), N, D, _ptr(consumed_ptr),
                                _ptr(consumed_idx), _ptr(filter_flag), k, item_base, _ptr(out_s),
                                _ptr(out_i), _ptr(ws), ws.numel(), _stream())
    return out_s, out_i


# --------------------------------------------------------------------------------------
# streaming in-batch softmax cross-entropy (two-tower retrieval loss)
# --------------------------------------------------------------------------------------
# Arithmetic of the two contractions of the streaming softmax cross-entropy: "split_bf16" (default) forms every f32 product as
# six bf16 MFMA products with f32 accumulation (error against f64 = that of the f32 fma chain: tests/test_softmax_ce_gpu.py runs
# every case under both); "f32_chain" is the exact f32 MFMA chain.  `LIBRECO_SCE_ARITH` sets the initial value.
SCE_ARITH = os.environ.get("LIBRECO_SCE_ARITH", "split_bf16")
if SCE_ARITH not in ("split_bf16", "f32_chain"):
    raise ValueError("LIBRECO_SCE_ARITH must be split_bf16 or f32_chain")


def set_sce_arith(mode: str) -> str:
    """Select the arithmetic of `softmax_ce_fwd` / `softmax_ce_bwd_cols` from now on; returns the previous setting."""
    global SCE_ARITH
    if mode not in ("split_bf16", "f32_chain"):
        raise ValueError("mode must be 'split_bf16' or 'f32_chain'")
    prev, SCE_ARITH = SCE_ARITH, mode
    return prev


def _sce_arith(arith: Optional[str] = None) -> int:
    """The arithmetic as the C ABI takes it (an explicit argument of the workspace query and of both launches: the library keeps
    no setting of its own, so callers with different settings cannot race) — `arith` or the module default `SCE_ARITH`."""
    arith = SCE_ARITH if arith is None else arith
    if arith not in ("split_bf16", "f32_chain"):
        raise ValueError("arith must be 'split_bf16' or 'f32_chain'")
    return 1 if arith == "split_bf16" else 0


def softmax_ce_supported(B: int, N: int, D: int) -> bool:
    return bool(_lib.load().lr_softmax_ce_supported(B, N, D))


def _sce_check(X, Y, col_bias, row_ids, col_ids, pos0):
    _req(X, torch.float32, "X", 2)
    _req(Y, torch.float32, "Y", 2)
    B, D = X.shape
    N = Y.shape[0]
    if Y.shape[1] != D:
        raise ValueError("X and Y must share the embedding width")
    if col_bias is not None:
        _req(col_bias, torch.float32, "col_bias", 1)
        if col_bias.numel() != N:
            raise ValueError("col_bias must hold one value per column")
    if (row_ids is None) != (col_ids is None):
        raise ValueError("row_ids and col_ids come together")
    if row_ids is not None:
        _req(row_ids, torch.int32, "row_ids", 1)
        _req(col_ids, torch.int32, "col_ids", 1)
        if row_ids.numel() != B or col_ids.numel() != N:
            raise ValueError("row_ids / col_ids must hold one id per row / column")
    if pos0 < 0 or pos0 + B > N:
        raise ValueError("the positives [pos0, pos0 + B) must be columns of Y")
    return B, N, D


def softmax_ce_fwd(X, Y, col_bias=None, row_ids=None, col_ids=None, pos0: int = 0, want_w: bool = True, arith: Optional[str] = None):
    """`softmax_cross_entropy` (tfops/loss.py:71-75) of logits = X @ Y.T + col_bias with the accidental-hit
    mask of `adjust_logits` (two_tower.py:458-479), without the B x N matrix: returns (lse, pos_logit, W)
    with W[i] = softmax(logits[i]) @ Y (None unless `want_w`)."""
    B, N, D = _sce_check(X, Y, col_bias, row_ids, col_ids, pos0)
    lse = torch.empty(B, dtype=torch.float32, device=X.device)
    pos = torch.empty(B, dtype=torch.float32, device=X.device)
    W = torch.empty((B, D), dtype=torch.float32, device=X.device) if want_w else None
    ar = _sce_arith(arith)
    need = _lib.load().lr_softmax_ce_fwd_ws_bytes(B, N, D, ar)
    ws = torch.empty(need, dtype=torch.uint8, device=X.device) if need else None
    _call("lr_softmax_ce_fwd_f32", _ptr(X), B, _ptr(Y), N, D, _ptr(col_bias), _ptr(row_ids), _ptr(col_ids), pos0,
          _ptr(lse), _ptr(pos), _ptr(W), _ptr(ws), need, ar, _stream())
    return lse, pos, W


def softmax_ce_bwd_cols(X, Y, lse, g, col_bias=None, row_ids=None, col_ids=None, pos0: int = 0, arith: Optional[str] = None):
    """V[j] = sum_i g[i] softmax(logits[i])[j] X[i] — the column-side gradient of `softmax_ce_fwd`'s loss
    (without the positives' -g[i] X[i])."""
    B, N, D = _sce_check(X, Y, col_bias, row_ids, col_ids, pos0)
    _req(lse, torch.float32, "lse", 1)
    _req(g, torch.float32, "g", 1)
    V = torch.empty((N, D), dtype=torch.float32, device=X.device)
    _call("lr_softmax_ce_bwd_cols_f32", _ptr(X), B, _ptr(Y), N, D, _ptr(col_bias), _ptr(row_ids), _ptr(col_ids), pos0,
          _ptr(lse), _ptr(g), _ptr(V), _sce_arith(arith), _stream())
    return V


class _SoftmaxCE(torch.autograd.Function):
    """loss_i = logsumexp_j logits[i][:] - logits[i][pos0 + i] on the streaming kernels; gradients w.r.t.
    X and Y (col_bias and the ids are constants of the reference's graph as well)."""

    @staticmethod
    def forward(ctx, X, Y, col_bias, row_ids, col_ids, pos0):
        X, Y = X.contiguous(), Y.contiguous()
        need = X.requires_grad or Y.requires_grad
        lse, pos, W = softmax_ce_fwd(X, Y, col_bias, row_ids, col_ids, pos0, want_w=need)
        ctx.save_for_backward(X, Y, lse, W if W is not None else lse)
        ctx.misc = (col_bias, row_ids, col_ids, pos0)
        return lse - pos

    @staticmethod
    def backward(ctx, g):
        X, Y, lse, W = ctx.saved_tensors
        col_bias, row_ids, col_ids, pos0 = ctx.misc
        g = g.contiguous()
        B = X.shape[0]
        dX = g[:, None] * (W - Y[pos0:pos0 + B])
        dY = softmax_ce_bwd_cols(X, Y, lse, g, col_bias, row_ids, col_ids, pos0)
        dY[pos0:pos0 + B] -= g[:, None] * X
        return dX, dY, None, None, None, None


def softmax_ce(X, Y, col_bias=None, row_ids=None, col_ids=None, pos0: int = 0) -> torch.Tensor:
    """Per-row in-batch softmax cross-entropy (differentiable w.r.t. X and Y), D <= 128 and D % 4 == 0."""
    return _SoftmaxCE.apply(X, Y, col_bias, row_ids, col_ids, pos0)


def pair_mlp_supported(H1: int, H2: int) -> bool:
    return bool(_lib.load().lr_pair_mlp_supported(H1, H2))


# Arithmetic of the pair MLP's H1 x H2 product: "split_bf16" (six bf16 MFMA products per f32 product, f32 accumulation; the default)
# or "f32_chain" — an explicit choice per call, `LIBRECO_PAIR_MLP_ARITH` only sets the default the Python callers pass.
PAIR_MLP_ARITH = os.environ.get("LIBRECO_PAIR_MLP_ARITH", "split_bf16")
if PAIR_MLP_ARITH not in ("split_bf16", "f32_chain"):
    raise ValueError("LIBRECO_PAIR_MLP_ARITH must be split_bf16 or f32_chain")


def pair_mlp(P: torch.Tensor, Q: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor, v3: torch.Tensor, c3: float,
             out: torch.Tensor, accumulate: bool = True, arith: Optional[str] = None) -> torch.Tensor:
    """out[u, i] (+)= relu(relu(P[u] + Q[i]) @ W2 + b2) @ v3 + c3 — the MLP tail of every (user, item) pair of a
    DeepFM catalogue ranking (see lr_pair_mlp_f32 / lr_pair_mlp_sb_f32); `out` [B, N] may be a column slice of a wider matrix."""
    arith = PAIR_MLP_ARITH if arith is None else arith
    if arith not in ("split_bf16", "f32_chain"):
        raise ValueError("arith must be 'split_bf16' or 'f32_chain'")
    for t_, n_ in ((P, "P"), (Q, "Q"), (W2, "W2"), (b2, "b2"), (v3, "v3")):
        _req(t_.contiguous(), torch.float32, n_)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.dim() == 2):
        raise RuntimeError("out must be a 2-D float32 tensor on the MI355X device (there is no CPU fallback)")
    B, H1 = P.shape
    N, H2 = Q.shape[0], W2.shape[1]
    if Q.shape[1] != H1 or W2.shape[0] != H1 or b2.numel() != H2 or v3.numel() != H2 or out.shape != (B, N) or out.stride(1) != 1:
        raise ValueError("shape mismatch")
    _call("lr_pair_mlp_sb_f32" if arith == "split_bf16" else "lr_pair_mlp_f32", _ptr(P.contiguous()), B, _ptr(Q.contiguous()), N, H1,
          _ptr(W2.contiguous()), _ptr(b2.contiguous()), H2,
          _ptr(v3.contiguous()), float(c3), _ptr(out), out.stride(0), 1 if accumulate else 0, _stream())
    return out


def topk_merge(scores: torch.Tensor, ids: torch.Tensor):
    """Merge per-shard ``[S,B,k]`` candidates into the global top-k."""
    _req(scores, torch.float32, "scores", 3)
    _req(ids, torch.int64, "ids", 3)
    S, B, k = scores.shape
    out_s = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    _call("lr_topk_merge_f32", _ptr(scores), _ptr(ids), S, B, k, _ptr(out_s), _ptr(out_i),
                                        _stream())
    return out_s, out_i


# --------------------------------------------------------------------------------------
# SpMM
# --------------------------------------------------------------------------------------
class SpmmPlan:
    """Workspace of the degree-bucketed SpMM bound to ONE graph: the chunk lists of the long rows depend on `rowptr` alone,
    so they are built by the first product and reused by every later one (LightGCN: six products per step on a static
    graph, `lightgcn_module.py:66-88`; the classification pre-pass re-ran on each of them before round 4)."""

    def __init__(self, rowptr: torch.Tensor, nnz: int, K: int):
        self.rowptr_ptr, self.rows, self.nnz, self.K = rowptr.data_ptr(), rowptr.numel() - 1, int(nnz), int(K)
        need = _lib.load().lr_spmm_csr_ws_bytes(self.rows, self.nnz, self.K)
        self.ws = torch.empty(max(need, 256), dtype=torch.uint8, device=rowptr.device)
        self.ready = False

    def matches(self, rowptr, nnz, K) -> bool:
        return rowptr.data_ptr() == self.rowptr_ptr and rowptr.numel() - 1 == self.rows and nnz == self.nnz and K == self.K


def row_slots(seg: Segments, row_slot: torch.Tensor, set: bool = True) -> None:
    """row_slot[seg.rows[s]] = s for the segments of `seg` (`set`), or -1 again."""
    _req(row_slot, torch.int32, "row_slot", 1)
    if row_slot.numel() != seg.V:
        raise ValueError("row_slot must hold one entry per table row")
    _call("lr_row_slots_i32", _ptr(seg.rows), _ptr(seg.n_seg), seg.n, _ptr(row_slot), 1 if set else 0, _stream())


def spmm_csr_adam(rowptr, col, val, X, w, m, v, hp: AdamHP, plan: "SpmmPlan", vmax=None, row_slot=None, gsum=None,
                  alpha: float = 1.0) -> None:
    """One Adam step of (w, m, v[, vmax]) with the gradient g = A X (+ alpha * gsum[row_slot[r]] on the rows with a slot), the
    product's rows never written (see lr_spmm_csr_adam_f32).  Raises ValueError for widths the bucketed kernels do not take."""
    _req(rowptr, torch.int64, "rowptr", 1)
    _req(col, torch.int32, "col", 1)
    _req(val, torch.float32, "val", 1)
    for t_, n_ in ((X, "X"), (w, "w"), (m, "m"), (v, "v")):
        _req(t_, torch.float32, n_, 2)
    rows, K, nnz = rowptr.numel() - 1, X.shape[1], col.numel()
    if not plan.matches(rowptr, nnz, K) or w.shape != (rows, K) or m.shape != w.shape or v.shape != w.shape:
        raise ValueError("plan / parameter shapes do not belong to this graph")
    if (row_slot is None) != (gsum is None):
        raise ValueError("row_slot and gsum come together")
    if row_slot is not None:
        _req(row_slot, torch.int32, "row_slot", 1)
        _req(gsum, torch.float32, "gsum", 2)
    _call("lr_spmm_csr_adam_f32", _ptr(rowptr), _ptr(col), _ptr(val), rows, nnz, _ptr(X), K, _ptr(w), _ptr(m), _ptr(v),
          _ptr(vmax), _ptr(row_slot), _ptr(gsum), float(alpha), hp, _ptr(plan.ws), plan.ws.numel(), 1 if plan.ready else 0, _stream())
    plan.ready = True


class RowBitmap:
    """One bit per row (uint32 words) marking the rows of a batch: `set(ids)` / `clear(ids)` with the same id list."""

    def __init__(self, n_rows: int, device):
        self.n = int(n_rows)
        self.words = torch.zeros((self.n + 31) // 32 + 1, dtype=torch.int32, device=device)
        self._marked = False

    def set(self, ids: torch.Tensor) -> "RowBitmap":
        _req(ids, torch.int32, "ids", 1)
        if self._marked:            # a step that raised between `set` and `clear` left bits behind: start from an empty map
            self.words.zero_()
        self._marked = True
        _call("lr_bitmap_ids_i32", _ptr(ids), ids.numel(), self.n, _ptr(self.words), 1, _stream())
        return self

    def clear(self, ids: torch.Tensor) -> None:
        _req(ids, torch.int32, "ids", 1)
        _call("lr_bitmap_ids_i32", _ptr(ids), ids.numel(), self.n, _ptr(self.words), 0, _stream())
        self._marked = False


def spmm_csr(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, X: torch.Tensor,
             out: Optional[torch.Tensor] = None, acc: Optional[torch.Tensor] = None,
             plan: Optional[SpmmPlan] = None, x_rows: Optional[RowBitmap] = None,
             y_rows: Optional[RowBitmap] = None, acc_only: bool = False) -> Optional[torch.Tensor]:
    """Y = A X.  `x_rows`: the rows of X outside the bitmap are zero and are not read; `y_rows`: only the rows of Y inside the
    bitmap are computed (the others keep what `out` held) — both need a `plan` and a compiled width (K in 16 / 32 / 64 / 128).
    `acc_only`: `acc += A X`, the product itself is not stored (returns None)."""
    _req(rowptr, torch.int64, "rowptr", 1)
    _req(col, torch.int32, "col", 1)
    _req(val, torch.float32, "val", 1)
    _req(X, torch.float32, "X", 2)
    rows = rowptr.numel() - 1
    K = X.shape[1]
    nnz = col.numel()
    if acc_only:
        if acc is None or out is not None:
            raise ValueError("`acc_only` adds the product to `acc` and stores nothing else: pass `acc`, no `out`")
    elif out is None:
        if y_rows is not None:
            raise ValueError("`y_rows` writes some rows only: pass the `out` buffer that holds the others")
        out = torch.empty((rows, K), dtype=torch.float32, device=X.device)
    if x_rows is not None or y_rows is not None:
        if plan is None or not plan.matches(rowptr, nnz, K):
            raise ValueError("row bitmaps need the graph's SpmmPlan")
        if (x_rows is not None and x_rows.n != X.shape[0]) or (y_rows is not None and y_rows.n != rows):
            raise ValueError("bitmap sizes must match the rows of X / Y")
        _call("lr_spmm_csr_masked_f32", _ptr(rowptr), _ptr(col), _ptr(val), rows, nnz, _ptr(X), K, _ptr(out), _ptr(acc),
              _ptr(x_rows.words) if x_rows is not None else 0, _ptr(y_rows.words) if y_rows is not None else 0,
              _ptr(plan.ws), plan.ws.numel(), 1 if plan.ready else 0, _stream())
        plan.ready = True
        return out
    if plan is not None:
        if not plan.matches(rowptr, nnz, K):
            raise ValueError("the SpmmPlan was made for another graph (rowptr / nnz / K differ)")
        _call("lr_spmm_csr_bucketed_f32", _ptr(rowptr), _ptr(col), _ptr(val), rows, nnz, _ptr(X), K,
              _ptr(out), _ptr(acc), _ptr(plan.ws), plan.ws.numel(), 1 if plan.ready else 0, _stream())
        plan.ready = True
        return out
    need = _lib.load().lr_spmm_csr_ws_bytes(rows, nnz, K)
    key = (X.device, "spmm")
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < need:
        ws = _WS_CACHE[key] = torch.empty(max(need, 256), dtype=torch.uint8, device=X.device)
    _call("lr_spmm_csr_bucketed_f32", _ptr(rowptr), _ptr(col), _ptr(val), rows, nnz, _ptr(X), K,
          _ptr(out), _ptr(acc), _ptr(ws), ws.numel(), 0, _stream())
    return out


_WS_CACHE = {}      # per-device scratch of the degree-bucketed SpMM (chunk lists + chunk sums)


# --------------------------------------------------------------------------------------
# DIN attention pooling
# --------------------------------------------------------------------------------------
def _din_params(W1, b1, W2, b2, K):
    _req(W1, torch.float32, "W1", 2)
    _req(b1, torch.float32, "b1", 1)
    _req(W2, torch.float32, "W2")
    _req(b2, torch.float32, "b2")
    H = W1.shape[1]
    if W1.shape[0] != 4 * K or b1.numel() != H or W2.numel() != H or b2.numel() != 1:
        raise ValueError("attention MLP must be [4K,H] -> [H] -> 1 (layers/attention.py:50-56)")
    return H


def din_attn_pool_fwd(item_table: torch.Tensor, item: torch.Tensor, seq: torch.Tensor,
                      seq_len: torch.Tensor, W1, b1, W2, b2, out=None, attn=None, hid=None, order_out=None):
    """Fused gather + din_attention (algorithms/din.py:241-250, layers/attention.py:28-64).  `out` [B, K] / `attn`
    [B, L]: persistent output buffers (e.g. the attention plane of the MLP-input block).  `hid` [B * L, 16] (optional): the
    hidden activations of the attention MLP are kept there for `din_attn_pool_bwd(..., hid=hid)`.  `order_out` int32 [B]
    (optional): the launch also writes the balanced walk order of the backward kernels (`din_attn_pool_bwd(..., order=...)`)."""
    _req(item_table, torch.float32, "item_table", 2)
    _req(item, torch.int32, "item", 1)
    _req(seq, torch.int32, "seq", 2)
    _req(seq_len, torch.int32, "seq_len", 1)
    V, K = item_table.shape
    B, L = seq.shape
    H = _din_params(W1, b1, W2, b2, K)
    if out is None:
        out = torch.empty((B, K), dtype=torch.float32, device=item_table.device)
    if attn is None:
        attn = torch.empty((B, L), dtype=torch.float32, device=item_table.device)
    if out.shape != (B, K) or attn.shape != (B, L) or not out.is_contiguous() or not attn.is_contiguous():
        raise ValueError("out / attn must be contiguous [B, K] / [B, L] tensors")
    _call("lr_din_attn_pool_fwd_f32", _ptr(item_table), V, K, _ptr(item), _ptr(seq),
                                               _ptr(seq_len), B, L, _ptr(W1), _ptr(b1), _ptr(W2),
                                               _ptr(b2), H, _ptr(out), _ptr(attn),
                                               _ptr(_din_hid(hid, B, L, K, order_out is not None)),
                                               _ptr(_din_order(order_out, B)), _stream())
    return out, attn


def din_hid_floats(B: int, L: int, K: int, with_order: bool = True) -> int:
    """Elements of the `hid` buffer of `din_attn_pool_fwd / _bwd`: the hidden activations [B * L, 16] and, when the balanced
    order is used too, the backward's transposed weight images behind them (see include/libreco_hip.h)."""
    return B * L * 16 + (3 * (K // 16) * 256 if with_order else 0)


def _din_hid(hid, B, L, K=0, with_order=False):
    need = din_hid_floats(B, L, K, with_order)
    if hid is not None and (hid.dtype != torch.float32 or not hid.is_contiguous() or hid.numel() < need):
        raise ValueError(f"hid must be a contiguous float32 buffer of at least {need} elements (ops.din_hid_floats)")
    return hid


def din_attn_pool_bwd(item_table, item, seq, seq_len, W1, b1, W2, b2, attn, gout, gq_out=None, gkey_out=None,
                      param_out=None, ws=None, parts=3, keep_pad_rows=False, hid=None, order=None):
    """`parts`: 1 = data half (gq, gkey), 2 = parameter half (needs the data half's spill in the same `ws`), 3 = both;
    `keep_pad_rows`: leave gkey rows past a sample's length untouched (caller drops those positions).
    `gq_out` [B, K] / `gkey_out` [B, L, K] (contiguous): write the query / key gradients there — e.g. straight into
    the combined gradient buffer of the table update instead of concatenating 210 MB afterwards.  `param_out` =
    (gW1, gb1, gW2, gb2) contiguous tensors shaped like the parameters (e.g. the `.grad` views of `DenseParams`);
    `ws`: persistent workspace of at least `lr_din_attn_ws_bytes` bytes."""
    _req(item_table, torch.float32, "item_table", 2)
    _req(attn, torch.float32, "attn", 2)
    _req(gout, torch.float32, "gout", 2)
    V, K = item_table.shape
    B, L = seq.shape
    H = _din_params(W1, b1, W2, b2, K)
    dev = item_table.device
    lib = _lib.load()
    need = max(lib.lr_din_attn_ws_bytes(B, L, K, H), 8)
    if parts != 3 and (ws is None or ws.numel() < need or gkey_out is None or gq_out is None):
        raise ValueError("a split backward needs the caller's persistent `ws` and output buffers")
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    gq = torch.empty((B, K), dtype=torch.float32, device=dev) if gq_out is None else gq_out
    gkey = torch.empty((B, L, K), dtype=torch.float32, device=dev) if gkey_out is None else gkey_out
    if gq.shape != (B, K) or gkey.shape != (B, L, K) or not gq.is_contiguous() or not gkey.is_contiguous():
        raise ValueError("gq_out / gkey_out must be contiguous [B, K] / [B, L, K] tensors")
    if param_out is not None:
        gW1, gb1, gW2, gb2 = param_out
        for g_, p_ in ((gW1, W1), (gb1, b1), (gW2, W2), (gb2, b2)):
            if g_.numel() != p_.numel() or not g_.is_contiguous() or g_.dtype != torch.float32:
                raise ValueError("param_out tensors must be contiguous fp32 tensors shaped like the parameters")
    else:
        gW1, gb1 = torch.empty_like(W1), torch.empty_like(b1)
        gW2, gb2 = torch.empty_like(W2), torch.empty_like(b2)
    _call("lr_din_attn_pool_bwd_parts_f32", _ptr(item_table), V, K, _ptr(item), _ptr(seq),
          _ptr(seq_len), B, L, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
          H, _ptr(attn), _ptr(gout), _ptr(gq), _ptr(gkey), _ptr(gW1),
          _ptr(gb1), _ptr(gW2), _ptr(gb2), _ptr(ws), ws.numel(), int(parts), int(bool(keep_pad_rows)),
          _ptr(_din_hid(hid, B, L, K, order is not None)), _ptr(_din_order(order, B)), _stream())
    return gq, gkey, gW1, gb1, gW2, gb2


def din_attn_dense_fwd(q, keys, seq_len, W1, b1, W2, b2):
    _req(q, torch.float32, "q", 2)
    _req(keys, torch.float32, "keys", 3)
    _req(seq_len, torch.int32, "seq_len", 1)
    B, L, K = keys.shape
    H = _din_params(W1, b1, W2, b2, K)
    out = torch.empty((B, K), dtype=torch.float32, device=q.device)
    attn = torch.empty((B, L), dtype=torch.float32, device=q.device)
    _call("lr_din_attn_dense_fwd_f32", _ptr(q), _ptr(keys), K, _ptr(seq_len), B, L,
                                                _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), H,
                                                _ptr(out), _ptr(attn), _stream())
    return out, attn


def din_attn_dense_bwd(q, keys, seq_len, W1, b1, W2, b2, attn, gout):
    _req(q, torch.float32, "q", 2)
    _req(keys, torch.float32, "keys", 3)
    B, L, K = keys.shape
    H = _din_params(W1, b1, W2, b2, K)
    dev = q.device
    lib = _lib.load()
    ws = torch.empty(max(lib.lr_din_attn_ws_bytes(B, L, K, H), 8), dtype=torch.uint8, device=dev)
    gq = torch.empty((B, K), dtype=torch.float32, device=dev)
    gkey = torch.empty((B, L, K), dtype=torch.float32, device=dev)
    gW1, gb1 = torch.empty_like(W1), torch.empty_like(b1)
    gW2, gb2 = torch.empty_like(W2), torch.empty_like(b2)
    _call("lr_din_attn_dense_bwd_f32", _ptr(q), _ptr(keys), K, _ptr(seq_len), B, L, _ptr(W1),
                                        _ptr(b1), _ptr(W2), _ptr(b2), H, _ptr(attn), _ptr(gout),
                                        _ptr(gq), _ptr(gkey), _ptr(gW1), _ptr(gb1), _ptr(gW2),
                                        _ptr(gb2), _ptr(ws), ws.numel(), _stream())
    return gq, gkey, gW1, gb1, gW2, gb2


# --------------------------------------------------------------------------------------
# device-side negative sampling
# --------------------------------------------------------------------------------------
def sample_negatives(items_pos: torch.Tensor, num_neg: int, n_items: int, seed: int,
                     users: Optional[torch.Tensor] = None, consumed_ptr: Optional[torch.Tensor] = None,
                     consumed_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 [n * num_neg] negatives, `num_neg` consecutive entries per positive (the layout the
    collators interleave, batch/collators.py:226-232).  With a consumed CSR the "unconsumed" rules
    apply, otherwise the "random" ones (see the header)."""
    _req(items_pos, torch.int32, "items_pos", 1)
    n = items_pos.numel()
    out = torch.empty(n * int(num_neg), dtype=torch.int32, device=items_pos.device)
    if consumed_ptr is not None:
        _req(consumed_ptr, torch.int64, "consumed_ptr", 1)
        _req(consumed_idx, torch.int32, "consumed_idx", 1)
        _req(users, torch.int32, "users", 1)
    _call("lr_sample_negatives_i32", _ptr(users), _ptr(items_pos), n, int(num_neg), int(n_items),
          _ptr(consumed_ptr), _ptr(consumed_idx), int(seed) & ((1 << 64) - 1), _ptr(out), _stream())
    return out


def fm_field_stats(table: torch.Tensor, seg: Segments, field_row_start: torch.Tensor, B: int,
                   chunks: int = 8):
    """(mean, biased var) [F*K] of the gathered block e[B,F,K] from the batch's segments (see the
    header): reads the distinct rows once instead of B*F rows.  fp64 combine of the partials."""
    _req(table, torch.float32, "table", 2)
    _req(field_row_start, torch.int32, "field_row_start", 1)
    K = table.shape[1]
    F = field_row_start.numel() - 1
    partial = torch.empty((F, chunks, 2, K), dtype=torch.float32, device=table.device)
    _call("lr_fm_field_stats_f32", _ptr(table), K, _ptr(seg.rows), _ptr(seg.start), _ptr(seg.n_seg),
          _ptr(field_row_start), F, chunks, _ptr(partial), _stream())
    tot = partial.double().sum(1)                                  # [F, 2, K]
    mean = tot[:, 0] / B
    var = torch.clamp(tot[:, 1] / B - mean * mean, min=0.0)
    return mean.reshape(-1).float(), var.reshape(-1).float()


# --------------------------------------------------------------------------------------
# first Dense layer over a materialised block (general feature nets) — csrc/dense_block.hip
# --------------------------------------------------------------------------------------
def table_colstats(table: torch.Tensor, idx: torch.Tensor, chunks: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """partial [F, chunks, 2, K]: per-chunk column sums / sums of squares of table[idx[b, f], :] (see the header)."""
    _req(table, torch.float32, "table", 2)
    _req(idx, torch.int32, "idx", 2)
    V, K = table.shape
    B, F = idx.shape
    if out is None:
        out = torch.empty((F, chunks, 2, K), dtype=torch.float32, device=table.device)
    elif out.shape != (F, chunks, 2, K):
        raise ValueError("out must be [F, chunks, 2, K]")
    _call("lr_table_colstats_f32", _ptr(table), V, K, _ptr(idx), B, F, int(chunks), _ptr(out), _stream())
    return out


def bn_remainder_(G: torch.Tensor, x: torch.Tensor, a: torch.Tensor, c: torch.Tensor, rows_per_plane: int,
                  period: int = 0) -> torch.Tensor:
    """In place: G[r] -= a[p] + c[p] * x[r]; p = r // rows_per_plane (planar block) or r % period (`period` > 0: row-major
    [B, period * Kp] block); G, x [rows, Kp], a, c [planes * Kp]."""
    for t_, n_ in ((G, "G"), (x, "x"), (a, "a"), (c, "c")):
        _req(t_, torch.float32, n_)
    rows, Kp = G.shape
    planes = period if period > 0 else -(-rows // rows_per_plane)
    if x.shape != G.shape or a.numel() < planes * Kp or c.numel() < planes * Kp:
        raise ValueError("shape mismatch")
    _call("lr_bn_remainder_f32", _ptr(G), _ptr(x), _ptr(a), _ptr(c), rows, int(rows_per_plane), int(period), Kp, _stream())
    return G


# --------------------------------------------------------------------------------------
# LightGCN: normalised bipartite Laplacian built on the device — csrc/laplacian.hip
# --------------------------------------------------------------------------------------
def csr_laplacian(users: torch.Tensor, items: torch.Tensor, n_users: int, n_items: int, want_tperm: bool = True):
    """(rowptr int64 [n+1], col int32 [nnz], val fp32 [nnz], tperm int32 [nnz] | None) of D^-1/2 [[0,R],[R^T,0]] D^-1/2
    from the interaction list (lightgcn_module.py:36-61).  One host read (the number of distinct pairs)."""
    _req(users, torch.int32, "users", 1)
    _req(items, torch.int32, "items", 1)
    E = users.numel()
    if items.numel() != E:
        raise ValueError("users / items must have the same length")
    dev = users.device
    lib = _lib.load()
    n = int(n_users) + int(n_items)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    col = torch.empty(max(2 * E, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(2 * E, 1), dtype=torch.float32, device=dev)
    tperm = torch.empty(max(2 * E, 1), dtype=torch.int32, device=dev) if want_tperm else None
    n_pairs = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(lib.lr_csr_laplacian_ws_bytes(E), 8), dtype=torch.uint8, device=dev)
    _call("lr_csr_laplacian_build", _ptr(users), _ptr(items), E, int(n_users), int(n_items), _ptr(rowptr), _ptr(col),
          _ptr(val), _ptr(tperm), _ptr(n_pairs), _ptr(ws), ws.numel(), _stream())
    nnz = 2 * int(n_pairs.item())
    del ws
    if nnz == 2 * E:
        return rowptr, col, val, tperm
    return rowptr, col[:nnz].clone(), val[:nnz].clone(), (tperm[:nnz].clone() if want_tperm else None)
