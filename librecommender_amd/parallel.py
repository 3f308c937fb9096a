"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

* **Row-sharded embedding tables** — global row ``r`` lives on rank ``r % W`` as local row
  ``r // W`` (round-robin spreads every field's Zipf head over all ranks).  A step
    1. de-duplicates the batch's row ids (the same device radix sort that the backward needs),
    2. all-to-all: ids to owners, rows (+ linear weight as an extra column) back,
    3. runs the fused FM kernels on the compact per-step *row cache*,
    4. all-to-all: per-row gradients to owners, which sum duplicates across peers and apply
       row-wise Adam (`lr_embed_scatter_adam_f32`).
  Per step and rank the exchange moves ``U·(K+1)·4`` bytes each way (U = distinct rows of the
  local batch) instead of ``B·F·K·4`` — on Zipf ids that is the difference that matters on
  per-link-bound xGMI.
* **Replicated dense parameters** — one flat gradient buffer, one all-reduce.
* **Item-sharded scoring** — every rank scores its slice with ``item_base``; candidates are
  all-gathered and merged (``lr_topk_merge_f32``).

Compute kernels are reached through a *kernel provider* so that the exchange logic can be
exercised on CPU (gloo, world_size 2) with the oracle standing in for the HIP ops in tests.
The product provider is :class:`HipKernels`; there is no CPU provider in the package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import os

import torch
import torch.distributed as dist


class HipKernels:
    """Product kernel provider: the C-ABI ops."""

    def __init__(self):
        from . import ops

        self.ops = ops
        self._builders = {}

    MAX_BUILDERS = 16

    def segments(self, idx: torch.Tensor, V: int, want_slots: bool = False, tag: str = ""):
        key = (V, idx.device, tag)
        b = self._builders.pop(key, None)
        if b is None or b.n_max < idx.numel():
            b = self.ops.SegmentBuilder(max(idx.numel(), 1), V, idx.device)
        self._builders[key] = b                      # (re-inserted last: the dict is the LRU order)
        while len(self._builders) > self.MAX_BUILDERS:      # a caller that keys by a per-batch value must not grow the cache forever
            self._builders.pop(next(iter(self._builders)))
        return b.build(idx.reshape(-1), want_slots=want_slots)

    def gather(self, table, ids):
        return self.ops.embed_gather(table, ids)

    def bag_pool(self, table, idx, combiner, oov):
        return self.ops.embed_bag_pool(table, idx, combiner, oov)

    def bag_pool_bwd(self, gout, idx, V, combiner, oov):
        return self.ops.embed_bag_pool_bwd(gout, idx, V, combiner, oov)

    def fm_pairwise(self, e):
        return self.ops.fm_pairwise_fwd(e)

    def fm_pairwise_bwd(self, e, fsum, gpair):
        return self.ops.fm_pairwise_bwd(e, fsum, gpair)

    def fm_fwd(self, cache, lin_cache, slots, want_e=True):
        return self.ops.fm_embed_fwd(cache, slots, want_e=want_e, lin=lin_cache)

    def fm_bwd_rows(self, cache, gdeep, gpair, fsum, B, F, seg, glin, bn_a, bn_c):
        return self.ops.fm_embed_bwd_rows(cache, gdeep, gpair, fsum, B, F, seg, glin=glin, bn_a=bn_a, bn_c=bn_c)

    def scatter_adam(self, table, m, v, grads, seg, hp):
        self.ops.embed_scatter_adam(table, m, v, grads, seg, hp)

    def peer_adam(self, table, m, v, grads, ids, peer_counts, hp, lin, lin_m, lin_v, glin, peer_tab):
        self.ops.embed_peer_adam(table, m, v, grads.contiguous(), ids, peer_counts, hp, lin, lin_m, lin_v,
                                 None if glin is None else glin.contiguous().view(-1), peer_tab)

    def adam_dense_rows(self, table, m, v, grads, seg, hp, l2=0.0):
        """Dense (TF1) Adam over every row of `table`, the gradient rows `grads` [n, K] summed per row through `seg` first."""
        key = (table.data_ptr(), table.shape[0])
        slot = self._row_slots.get(key) if hasattr(self, "_row_slots") else None
        if slot is None:
            if not hasattr(self, "_row_slots"):
                self._row_slots = {}
            slot = self._row_slots[key] = torch.full((table.shape[0],), -1, dtype=torch.int32, device=table.device)
        grows = self.ops.embed_segment_sum(grads.contiguous(), seg) if seg.n > 0 else None
        self.ops.adam_dense(table, m, v, hp, grows=grows, seg=seg if seg.n > 0 else None, row_slot=slot, l2=l2)

    def scatter_adam_lin(self, table, m, v, grads, lin, lin_m, lin_v, glin, seg, hp):
        """Owner-side update of a table and its linear weights from one pass over the received rows."""
        self.ops.embed_scatter_adam_lin(table, m, v, grads.contiguous(), lin, lin_m, lin_v, glin.contiguous(), seg, hp)

    def segment_sum(self, grads, seg):
        """[n_pos, K] per-position gradients -> [n_runs(+), K] per distinct row, run order."""
        return self.ops.embed_segment_sum(grads, seg)

    def score_topk(self, users, items, k, ptr, cidx, flag, item_base):
        return self.ops.score_topk(users, items, k, ptr, cidx, flag, item_base=item_base)

    def topk_merge(self, scores, ids):
        return self.ops.topk_merge(scores, ids)

    def softmax_ce(self, X, Y, col_bias, row_ids, col_ids, pos0):
        """Per-row in-batch softmax cross-entropy of X @ Y.T (+ col_bias, accidental-hit mask), streaming."""
        return self.ops.softmax_ce(X, Y, col_bias, row_ids, col_ids, pos0)

    def din_attention(self, q, keys, lens, W1, b1, W2, b2):
        """`din_attention` (layers/attention.py:28-64) on materialised (query, keys) rows, differentiable: the MFMA
        attention kernels in their dense form (key widths up to 128, zero-padded to the next compiled width)."""
        from .nets.feat_nets import din_attention_dense, din_attention_torch
        att = din_attention_dense(q, keys, lens, W1, b1, W2, b2)
        return att if att is not None else din_attention_torch(q, keys, lens, W1, b1, W2, b2)

    def adam_hp(self, lr, step, eps):
        return self.ops.adam_hp(lr, step, eps=eps, tf_style=True)

    def adam_hp_torch(self, lr, step, eps, weight_decay=0.0):
        return self.ops.adam_hp(lr, step, eps=eps, weight_decay=weight_decay, tf_style=False)

    def _spmm_plan(self, rowptr, nnz, K):
        """The chunk lists of a graph's long rows are built by its first product and reused by every later one."""
        plans = getattr(self, "_plans", None)
        if plans is None:
            plans = self._plans = {}
        import weakref

        key = (rowptr.data_ptr(), int(nnz), int(K))
        ent = plans.get(key)
        # the lists belong to ONE rowptr tensor: a later graph of the same shape that the allocator places at the freed address
        # must not inherit them (weak reference to the tensor object, like ops' CDF cache)
        if ent is None or ent[0]() is not rowptr or not ent[1].matches(rowptr, nnz, K):
            if len(plans) >= 32:            # (a net holds one graph, or its handful of column blocks)
                plans.clear()
            ent = plans[key] = (weakref.ref(rowptr), self.ops.SpmmPlan(rowptr, nnz, K))
        return ent[1]

    def spmm(self, rowptr, col, val, X, out, acc, x_rows=None, y_rows=None):
        """`x_rows` / `y_rows` (`row_bitmap` objects): rows of X outside are zero / only these rows of `out` are wanted.
        `out` None with `acc`: `acc += A X` and nothing else is stored."""
        acc_only = out is None and acc is not None
        if X.shape[1] in (16, 32, 64, 128):
            return self.ops.spmm_csr(rowptr, col, val, X, out=out, acc=acc, plan=self._spmm_plan(rowptr, col.numel(), X.shape[1]),
                                     x_rows=x_rows, y_rows=y_rows, acc_only=acc_only)
        return self.ops.spmm_csr(rowptr, col, val, X, out=out, acc=acc, acc_only=acc_only)

    def row_bitmap(self, n_rows, device):
        return self.ops.RowBitmap(n_rows, device)

    def spmm_adam(self, rowptr, col, val, X, w, m, v, hp, vmax, seg, g, alpha, row_slot):
        """One Adam step of (w, m, v[, vmax]) with the gradient A X + alpha * (the routed gradient rows `g`, summed per row of
        `seg`): the epilogue of the product (no gradient table).  False when the width is not compiled (caller falls back)."""
        if X.shape[1] not in (16, 32, 64, 128):
            return False
        gsum = None
        if seg is not None:
            gsum = self.ops.embed_segment_sum(g, seg)
            self.ops.row_slots(seg, row_slot, True)
        self.ops.spmm_csr_adam(rowptr, col, val, X, w, m, v, hp, self._spmm_plan(rowptr, col.numel(), X.shape[1]), vmax=vmax,
                               row_slot=row_slot if seg is not None else None, gsum=gsum, alpha=alpha)
        if seg is not None:
            self.ops.row_slots(seg, row_slot, False)
        return True

    def scatter_add(self, table, grads, seg, alpha=1.0):
        self.ops.embed_scatter_add(table, grads, seg, alpha=alpha)

    def dense_adam(self, flat, m, v, grad, hp):
        self.ops.adam_dense(flat.view(-1, 1), m.view(-1, 1), v.view(-1, 1), hp, grows=grad)

    def fm_bwd_adam(self, t, gdeep, gpair, fsum, B, F, seg, glin, bn_a, bn_c, hp):
        """Fused FM backward + row-wise Adam on the tables of `t` (embed/m/v, lin/lin_m/lin_v)."""
        need = self.ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F)
        ws = getattr(t, "_bwd_ws", None)
        if ws is None or ws.numel() < need:
            ws = t._bwd_ws = torch.empty(need, dtype=torch.uint8, device=t.embed.device)
        self.ops.fm_embed_bwd_adam(t.embed, t.m, t.v, gdeep, gpair, fsum, B, F, seg, hp, lin=t.lin,
                                   lin_m=t.lin_m, lin_v=t.lin_v, glin=glin, bn_a=bn_a, bn_c=bn_c, ws=ws)

    def adam_table(self, table, m, v, grad, hp, vmax=None):
        """Adam over every row of a [V, K] parameter with a dense gradient (optional AMSGrad)."""
        self.ops.adam_dense(table, m, v, hp, grows=grad, vmax=vmax)


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo has no device all-to-all / all-gather: with that backend (functional checks of the
    HIP + exchange path with several ranks sharing ONE GPU) collectives are staged through host
    memory.  Under RCCL ("nccl") buffers stay on the device."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _a2a_single(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None) -> None:
    if dist.get_world_size(group) == 1:      # no peer: RCCL would still launch a copy kernel (~0.16 ms per call at 400 MB)
        out.copy_(inp)
    elif _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    if _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _all_gather_into_async(out: torch.Tensor, inp: torch.Tensor, group=None):
    """`_all_gather_into` enqueued beside the current stream where the backend runs collectives on a stream of its own (RCCL):
    returns the work handle to `wait()` on before `out` is read — kernels launched in between overlap the transfer.  None: the
    gather is already complete (gloo: functional runs, staged through host memory)."""
    if _host_staged(inp, group) or dist.get_backend(group) == "gloo":
        _all_gather_into(out, inp, group=group)
        return None
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)


def _all_to_all_rows(send: torch.Tensor, send_counts: List[int], recv_counts: List[int], group=None) -> torch.Tensor:
    if dist.get_world_size(group) == 1:      # a group of one exchanges nothing: the rows to "receive" are the rows to send
        return send.contiguous()
    out = torch.empty((sum(recv_counts), *send.shape[1:]), dtype=send.dtype, device=send.device)
    _a2a_single(out, send.contiguous(), recv_counts, send_counts, group=group)
    return out


@dataclass
class LookupPlan:
    idx: torch.Tensor           # the batch's global row ids the plan was built for
    seg: object
    n_rows: int
    send_counts: List[int]
    recv_counts: List[int]
    slots: torch.Tensor
    parity: int = 0             # which of the two segment workspaces holds `seg` / `slots`
    send_ids: Optional[torch.Tensor] = None   # field plans: the owners' local rows in owner-major order (capacity-sized)
    slotsT: Optional[torch.Tensor] = None     # field plans: int32 [F, B] position -> cache row, field-major
    fseg: Optional[object] = None             # field plans: the field-wise runs of the global ids (`FieldSegmentBuilder`)
    pending: Optional[tuple] = None           # (pinned [2, W (+1)] int64 counts, event): the host read has not happened yet
    n_pos: int = -1                           # field plans: positions of the batch (all of them must have been kept)

    def resolve(self) -> "LookupPlan":
        """The plan's one host read (rows per peer), taken when the exchange needs it: the counts were copied to pinned
        memory behind the plan's kernels, so building a plan never blocks the host."""
        if self.pending is not None:
            host, ev = self.pending
            if ev is not None:
                ev.synchronize()
            send, recv = host.tolist()
            self.pending = None
            if self.n_pos >= 0:
                W = len(send) - 1
                if send[W] != self.n_pos:
                    raise ValueError(f"{self.n_pos - send[W]} id(s) of the batch lie outside the row range of their column's "
                                     f"field (`field_row_start`): column f of `idx` may only hold rows of field f")
                send, recv = send[:W], recv[:W]
            self.send_counts, self.recv_counts = send, recv
            self.n_rows = sum(self.send_counts)
        return self


@dataclass
class LookupCtx:
    seg: object                 # segments of the local batch's (owner-major) row keys
    n_rows: int                 # distinct rows U
    send_counts: List[int]      # rows requested from each owner
    recv_counts: List[int]      # rows each peer requests from this rank
    recv_ids: torch.Tensor      # local row ids requested from this rank (int32)
    cache: torch.Tensor         # [U, K] rows in run order
    lin_cache: Optional[torch.Tensor]  # [U, 1]
    slots: torch.Tensor         # int32 [B, F] position -> run number
    parity: int = 0
    slotsT: Optional[torch.Tensor] = None     # field plans (see LookupPlan)
    fseg: Optional[object] = None


class ShardedFieldTables:
    """Row-sharded (round-robin) embedding + linear tables with Adam state.

    The batch's row ids are re-keyed ``key = (row % W) * ceil(V/W) + row // W`` before the
    device sort, so the runs ("segments") come out owner-major: the de-duplicated request list,
    the returned row cache and the per-row gradients are all contiguous per peer and in run
    order — no permutation passes, and the owner/local-row pair is read back off the key."""

    def __init__(self, V: int, K: int, device, kern, rank: Optional[int] = None,
                 world: Optional[int] = None, with_linear: bool = True, group=None, seed: int = 42):
        self.V, self.K, self.device, self.kern, self.group = int(V), int(K), device, kern, group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.V_local = (self.V - self.rank + self.world - 1) // self.world
        self.V_stride = (self.V + self.world - 1) // self.world      # key stride per owner
        gen = torch.Generator(device=device)
        gen.manual_seed(seed + 1000 * self.rank)
        self.embed = (torch.rand((self.V_local, K), generator=gen, device=device) - 0.5) * 0.02
        self.lin = (torch.rand((self.V_local, 1), generator=gen, device=device) - 0.5) * 0.02 if with_linear else None
        self._init_state()

    def _init_state(self):
        self.m = torch.zeros_like(self.embed)
        self.v = torch.zeros_like(self.embed)
        if self.lin is not None:
            self.lin_m = torch.zeros_like(self.lin)
            self.lin_v = torch.zeros_like(self.lin)

    # ---- id-space layout of the feature models ([user | item | sparse] global rows, layers/embedding.py:FieldTables) ----
    def set_layout(self, n_users: int, n_items: int) -> None:
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.user_off, self.item_off, self.sparse_off = 0, self.n_users + 1, self.n_users + 1 + self.n_items + 1

    def global_idx(self, users, items, sparse_indices=None) -> torch.Tensor:
        """[B, 2 + Fs] int32 global rows in the field order of deepfm.py:210-214 (user, item, sparse columns)."""
        cols = [users.to(torch.int32).view(-1, 1) + self.user_off, items.to(torch.int32).view(-1, 1) + self.item_off]
        if sparse_indices is not None:
            cols.append(sparse_indices.to(torch.int32) + self.sparse_off)
        return torch.cat(cols, dim=1).contiguous()

    @torch.no_grad()
    def assign_oov(self, sparse_oov_rows) -> None:
        """OOV rows := mean of the real rows of their id space (`bases/tf_base.py:310-353`), on row-sharded tables: every
        rank sums its rows of each range, one all-reduce of [ranges, K + 1] sums, the owner of a target row writes it."""
        ranges = [(self.user_off, self.user_off + self.n_users, self.user_off + self.n_users),
                  (self.item_off, self.item_off + self.n_items, self.item_off + self.n_items)]
        start = 0
        for oov in (sparse_oov_rows if sparse_oov_rows is not None else []):
            oov = int(oov)
            if start < oov:
                ranges.append((self.sparse_off + start, self.sparse_off + oov, self.sparse_off + oov))
                start = oov + 1
        dev = self.embed.device
        g = torch.arange(self.rank, self.V, self.world, device=dev)[: self.embed.shape[0]]      # global ids of the local rows
        lo = torch.tensor([r[0] for r in ranges], device=dev)
        hi = torch.tensor([r[1] for r in ranges], device=dev)
        rid = torch.searchsorted(lo, g, right=True) - 1                       # range starts ascend; -1: before the first
        ok = (rid >= 0) & (g < hi[rid.clamp(min=0)])
        rows = torch.cat([self.embed, self.lin if self.lin is not None else self.embed[:, :0]], dim=1).double()
        sums = torch.zeros((len(ranges), rows.shape[1]), dtype=torch.float64, device=dev)
        sums.index_add_(0, rid[ok], rows[ok])
        allreduce_sum_(sums, self.group)
        for k, (a, b, tgt) in enumerate(ranges):
            if tgt % self.world == self.rank and b > a:
                mean = (sums[k] / (b - a)).float()
                self.embed[tgt // self.world] = mean[: self.K]
                if self.lin is not None:
                    self.lin[tgt // self.world] = mean[self.K:]

    def load_full(self, full_embed: torch.Tensor, full_lin: Optional[torch.Tensor] = None):
        """Take this rank's rows out of an unsharded table (tests / checkpoint load)."""
        self.embed = full_embed[self.rank::self.world].to(self.device).contiguous().clone()
        if full_lin is not None:
            self.lin = full_lin.reshape(-1, 1)[self.rank::self.world].to(self.device).contiguous().clone()
        self._init_state()

    def gather_full(self) -> torch.Tensor:
        """All-gather the shards back into the unsharded layout (tests / export)."""
        parts = [None] * self.world
        dist.all_gather_object(parts, (self.embed.cpu(), None if self.lin is None else self.lin.cpu()), group=self.group)
        full = torch.empty((self.V, self.K))
        full_lin = torch.empty((self.V, 1)) if self.lin is not None else None
        for r, (e, l) in enumerate(parts):
            full[r::self.world] = e
            if l is not None:
                full_lin[r::self.world] = l
        return full, full_lin

    # ---- per-shard checkpoint (utils/save_load.py:70-115 semantics, one file set per rank) -------------
    # `<name>_shard<r>of<W>.npz` holds the shard's META data (V, K, rank, world); every array is its own
    # `<name>_shard<r>of<W>.<key>.npy`, so a loader on another world size can MEMORY-MAP an old shard and copy only the
    # rows it owns (round-2/3 advisor item: `.npz` members cannot be mapped, every rank used to read every old shard whole —
    # tens of GB of host RAM per rank at 100 M x 128).  Checkpoints of the earlier layout (arrays inside the .npz) still load.
    _CKPT_KEYS = ("embed", "m", "v", "lin", "lin_m", "lin_v")

    def save_shard(self, path: str, name: str = "tables") -> str:
        """Write THIS rank's rows, linear weights and Adam moments (no gather: a 100 M x 128 table never exists in one
        place).  `load_shard` restores them on the same world size; `load_shards_resharded` re-distributes onto a
        different one."""
        import os

        import numpy as np

        os.makedirs(path, exist_ok=True)
        stem = os.path.join(path, f"{name}_shard{self.rank}of{self.world}")
        keys = [k for k in self._CKPT_KEYS if getattr(self, k, None) is not None]
        # Every save writes its arrays under its OWN tag (`<stem>.s<tag>.<key>.npy`), then the meta file — which names the tag
        # and the keys — through a temporary + rename, then removes the side files of other tags.  A save interrupted at any
        # point leaves the previous meta AND the arrays it names untouched (round-5 advisor finding: replacing untagged side
        # files in place could leave a new `embed` beside old moments under the old meta); its orphans go with the next save.
        import glob

        tag = 1
        if os.path.exists(stem + ".npz"):
            try:
                with np.load(stem + ".npz") as z:
                    tag = (int(z["tag"]) if "tag" in z.files else 0) + 1
            except Exception:  # noqa: BLE001  (an unreadable old meta: start over)
                tag = 1
        for k in keys:
            tmp = f"{stem}.s{tag}.{k}.tmp.npy"
            np.save(tmp, getattr(self, k).cpu().numpy())
            os.replace(tmp, f"{stem}.s{tag}.{k}.npy")
        tmp = stem + ".tmp.npz"
        np.savez(tmp, V=np.int64(self.V), K=np.int64(self.K), rank=np.int64(self.rank), world=np.int64(self.world),
                 keys=np.asarray(keys), tag=np.int64(tag))
        os.replace(tmp, stem + ".npz")
        for f in glob.glob(glob.escape(stem) + ".*.npy"):           # other tags, the untagged files of earlier layouts
            if not os.path.basename(f).startswith(os.path.basename(stem) + f".s{tag}."):
                os.remove(f)
        return stem + ".npz"

    @staticmethod
    def shard_array(path: str, name: str, r: int, w: int, key: str, meta, mmap: bool = False):
        """Array `key` of shard r-of-w: the side file (memory-mapped on request) or, for checkpoints of the earlier layout,
        the member of the .npz."""
        import os

        import numpy as np

        tagged = "tag" in meta                     # (round 6: arrays carry the tag of the save that wrote `meta`)
        f = os.path.join(path, f"{name}_shard{r}of{w}.s{int(meta['tag'])}.{key}.npy" if tagged else f"{name}_shard{r}of{w}.{key}.npy")
        listed = [str(k) for k in meta["keys"]] if "keys" in meta else None     # the arrays the save that wrote `meta` holds
        if listed is not None and key not in listed:
            raise FileNotFoundError(f"{name}_shard{r}of{w}: no array `{key}` (the checkpoint lists {listed})")
        if os.path.exists(f):
            return np.load(f, mmap_mode="r" if mmap else None)
        if tagged:
            raise FileNotFoundError(f"{f}: the checkpoint's meta file names save {int(meta['tag'])} but its array `{key}` is missing")
        if key in meta:
            return meta[key]
        raise FileNotFoundError(f"{name}_shard{r}of{w}: no array `{key}`")

    def load_shard(self, path: str, name: str = "tables") -> None:
        import os

        import numpy as np

        f = os.path.join(path, f"{name}_shard{self.rank}of{self.world}.npz")
        with np.load(f) as z:
            if int(z["V"]) != self.V or int(z["K"]) != self.K or int(z["world"]) != self.world:
                raise ValueError(f"{f} holds a [{int(z['V'])}, {int(z['K'])}] table sharded {int(z['world'])}-way, "
                                 f"this is [{self.V}, {self.K}] sharded {self.world}-way (use load_shards_resharded)")
            t = lambda k: torch.from_numpy(np.ascontiguousarray(self.shard_array(path, name, self.rank, self.world, k, z))).to(self.device).contiguous()  # noqa: E731
            self.embed, self.m, self.v = t("embed"), t("m"), t("v")
            if self.lin is not None:
                self.lin, self.lin_m, self.lin_v = t("lin"), t("lin_m"), t("lin_v")

    def load_shards_resharded(self, path: str, name: str = "tables") -> None:
        """Load a checkpoint written on a DIFFERENT world size: global row r lives in old shard r % W_old at local
        row r // W_old and goes to local row r // W of rank r % W here.  Every rank memory-maps every old shard and reads
        only the rows it owns.  All `W_old` files of ONE world size must be present; the `rank` / `world` stored in each
        file are checked against its name."""
        import glob
        import os

        import numpy as np

        files = sorted(glob.glob(os.path.join(path, f"{name}_shard*of*.npz")))
        if not files:
            raise FileNotFoundError(f"no {name}_shard*of*.npz under {path}")
        worlds = {int(f.rsplit("of", 1)[1].split(".")[0]) for f in files}
        if len(worlds) != 1:
            raise ValueError(f"{path} holds {name} shards of several world sizes {sorted(worlds)}: remove the stale set")
        w_old = worlds.pop()
        if len(files) != w_old:
            raise FileNotFoundError(f"{len(files)} of {w_old} {name} shards found under {path}")
        mine = torch.arange(self.rank, self.V, self.world)                   # my global rows, local order
        keys = ["embed", "m", "v"] + (["lin", "lin_m", "lin_v"] if self.lin is not None else [])
        out = {k: None for k in keys}
        for r_old in range(w_old):
            f = os.path.join(path, f"{name}_shard{r_old}of{w_old}.npz")
            sel = (mine % w_old) == r_old
            with np.load(f) as z:
                if int(z["V"]) != self.V or int(z["K"]) != self.K:
                    raise ValueError(f"{f}: table shape differs")
                if int(z["world"]) != w_old or int(z["rank"]) != r_old:
                    raise ValueError(f"{f}: stored rank / world ({int(z['rank'])} of {int(z['world'])}) do not match the file name")
                src_rows = (mine[sel] // w_old).numpy()
                for k in keys:
                    a = self.shard_array(path, name, r_old, w_old, k, z, mmap=True)      # mapped: only the rows below are read
                    if out[k] is None:
                        out[k] = np.empty((len(mine),) + a.shape[1:], dtype=a.dtype)
                    out[k][sel.numpy()] = a[src_rows]
        for k in keys:
            setattr(self, k, torch.from_numpy(out[k]).to(self.device).contiguous())

    def load_shards_mapped(self, path: str, name: str, src, dst, V_old: int, keys=None) -> None:
        """Retraining under a process group (`tfops/rebuild.py:12-139`): global row src[i] of a sharded checkpoint of the
        OLD (smaller) table becomes global row dst[i] of this table; rows not named keep their fresh initialisation, moments
        of new rows stay zero.  The old shards are memory-mapped; every rank reads only the rows it will own."""
        import glob
        import os

        import numpy as np

        files = sorted(glob.glob(os.path.join(path, f"{name}_shard*of*.npz")))
        worlds = {int(f.rsplit("of", 1)[1].split(".")[0]) for f in files}
        if len(worlds) != 1 or len(files) != next(iter(worlds), -1):
            raise ValueError(f"{path}: expected the complete {name} shard set of ONE world size, found {len(files)} files of "
                             f"world sizes {sorted(worlds)}")
        w_old = worlds.pop()
        src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
        own = (dst % self.world) == self.rank
        src, dst_local = src[own], dst[own] // self.world
        keys = list(keys) if keys is not None else ["embed", "m", "v"] + (["lin", "lin_m", "lin_v"] if self.lin is not None else [])
        for r_old in range(w_old):
            sel = (src % w_old) == r_old
            if not sel.any():
                continue
            with np.load(os.path.join(path, f"{name}_shard{r_old}of{w_old}.npz")) as z:
                if int(z["V"]) != int(V_old) or int(z["K"]) != self.K:
                    raise ValueError(f"{name}_shard{r_old}of{w_old}: a [{int(z['V'])}, {int(z['K'])}] table, expected [{V_old}, {self.K}]")
                rows_old = src[sel] // w_old
                order = np.argsort(rows_old, kind="stable")                  # ascending reads of the mapped file
                to = torch.from_numpy(dst_local[sel][order]).to(self.device)
                for k in keys:
                    a = self.shard_array(path, name, r_old, w_old, k, z, mmap=True)
                    getattr(self, k)[to] = torch.from_numpy(np.ascontiguousarray(a[rows_old[order]])).to(self.device)

    # ---- forward exchange ------------------------------------------------------------------
    def plan(self, idx: torch.Tensor) -> "LookupPlan":
        """The part of a lookup that depends on the ids only: owner-major de-duplication and the
        per-peer row counts (one exchange of W integers + ONE host read of both count lists).  Plans
        alternate between two segment workspaces, so the plan of step t+1 can be built (`prefetch`)
        while step t still uses its own."""
        W, Vs = self.world, self.V_stride
        B, F = idx.shape
        n_pos = B * F
        self._plan_no = getattr(self, "_plan_no", 0) + 1
        parity = self._plan_no & 1
        if self._field_plans(idx):
            return self._plan_fields(idx, parity)
        key = idx if W == 1 else (idx % W) * Vs + torch.div(idx, W, rounding_mode="floor")
        seg = self.kern.segments(key.to(torch.int32), W * Vs, want_slots=True, tag=f"lookup{parity}")
        valid = torch.arange(n_pos, device=idx.device, dtype=torch.int32) < seg.n_seg
        owner = torch.where(valid, torch.div(seg.rows[:n_pos], Vs, rounding_mode="floor"), W).long()
        send_counts_t = torch.bincount(owner, minlength=W + 1)[:W]
        recv_counts_t = torch.empty_like(send_counts_t)
        _a2a_single(recv_counts_t, send_counts_t, group=self.group)
        send_counts, recv_counts = torch.stack([send_counts_t, recv_counts_t]).tolist()     # host read
        return LookupPlan(idx, seg, sum(send_counts), send_counts, recv_counts, seg.slots.view(B, F), parity)

    # ---- field plans: the exchange plan off the field-wise sort --------------------------------------------------
    def set_fields(self, field_row_start: torch.Tensor) -> None:
        """`field_row_start` int32 [F + 1] on the device (first global row of every field, ascending): column f of every
        batch then only holds rows of field f, and plans are built from the field-wise LDS sort (`lr_segments_build_fields`)
        + the owner partition (`lr_owner_partition_i32`) instead of a device-wide radix sort of owner-major keys.  The
        same runs drive the fused step (first layer, statistics, row gradients), which then builds nothing itself."""
        self._frs = field_row_start

    def _field_plans(self, idx: torch.Tensor) -> bool:
        from . import ops

        return (getattr(self, "_frs", None) is not None and idx.is_cuda and isinstance(self.kern, HipKernels)
                and idx.shape[0] <= ops.FieldSegmentBuilder.MAX_B and idx.shape[1] + 1 == self._frs.numel()
                and self.world <= 64)

    def _plan_fields(self, idx: torch.Tensor, parity: int) -> "LookupPlan":
        from . import ops

        W = self.world
        B, F = idx.shape
        dev = idx.device
        if not hasattr(self, "_fbufs"):
            self._fbufs = {}
        bufs = self._fbufs.get((parity, B))
        if bufs is None:
            bufs = self._fbufs[(parity, B)] = dict(
                idxT=torch.empty((F, B), dtype=torch.int32, device=dev),
                fseg=ops.FieldSegmentBuilder(B, F, self.V, dev, want_runs=True),
                slots=torch.empty((B, F), dtype=torch.int32, device=dev),
                slotsT=torch.empty((F, B), dtype=torch.int32, device=dev) if W > 1 else None,
                part=ops.OwnerPartition(B * F, W, dev) if W > 1 else None,
                counts=torch.zeros((2, W + 1), dtype=torch.int64, device=dev),      # [send | recv] x W owners, + positions kept
                pin=torch.zeros((2, W + 1), dtype=torch.int64).pin_memory())
        idxT = ops.idx_transpose(idx, out=bufs["idxT"])
        seg = bufs["fseg"].build(idxT, self._frs)
        counts = bufs["counts"]
        if W == 1:                                # one owner: run order IS the exchange order, global row == local row
            slotsT, send_ids = seg.runT, seg.rows
            counts[0, :1].copy_(seg.n_seg)
            counts[1, :1].copy_(seg.n_seg)
        else:
            perm, send_ids, c = bufs["part"].run(seg.rows, seg.n_seg)
            slotsT = bufs["slotsT"]
            torch.index_select(perm, 0, seg.runT.reshape(-1).clamp_min(0), out=slotsT.view(-1))
            counts[0, :W].copy_(c[:W])
            _a2a_single(counts[1, :W], counts[0, :W], group=self.group)
        # positions the field-wise build kept (= seg_start[n_seg]): an id outside its column's row range is dropped there
        # and would have no cache row — checked on the host when the plan is resolved, at no extra synchronisation
        counts[0, W:].copy_(torch.index_select(seg.start, 0, seg.n_seg))
        slots = ops.idx_transpose(slotsT, out=bufs["slots"])          # [F, B] -> [B, F]
        bufs["pin"].copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return LookupPlan(idx, seg, -1, [], [], slots, parity, send_ids=send_ids, slotsT=slotsT, fseg=seg,
                          pending=(bufs["pin"], ev), n_pos=B * F)

    def prefetch(self, idx: torch.Tensor, ready: Optional["torch.cuda.Event"] = None) -> None:
        """Build the plan of a FUTURE batch on a side stream.  Call it after the current step has been
        enqueued: the host read at the end of `plan` then waits for a few small kernels that run
        beside the current step instead of stalling the launch pipeline in front of the next one (the
        per-step host sync of the first row-sharded path cost ~0.9 ms per step at world size 1).
        `idx` must already be resident (pass the event that marks it ready otherwise)."""
        if idx.is_cuda:
            if getattr(self, "_plan_stream", None) is None:
                # a HIGH-PRIORITY stream: streams of equal priority were mapped onto the same hardware queue as the step's stream
                # (rocprofv3 trace of round 6: plan kernels queued in line with the step, +0.2 ms) — another priority is
                # another queue, so the plan's small kernels really run beside the step's large ones
                prio = int(os.environ.get("LIBRECO_PLAN_STREAM_PRIORITY", "-1"))
                self._plan_stream = torch.cuda.Stream(device=idx.device, priority=prio)
            if ready is not None:
                self._plan_stream.wait_event(ready)
            # The plan about to be built re-uses the segment workspace of parity (plan_no + 1) & 1, last read by the
            # step before the one just enqueued: wait for THAT step's end-of-step event (recorded in
            # `apply_gradients`) — not for the current step, beside which the plan kernels are meant to run.  A
            # lookup of that parity that never reached `apply_gradients` (inference) leaves no event: wait for
            # everything enqueued so far instead.  Without this the host, which only ever synchronises with the
            # side stream, can run a step ahead and rebuild `seg.rows / slots` under kernels that still read them.
            p = (getattr(self, "_plan_no", 0) + 1) & 1
            ev = getattr(self, "_ws_free", {}).get(p)
            if getattr(self, "_ws_dirty", {}).get(p, False) or (ev is None and getattr(self, "_plan_no", 0) >= 2):
                self._plan_stream.wait_stream(torch.cuda.current_stream(idx.device))
            elif ev is not None:
                self._plan_stream.wait_event(ev)
            with torch.cuda.stream(self._plan_stream):
                self._next_plan = self.plan(idx)
        else:
            self._next_plan = self.plan(idx)

    def lookup(self, idx: torch.Tensor) -> LookupCtx:
        nxt = getattr(self, "_next_plan", None)
        self._next_plan = None
        if nxt is not None and nxt.idx is idx:
            plan = nxt
            self.plans_prefetched = getattr(self, "plans_prefetched", 0) + 1     # (diagnostics: built a step ahead / in line)
            if idx.is_cuda:   # the plan's arrays were produced on the side stream
                torch.cuda.current_stream(idx.device).wait_stream(self._plan_stream)
        else:
            plan = self.plan(idx)
            self.plans_inline = getattr(self, "plans_inline", 0) + 1
        plan.resolve()
        seg, n, send_counts, recv_counts = plan.seg, plan.n_rows, plan.send_counts, plan.recv_counts
        Vs = self.V_stride
        send_ids = plan.send_ids[:n] if plan.send_ids is not None else (seg.rows[:n] % Vs).to(torch.int32)
        recv_ids = _all_to_all_rows(send_ids, send_counts, recv_counts, self.group)
        cache = _all_to_all_rows(self.kern.gather(self.embed, recv_ids), recv_counts, send_counts, self.group)
        lin_cache = None
        if self.lin is not None:
            lin_cache = _all_to_all_rows(self.kern.gather(self.lin, recv_ids), recv_counts, send_counts, self.group)
        if not hasattr(self, "_ws_dirty"):
            self._ws_dirty, self._ws_free = {}, {}
        self._ws_dirty[plan.parity] = True          # in use until the step's `apply_gradients` records its event
        return LookupCtx(seg, n, send_counts, recv_counts, recv_ids, cache, lin_cache, plan.slots, plan.parity,
                         slotsT=plan.slotsT, fseg=plan.fseg)

    # ---- backward exchange -----------------------------------------------------------------
    def _release_ws(self, ctx: LookupCtx) -> None:
        """End of the step that used `ctx`: its segment workspace may be rebuilt once everything enqueued so far ran."""
        if ctx.cache.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(ctx.cache.device))
            self._ws_free[ctx.parity] = ev
        self._ws_dirty[ctx.parity] = False

    def apply_gradients(self, ctx: LookupCtx, grows: torch.Tensor, glin_rows: Optional[torch.Tensor], hp):
        try:
            self._apply_gradients(ctx, grows, glin_rows, hp)
        finally:
            self._release_ws(ctx)

    def _apply_gradients(self, ctx: LookupCtx, grows: torch.Tensor, glin_rows: Optional[torch.Tensor], hp):
        recv = _all_to_all_rows(grows[: ctx.n_rows], ctx.send_counts, ctx.recv_counts, self.group)
        recv_lin = None
        if self.lin is not None:
            recv_lin = _all_to_all_rows(glin_rows[: ctx.n_rows].reshape(-1, 1), ctx.send_counts,
                                        ctx.recv_counts, self.group)
        if getattr(self, "dense_adam", False):
            # TF1 semantics (training/tf_trainer.py:120: every row decays m, v and moves every step; `reg` adds 2 * l2 * w to
            # every row's gradient): each owner runs the dense update over ALL its local rows, the received row gradients
            # summed per row first — N ranks move exactly the rows one rank would
            seg = self.kern.segments(ctx.recv_ids, self.V_local, tag="owner")
            l2 = float(getattr(self, "l2", 0.0) or 0.0)
            self.kern.adam_dense_rows(self.embed, self.m, self.v, recv, seg, hp, l2)
            if self.lin is not None:
                self.kern.adam_dense_rows(self.lin, self.lin_m, self.lin_v, recv_lin, seg, hp, l2)
            return
        if recv.shape[0] == 0:
            return
        if hasattr(self.kern, "peer_adam") and self.K in (16, 32, 64, 128) and self.world <= 64:
            # every peer's list is de-duplicated: rows are grouped through a [V_local, W] table instead of a sort
            if self.world > 1 and getattr(self, "_peer_tab", None) is None:
                self._peer_tab = torch.zeros(self.V_local * self.world, dtype=torch.int32, device=self.embed.device)
            self.kern.peer_adam(self.embed, self.m, self.v, recv, ctx.recv_ids, ctx.recv_counts, hp, self.lin,
                                getattr(self, "lin_m", None), getattr(self, "lin_v", None), recv_lin,
                                getattr(self, "_peer_tab", None))
            return
        seg = self.kern.segments(ctx.recv_ids, self.V_local, tag="owner")   # peers may ask for the same row
        if self.lin is not None and hasattr(self.kern, "scatter_adam_lin"):
            self.kern.scatter_adam_lin(self.embed, self.m, self.v, recv, self.lin, self.lin_m, self.lin_v, recv_lin, seg, hp)
            return
        self.kern.scatter_adam(self.embed, self.m, self.v, recv, seg, hp)
        if self.lin is not None:
            self.kern.scatter_adam(self.lin, self.lin_m, self.lin_v, recv_lin, seg, hp)


def _reduce_scatter_sum(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    """out = this rank's row block of the sum over ranks of `inp` ([W*B, C] -> [B, C]).  gloo has no
    reduce-scatter: there (functional checks) it is an all-reduce + slice."""
    if dist.get_backend(group) == "gloo":
        h = inp.cpu() if inp.is_cuda else inp.clone()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        r, B = dist.get_rank(group), out.shape[0]
        out.copy_(h[r * B:(r + 1) * B])
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


def allreduce_sum_(flat_grad: torch.Tensor, group=None) -> None:
    """Dense-parameter gradients: one collective over the flat buffer (a few MB — never put
    table-sized tensors through a ring all-reduce on per-link-bound xGMI)."""
    if dist.get_world_size(group) == 1:
        return
    if _host_staged(flat_grad, group):
        h = flat_grad.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        flat_grad.copy_(h)
    else:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)


def rank_average(group=None):
    """`avg(t)`: all-reduce a (small) tensor in place and divide by the world size; returns it.  The hook of the
    global-batch BatchNorm of the data-parallel nets (`layers/dense.py:_SyncBatchNorm`, `TFBatchNorm.sync`)."""
    W = dist.get_world_size(group)

    def avg(t: torch.Tensor) -> torch.Tensor:
        if W > 1:
            allreduce_sum_(t, group)
            t.div_(W)
        return t

    return avg


def sharded_score_topk(kern, users: torch.Tensor, items_local: torch.Tensor, k: int, item_base: int,
                       consumed_ptr=None, consumed_idx=None, filter_flag=None, group=None):
    """Item-sharded full-catalog scoring: local top-k on this rank's slice, all-gather of the
    [B,k] candidates (B·k·12 bytes per rank), k-way merge.  Every rank returns the same result."""
    W = dist.get_world_size(group)
    k_loc = min(k, items_local.shape[0])
    if k_loc == 0:      # a rank whose row range holds no item (node-partitioned graph models)
        s = torch.empty((users.shape[0], 0), dtype=users.dtype, device=users.device)
        i = torch.empty((users.shape[0], 0), dtype=torch.int64, device=users.device)
    else:
        s, i = kern.score_topk(users, items_local, k_loc, consumed_ptr, consumed_idx, filter_flag, item_base)
    if k_loc < k:  # pad short shards with empty slots
        pad_s = torch.full((s.shape[0], k - k_loc), float("-inf"), dtype=s.dtype, device=s.device)
        pad_i = torch.full((s.shape[0], k - k_loc), -1, dtype=i.dtype, device=i.device)
        s, i = torch.cat([s, pad_s], 1), torch.cat([i, pad_i], 1)
    B = s.shape[0]
    all_s = torch.empty((W * B, k), dtype=s.dtype, device=s.device)   # rank-major concatenation
    all_i = torch.empty((W * B, k), dtype=i.dtype, device=i.device)
    _all_gather_into(all_s, s.contiguous(), group=group)
    _all_gather_into(all_i, i.contiguous(), group=group)
    return kern.topk_merge(all_s.view(W, B, k), all_i.view(W, B, k))
