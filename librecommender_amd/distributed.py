"""Multi-GPU behind the model API: when `torch.distributed` is initialised with more than one rank, the
embed models build their row-sharded nets (`nets.ShardedTwoTowerNet`, `nets.graph_nets.ShardedLightGCNNet`) and so do
`DeepFM` (plain sparse columns: `nets.fm_nets.ShardedDeepFMNet`) and `DIN` (pure ids: `nets.feat_nets.ShardedDINNet`), `fit()`
runs the data-parallel step (every rank iterates the SAME seeded loader and takes its contiguous slice of each batch, so
N ranks train on exactly the batches one rank would see), and the exported item embeddings stay SHARDED: rank r keeps the
tower outputs of items [r * per, (r + 1) * per) and `recommend_user` is `parallel.sharded_score_topk` (local fused
score + top-k with `item_base`, all-gather of the [B, k] candidates, merge) — the 100 M x 128 catalogue of BASELINE
cfg 4 never exists on one GPU (`bases/embed_base.py:64-76`, `recommendation/recommend.py:57-78`).

One process per GPU; launch with `python -m torch.distributed.run --nproc-per-node N script.py` and call
`torch.distributed.init_process_group("nccl")` before building the model.  Every rank must make the same API calls in
the same order (collectives inside)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

# Test hook (CPU gloo tests): a kernel provider standing in for `parallel.HipKernels` and the device the nets live on.
# The product leaves both at None: HIP kernels on the rank's GPU, and no CPU execution path.
KERNEL_PROVIDER = None
DEVICE_OVERRIDE: Optional[torch.device] = None
FORCE_WORLD_ONE = False      # tests: take the sharded path under a process group of ONE rank (the N-rank reference run)


def active(group=None):
    """(rank, world) when running under an initialised process group of more than one rank, else None."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    w = dist.get_world_size(group)
    return (dist.get_rank(group), w) if (w > 1 or FORCE_WORLD_ONE) else None


def kernels():
    if KERNEL_PROVIDER is not None:
        return KERNEL_PROVIDER
    from .parallel import HipKernels

    return HipKernels()


def device_for(arg):
    if DEVICE_OVERRIDE is not None:
        return DEVICE_OVERRIDE
    from .bases.base import hip_device

    return hip_device(arg)


def batch_slice(n: int, rank: int, world: int) -> slice:
    """This rank's contiguous share of an n-sample batch; all ranks get the same count (collectives need equal block
    sizes): the last n % world samples of a batch are dropped."""
    per = n // world
    if n % world and (n, world) not in _SLICE_WARNED:          # said once per batch size (the short last batch of an epoch)
        _SLICE_WARNED.add((n, world))
        import warnings

        warnings.warn(f"a batch of {n} samples over {world} ranks: the last {n % world} sample(s) of every such batch are not "
                      f"trained on (equal per-rank blocks); choose a batch size that is a multiple of the world size to use "
                      f"every sample", stacklevel=2)
    return slice(rank * per, (rank + 1) * per)


_SLICE_WARNED = set()


def take(x, sl: slice):
    if x is None:
        return None
    return x[sl]


class ShardedItemEmbeds:
    """Item embeddings of an embed model, sharded by contiguous item-id ranges: this rank holds rows
    [base, base + n_local) of the [n_items, D] matrix (`local`, possibly padded behind n_local) plus the replicated OOV
    row (mean over the real items, `bases/embed_base.py:257-265`)."""

    def __init__(self, local: torch.Tensor, n_items: int, base: int, n_local: int, group=None, kern=None):
        self.local, self.n_items, self.group = local, int(n_items), group
        self.kern = kern or kernels()
        self.base, self.n_local = int(base), int(n_local)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        s = local[: self.n_local].double().sum(dim=0)
        from .parallel import allreduce_sum_

        allreduce_sum_(s, group)
        self.oov = (s / max(n_items, 1)).to(local.dtype)

    @property
    def device(self):
        return self.local.device

    @property
    def shape(self):
        return (self.n_items + 1, self.local.shape[1])

    def topk(self, users: torch.Tensor, k: int, ptr, cidx, flag):
        """Global top-k of `users @ items.T` over the sharded catalogue (consumed ids are GLOBAL item ids)."""
        from .parallel import sharded_score_topk

        loc = self.local[: self.n_local].contiguous()
        return sharded_score_topk(self.kern, users, loc, k, self.base, ptr, cidx, flag, group=self.group)

    def random_topk(self, users: torch.Tensor, n_rec: int, ptr, cidx, flag) -> torch.Tensor:
        """`random_rec=True` (`recommendation/ranking.py:65-73`: n_rec items without replacement with weights
        softmax(score)^0.75 + 1e-8, consumed items excluded, returned in score order) over the SHARDED catalogue, without any
        rank holding [B, N]: (1) every rank folds its block into an online-softmax state, the states are all-gathered and
        combined -> the global log-normaliser; (2) every rank keeps the n_rec largest keys log w + Gumbel of its block
        (Gumbel-top-k: the n_rec largest perturbed keys ARE a draw without replacement with probabilities ~ w); (3) the
        [B, n_rec] candidates are all-gathered and the n_rec largest keys overall win.  Every rank returns the same lists."""
        from .recommendation.recommend import random_rec_local_topk, random_rec_normaliser
        from .parallel import _all_gather_into

        loc = self.local[: self.n_local]
        B, W = users.shape[0], self.world
        m, ssum = random_rec_normaliser(users, loc)
        st = torch.stack([m, ssum], dim=1).contiguous()                       # [B, 2] fp64
        allst = torch.empty((W * B, 2), dtype=st.dtype, device=st.device)
        _all_gather_into(allst, st, group=self.group)
        allst = allst.view(W, B, 2)
        mg = allst[:, :, 0].max(dim=0).values
        sg = (allst[:, :, 1] * torch.exp(allst[:, :, 0] - mg[None, :])).sum(dim=0)   # empty blocks: exp(-inf) * 0 = 0
        lse = mg + torch.log(sg)
        # the ranks' Gumbel streams must be independent even when every rank was seeded alike
        gen = torch.Generator(device=users.device)
        gen.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) + 1_000_003 * (self.rank + 1))
        k_loc, i_loc, s_loc = random_rec_local_topk(users, loc, lse, ptr, cidx, flag, n_rec, self.base, generator=gen)
        pack = torch.cat([k_loc, i_loc.double(), s_loc.double()], dim=1).contiguous()          # [B, 3 n_rec]
        allp = torch.empty((W * B, 3 * n_rec), dtype=pack.dtype, device=pack.device)
        _all_gather_into(allp, pack, group=self.group)
        allp = allp.view(W, B, 3, n_rec).permute(1, 2, 0, 3).reshape(B, 3, W * n_rec)
        top = torch.topk(allp[:, 0], n_rec, dim=1)
        ids = torch.gather(allp[:, 1], 1, top.indices).long()
        sc = torch.gather(allp[:, 2], 1, top.indices)
        order = torch.argsort(sc, dim=1, descending=True)
        return torch.gather(ids, 1, order)

    def rows(self, ids: torch.Tensor) -> torch.Tensor:
        """[n, D] rows of arbitrary global item ids (id n_items = the OOV row): every rank contributes the rows of
        its range, one all-reduce."""
        from .parallel import allreduce_sum_

        ids = ids.long()
        mine = (ids >= self.base) & (ids < self.base + self.n_local)
        out = torch.zeros((ids.numel(), self.local.shape[1]), dtype=self.local.dtype, device=self.local.device)
        out[mine] = self.local[(ids[mine] - self.base)]
        allreduce_sum_(out, self.group)
        oov = ids >= self.n_items
        if bool(oov.any()):
            out[oov] = self.oov
        return out

    def gather(self) -> torch.Tensor:
        """The full [n_items + 1, D] matrix (export / kNN at sizes where that is acceptable)."""
        parts = [None] * self.world
        dist.all_gather_object(parts, (self.base, self.local[: self.n_local].cpu()), group=self.group)
        full = torch.zeros((self.n_items, self.local.shape[1]), dtype=self.local.dtype)
        for base, rows in parts:
            full[base: base + rows.shape[0]] = rows
        return torch.cat([full.to(self.local.device), self.oov.view(1, -1)], dim=0)


def blockwise_tower(net, side: str, n: int, row_offset: int, rank: int, world: int, chunk: int = 1 << 16,
                    sparse_unique=None, sparse_offset: int = 0, dense_unique=None):
    """Tower outputs of ids [rank * per, (rank + 1) * per) of one side, fetched through the sharded tables' lookup
    collective in equal-sized chunks (every rank issues the same number of lookups).  `sparse_unique` [>= n, Fs]: the
    side's stored sparse feature rows (`DataInfo.user_sparse_unique` / `item_sparse_unique`,
    `bases/dyn_embed_base.py:240-269`) — the tower input is then [id row, feature rows...]; `dense_unique` [>= n, n_dense of
    the side]: its stored dense feature values.  -> ([per, D], per)"""
    per = -(-n // world)
    lo = rank * per
    outs = []
    dev = net.device
    feats = None if sparse_unique is None else torch.as_tensor(np.asarray(sparse_unique), device=dev).to(torch.int32)
    dvals = None if dense_unique is None else torch.as_tensor(np.asarray(dense_unique), device=dev, dtype=torch.float32)
    for s in range(0, per, chunk):
        e = min(s + chunk, per)
        ids = torch.arange(lo + s, lo + e, device=dev, dtype=torch.int64).clamp_(max=max(n - 1, 0))   # padded tail: any valid id
        rows = (ids + row_offset).to(torch.int32).view(-1, 1)
        if feats is not None:
            rows = torch.cat([rows, feats[ids] + int(sparse_offset)], dim=1).contiguous()
        outs.append(net.embed(side, rows) if dvals is None else net.embed(side, rows, dense=dvals[ids]))
    out = torch.cat(outs, dim=0)
    valid = (torch.arange(lo, lo + per, device=dev) < n)
    out[~valid] = 0
    return out.contiguous(), per


def all_gather_rows(local: torch.Tensor, n: int, group=None) -> torch.Tensor:
    from .parallel import _all_gather_into

    world = dist.get_world_size(group)
    full = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    _all_gather_into(full, local.contiguous(), group=group)
    return full[:n].contiguous()


def np_take(x, sl):
    return None if x is None else np.asarray(x)[sl]


# ---------------------------------------------------------------------------------------------------------------
# Checkpoints under a process group: row-sharded tables are written PER SHARD (no gather: a 100 M x 128 table never exists
# in one place), the replicated dense parameters + Adam state + BatchNorm moving statistics once by rank 0.  Every rank
# calls `save` / `load` (the embed models' export files need collectives).  A checkpoint written on another world size
# is re-sharded on load (`ShardedFieldTables.load_shards_resharded`).
# ---------------------------------------------------------------------------------------------------------------
def _batch_norms(net):
    from .layers.dense import DenseStack, TFBatchNorm

    out = {}
    for name, obj in vars(net).items():
        if isinstance(obj, TFBatchNorm):
            out[name] = obj
        elif isinstance(obj, DenseStack):
            for k, bn in enumerate([obj.bn_in] + list(obj.bns)):
                if bn is not None:
                    out[f"{name}.{k}"] = bn
    return out


def save_sharded(model, path: str, model_name: str) -> None:
    import json
    import os

    net, group = model.net, getattr(model.net, "group", None)
    rank = dist.get_rank(group)
    os.makedirs(path, exist_ok=True)
    if hasattr(net, "tables"):
        net.tables.save_shard(path, f"{model_name}_tables")
    else:                                   # row-partitioned node table of the graph models
        extra = {"vmax": net.vmax.cpu().numpy()} if getattr(net, "vmax", None) is not None else {}      # AMSGrad state
        np.savez(os.path.join(path, f"{model_name}_nodes_shard{rank}of{dist.get_world_size(group)}.npz"),
                 E=net.E.cpu().numpy(), m=net.m.cpu().numpy(), v=net.v.cpu().numpy(), lo=np.int64(net.lo), hi=np.int64(net.hi),
                 n=np.int64(net.n), world=np.int64(net.world), **extra)
    if rank == 0:
        with open(os.path.join(path, f"{model_name}_hyper_parameters.json"), "w") as f:
            json.dump(model._hparams(), f, separators=(",", ":"), indent=4)
        arrays = {"step": np.int64(getattr(net, "step", 0))}
        P = getattr(net, "P", None)
        if P is not None:
            arrays.update(flat=P.flat.detach().cpu().numpy(), m=P.m.cpu().numpy(), v=P.v.cpu().numpy(),
                          names=np.asarray(list(P.params)))
            for k, bn in _batch_norms(net).items():
                arrays[f"bn::{k}::mean"], arrays[f"bn::{k}::var"] = bn.moving_mean.cpu().numpy(), bn.moving_var.cpu().numpy()
        if getattr(model, "default_recs", None) is not None:
            arrays["default_recs"] = np.asarray(model.default_recs)
        np.savez(os.path.join(path, f"{model_name}_replicated.npz"), **arrays)
    dist.barrier(group)


def _node_shard_files(path, model_name, world):
    """[(rank, world_old, file)] of ONE node-table checkpoint: the set written on the current world size when it is
    complete, else the only other world size present.  Mixed leftovers (two world sizes, neither the current one), a
    missing shard or a mis-named file are errors — the validation `ShardedFieldTables.load_shards_resharded` applies to
    the row-sharded tables (a stale shard would otherwise overlay rows of another checkpoint silently)."""
    import glob
    import os
    import re

    pat = re.compile(re.escape(model_name) + r"_nodes_shard(\d+)of(\d+)\.npz$")
    by_world = {}
    for f in glob.glob(os.path.join(path, f"{model_name}_nodes_shard*of*.npz")):
        m = pat.search(os.path.basename(f))
        if m is None:
            raise ValueError(f"unexpected shard file name {f}")
        by_world.setdefault(int(m.group(2)), {})[int(m.group(1))] = f
    if not by_world:
        raise FileNotFoundError(f"no {model_name}_nodes_shard*of*.npz under {path}")
    if world in by_world and sorted(by_world[world]) == list(range(world)):
        w_old = world
    elif len(by_world) == 1:
        w_old = next(iter(by_world))
    else:
        raise ValueError(f"{path} holds {model_name}_nodes shards of world sizes {sorted(by_world)}: remove the stale set")
    got = by_world[w_old]
    if sorted(got) != list(range(w_old)):
        raise ValueError(f"{model_name}_nodes checkpoint of world size {w_old} under {path} is incomplete: shards {sorted(got)}")
    return [(r, w_old, got[r]) for r in range(w_old)]


def load_sharded(model, path: str, model_name: str) -> None:
    """Into a model whose `build_model()` ran under the CURRENT process group."""
    import glob
    import os

    net, group = model.net, getattr(model.net, "group", None)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if hasattr(net, "tables"):
        name = f"{model_name}_tables"
        if os.path.exists(os.path.join(path, f"{name}_shard{rank}of{world}.npz")):
            net.tables.load_shard(path, name)
        else:
            net.tables.load_shards_resharded(path, name)
    else:
        files = _node_shard_files(path, model_name, world)
        covered = []
        for r_old, w_old, f in files:       # contiguous row ranges: copy the overlap with this rank's range
            with np.load(f) as z:
                lo, hi = int(z["lo"]), int(z["hi"])
                if int(z["n"]) != net.n:
                    raise ValueError(f"{f} holds a {int(z['n'])}-node table, this model has {net.n} nodes")
                if int(z["world"]) != w_old or not 0 <= lo <= hi <= net.n:
                    raise ValueError(f"{f}: stored world size {int(z['world'])} / row range [{lo}, {hi}) do not match the file name")
                covered.append((lo, hi))
                a, b = max(lo, net.lo), min(hi, net.hi)
                if a < b:
                    pairs = [("E", net.E), ("m", net.m), ("v", net.v)]
                    if getattr(net, "vmax", None) is not None and "vmax" in z:
                        pairs.append(("vmax", net.vmax))
                    for key, dst in pairs:
                        dst[a - net.lo: b - net.lo] = torch.from_numpy(z[key][a - lo: b - lo]).to(dst.device)
        end = 0                              # the shards must tile [0, n) exactly: no hole, no overlap of two checkpoints
        for lo, hi in sorted(covered):
            if lo != end:
                raise ValueError(f"{model_name}_nodes shards under {path} leave rows [{end}, {lo}) uncovered or overlap "
                                 f"(ranges {sorted(covered)})")
            end = hi
        if end != net.n:
            raise ValueError(f"{model_name}_nodes shards under {path} end at row {end}, the table has {net.n}")
    with np.load(os.path.join(path, f"{model_name}_replicated.npz")) as z:
        net.step = int(z["step"])
        P = getattr(net, "P", None)
        if P is not None:
            with torch.no_grad():
                P.flat.copy_(torch.from_numpy(z["flat"]))
                P.m.copy_(torch.from_numpy(z["m"]))
                P.v.copy_(torch.from_numpy(z["v"]))
            for k, bn in _batch_norms(net).items():
                bn.moving_mean.copy_(torch.from_numpy(z[f"bn::{k}::mean"]))
                bn.moving_var.copy_(torch.from_numpy(z[f"bn::{k}::var"]))
        if "default_recs" in z:
            model.default_recs = z["default_recs"]
    dist.barrier(group)


def has_sharded_checkpoint(path: str, model_name: str) -> bool:
    import glob
    import os

    return (os.path.exists(os.path.join(path, f"{model_name}_replicated.npz"))
            and bool(glob.glob(os.path.join(path, f"{model_name}_tables_shard*of*.npz"))
                     or glob.glob(os.path.join(path, f"{model_name}_nodes_shard*of*.npz"))))


def load_sharded_single(model, path: str, model_name: str) -> None:
    """A checkpoint written under a process group (`save_sharded`: tables / node tables per shard + the replicated
    parameters) loaded by ONE process without a process group — a model trained on N GPUs served from one (round-3
    advisor finding: such a checkpoint could only be read back under a process group).  The unsharded net uses the same
    global row layout as the sharded one ([user | item | sparse] for the feature / two-tower tables, [users | items] for the
    graph nets), so row r of the table is row r // W of shard r % W (round-robin tables) or row r - lo of the shard whose
    range [lo, hi) holds it (graph nets).  One array of one shard is resident on the host at a time."""
    import glob
    import os
    import re

    net = model.net
    dev = model.device
    t = getattr(net, "tables", None)
    if t is not None and hasattr(t, "embed"):
        name = f"{model_name}_tables"
        files = sorted(glob.glob(os.path.join(path, f"{name}_shard*of*.npz")))
        worlds = {int(re.search(r"of(\d+)\.npz$", f).group(1)) for f in files}
        if len(worlds) != 1 or len(files) != next(iter(worlds)):
            raise ValueError(f"{path}: expected the complete shard set of ONE world size for {name}, found {len(files)} files of "
                             f"world sizes {sorted(worlds)}")
        w_old = worlds.pop()
        V, K = t.embed.shape
        pairs = [("embed", t.embed), ("m", t.m), ("v", t.v)]
        if getattr(t, "lin", None) is not None:
            pairs += [("lin", t.lin), ("lin_m", t.lin_m), ("lin_v", t.lin_v)]
        for r_old in range(w_old):
            with np.load(os.path.join(path, f"{name}_shard{r_old}of{w_old}.npz")) as z:
                if int(z["V"]) != V or int(z["K"]) != K or int(z["world"]) != w_old or int(z["rank"]) != r_old:
                    raise ValueError(f"{name}_shard{r_old}of{w_old}.npz holds a [{int(z['V'])}, {int(z['K'])}] table (rank "
                                     f"{int(z['rank'])} of {int(z['world'])}); this model's table is [{V}, {K}]")
                from .parallel import ShardedFieldTables

                for key, dst in pairs:
                    try:
                        arr = ShardedFieldTables.shard_array(path, name, r_old, w_old, key, z)
                    except FileNotFoundError:
                        raise ValueError(f"{name}_shard{r_old}of{w_old} has no `{key}` (a table without linear weights?)") from None
                    a = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
                    dst[r_old::w_old] = a.reshape(dst[r_old::w_old].shape)
    else:
        files = _node_shard_files(path, model_name, world=-1)
        end = 0
        # (the un-sharded LightGCNNet a single process builds has n_users / n_items and the node table E, no `n`)
        n_nodes = int(getattr(net, "n", net.E.shape[0]))
        for r_old, w_old, f in files:
            with np.load(f) as z:
                lo, hi = int(z["lo"]), int(z["hi"])
                if int(z["n"]) != n_nodes or int(z["world"]) != w_old or lo != end:
                    raise ValueError(f"{f}: node range [{lo}, {hi}) of a {int(z['n'])}-node table does not continue at row {end} "
                                     f"of this {n_nodes}-node model")
                pairs = [("E", net.E), ("m", net.m), ("v", net.v)]
                if getattr(net, "vmax", None) is not None and "vmax" in z:
                    pairs.append(("vmax", net.vmax))
                for key, dst in pairs:
                    dst[lo:hi] = torch.from_numpy(z[key][: hi - lo]).to(dev)
                end = hi
        if end != n_nodes:
            raise ValueError(f"{model_name}_nodes shards under {path} end at row {end}, the table has {n_nodes}")
    with np.load(os.path.join(path, f"{model_name}_replicated.npz")) as z:
        net.step = int(z["step"])
        P = getattr(net, "P", None)
        if P is not None and "flat" in z:
            if "names" in z and list(z["names"]) != list(P.params):
                raise ValueError("the dense parameters of the sharded checkpoint are not those of this model: "
                                 f"{list(z['names'])[:4]}... vs {list(P.params)[:4]}...")
            if z["flat"].shape[0] != P.flat.numel():
                raise ValueError(f"the sharded checkpoint holds {z['flat'].shape[0]} dense parameters, this model {P.flat.numel()}")
            with torch.no_grad():
                P.flat.copy_(torch.from_numpy(z["flat"]))
                P.m.copy_(torch.from_numpy(z["m"]))
                P.v.copy_(torch.from_numpy(z["v"]))
            for k, bn in _batch_norms(net).items():
                if f"bn::{k}::mean" in z:
                    bn.moving_mean.copy_(torch.from_numpy(z[f"bn::{k}::mean"]))
                    bn.moving_var.copy_(torch.from_numpy(z[f"bn::{k}::var"]))
        if "default_recs" in z:
            model.default_recs = z["default_recs"]
