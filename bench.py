#!/usr/bin/env python
"""Benchmark of the MI355X hot path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], SURVEY §8d cfg 2): DeepFM ranking training, 1M users x 1M
items x 200 sparse fields (vocab 50,000 + OOV each => 10,000,200 sparse rows), embed_size 64,
hidden (128,64,32), post-sampling batch 16,384 per GPU, synthetic Zipf(1.05) ids, labels
Bernoulli(0.5).  One "step" = one full training step (gather + FM + MLP forward, loss, backward,
row-wise Adam on the embedding rows, Adam on the dense parameters).  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line (rank 0).  With N > 1 ranks (one per GPU, weak scaling:
the same batch per GPU) the step is the ROW-SHARDED DeepFM (`ShardedDeepFMNet`: tables sharded row-wise,
RCCL all-to-all of de-duplicated ids / rows / row gradients with the exchange plan of the next batch built
beside the current step, one all-reduce of the dense gradients); `--parallel field` selects the
field-partitioned / tensor-parallel alternative of nets/field_parallel.py explicitly (never as a fallback).

`--workload {din,twotower,lightgcn}` prints the same kind of line for the other BASELINE.json configurations at full
size on one GPU (bench_workloads.py); the default (`deepfm`) is the configuration the metric is quoted on.  With
`--gpus N` and no torch.distributed.run environment the script re-launches itself as N ranks on 127.0.0.1;
`--workload twotower --gpus N` (cfg 4: table row-sharded, global in-batch softmax) and `--workload lightgcn --gpus N`
(cfg 5: node table and Laplacian row-partitioned) are the strong-scaling legs of the 8-GPU configurations.

Extra objects in the line:
  roofline      dominant hand-written kernel of the step, algorithmic bytes / HIP-event time
  cpu_baseline  the oracle (PyTorch-CPU restatement of the reference TF graph, TF1 dense Adam)
                timed on the host cores on a bounded sample of the same workload (rank 0, N=1)
  recommend     items-scored/sec of full-catalog top-k scoring (the metric's second half)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3    # f32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0  # bf16 MFMA dense peak (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)

CFG = dict(n_users=1_000_000, n_items=1_000_000, n_sparse_fields=200, vocab=50_000, embed_size=64,
           hidden_units=(128, 64, 32), batch=16_384)


def zipf_ids(rng, vocab, size, a=1.05):
    return ((rng.zipf(a, size=size) - 1) % vocab).astype(np.int32)


def make_batches(cfg, n_batches, seed):
    """Synthetic interaction stream: [n_batches, B, 2+Fs] global-row-free ids + labels."""
    rng = np.random.default_rng(seed)
    B, Fs, vocab = cfg["batch"], cfg["n_sparse_fields"], cfg["vocab"]
    out = []
    off = (np.arange(Fs, dtype=np.int64) * (vocab + 1)).astype(np.int32)
    for _ in range(n_batches):
        users = zipf_ids(rng, cfg["n_users"], B)
        items = zipf_ids(rng, cfg["n_items"], B)
        sparse = zipf_ids(rng, vocab, (B, Fs)) + off
        labels = rng.integers(0, 2, B).astype(np.float32)
        out.append((users, items, sparse, labels))
    return out


def field_row_start(cfg):
    """First global table row of every field (user ids, item ids, sparse field 0..Fs-1) + the end."""
    Fs, vocab = cfg["n_sparse_fields"], cfg["vocab"]
    s_off = cfg["n_users"] + 1 + cfg["n_items"] + 1
    return np.concatenate([[0, cfg["n_users"] + 1, s_off], s_off + (np.arange(Fs) + 1) * (vocab + 1)]).astype(np.int64)


def global_rows(cfg, users, items, sparse):
    """[B, 2+Fs] global table rows of a host batch of `make_batches`."""
    i_off, s_off = cfg["n_users"] + 1, cfg["n_users"] + 1 + cfg["n_items"] + 1
    return np.concatenate([users.reshape(-1, 1).astype(np.int64), items.reshape(-1, 1).astype(np.int64) + i_off,
                           sparse.astype(np.int64) + s_off], axis=1).astype(np.int32)


def device_batch_maker(cfg, dev, seed):
    """One batch of the synthetic interaction stream per call, drawn ON THE DEVICE (ids Zipf(1.05) over each
    vocabulary — the exact law of `make_batches`, `bench_workloads.zipf_ids_device` — labels Bernoulli(0.5)):
    (global rows [B, 2 + Fs] int32, labels [B] f32)."""
    from bench_workloads import zipf_ids_device

    g = torch.Generator(device=dev).manual_seed(seed)
    B, Fs, vocab = cfg["batch"], cfg["n_sparse_fields"], cfg["vocab"]
    i_off, s_off = cfg["n_users"] + 1, cfg["n_users"] + 1 + cfg["n_items"] + 1
    off = (torch.arange(Fs, device=dev, dtype=torch.int64) * (vocab + 1) + s_off).to(torch.int32)

    def one():
        users = zipf_ids_device(B, cfg["n_users"], g, dev)
        items = zipf_ids_device(B, cfg["n_items"], g, dev) + i_off
        sparse = zipf_ids_device(B * Fs, vocab, g, dev).view(B, Fs) + off[None, :]
        labels = torch.randint(0, 2, (B,), device=dev, generator=g).float()
        return (torch.cat([users[:, None], items[:, None], sparse], dim=1).contiguous(), labels)
    return one


def host_batch(cfg, batch):
    """A device batch of `device_batch_maker` in the (users, items, sparse, labels) numpy form of `make_batches`."""
    idx, labels = batch[0].cpu().numpy(), batch[1].cpu().numpy()
    i_off, s_off = cfg["n_users"] + 1, cfg["n_users"] + 1 + cfg["n_items"] + 1
    return (idx[:, 0].copy(), idx[:, 1] - i_off, idx[:, 2:] - s_off, labels)


def algorithmic_bytes_per_sample(F, K):
    """SURVEY §8(d) cfg 2."""
    fwd = F * (K * 4) + F * 4 + F * 4                 # rows + linear + ids            = 53.3 KB
    bwd_scatter = 2 * F * (K * 4 + 4)                 # RMW of the touched rows        = 105 KB
    adam = 4 * F * (K * 4)                            # m, v read+write                = 207 KB
    return dict(fwd=fwd, bwd=bwd_scatter, bwd_adam=bwd_scatter + adam)


class ClockProbe:
    """Mean shader clock over a timed region: `lr_clock_probe` stores (s_memtime, s_memrealtime) per COMPUTE UNIT before and
    after; per CU seen both times MHz = d(shader ticks) / d(real-time ticks) x the real-time counter's rate (calibrated against the
    host clock over 50 ms at construction); the MEDIAN over the CUs is reported when at least half of them lie within 5 % of it
    and it is a plausible shader clock, else None (idle chip: the counters stop with the clock).  Boxes of this pool differ by
    several % in step time: the line carries the clock the part actually sustained, so the spread is attributable."""

    def __init__(self, dev):
        from librecommender_amd import _lib, ops

        self.ops = ops
        self.n = int(_lib.load().lr_clock_probe_slots())
        self.buf = torch.zeros((2, self.n, 2), dtype=torch.int64, device=dev)     # [mark][CU slot][memtime, realtime]
        self.mark(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        time.sleep(0.05)
        self.mark(1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        b = self.buf.cpu()
        ok = (b[0, :, 1] > 0) & (b[1, :, 1] > b[0, :, 1])
        self.rt_hz = float((b[1, :, 1] - b[0, :, 1])[ok].double().median()) / dt if dt > 0 and bool(ok.any()) else 0.0
        self.buf.zero_()

    def mark(self, i):
        self.ops._call("lr_clock_probe", self.buf[i].data_ptr(), self.ops._stream())

    def mhz(self):
        torch.cuda.synchronize()
        b = self.buf.cpu()
        d_sh, d_rt = (b[1, :, 0] - b[0, :, 0]).double(), (b[1, :, 1] - b[0, :, 1]).double()
        ok = (b[0, :, 1] > 0) & (b[1, :, 1] > 0) & (d_rt > 0)
        if int(ok.sum()) < 32 or self.rt_hz <= 0:
            return None
        r = (d_sh[ok] / d_rt[ok]) * self.rt_hz / 1e6
        med = float(r.median())
        agree = int(((r - med).abs() <= 0.05 * abs(med)).sum())
        return round(med, 1) if agree * 2 >= int(ok.sum()) and 300.0 < med < 3200.0 else None


def pmc_traffic(kernel, workload="deepfm"):
    """HBM-side bytes per launch measured with rocprofv3 PMC for THIS workload (committed under
    profiles/; None for other shapes / kernels)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):   # the newest table that holds the entry
        try:
            with open(os.path.join(here, name)) as fh:
                tr = json.load(fh).get(workload, {}).get(kernel, {}).get("traffic_bytes")
        except OSError:
            continue
        if tr:
            return tr
    return None


PROFILES_TIMES = "r06_kernel_times.json"
# C-ABI entry point behind each arithmetic of ops.score_topk, and the words the bench line says about it
TOPK_ENTRY = {"filter": "lr_score_topk_filter_f32", "filter_f32_chain": "lr_score_topk_filter_f32",
              "split_bf16": "lr_score_topk_sb_f32", "f32_chain": "lr_score_topk_f32"}
TOPK_ARITH_NOTE = {
    "filter": "filtered: a one-term bf16 MFMA pass ranks items by an upper bound of the exact score (approx + 0.004 |u| |i|, the term "
              "inside the MFMA chain) and keeps k' = 2k + 56 candidates per user, their scores are recomputed in f32, a user is "
              "certified when its k-th exact score exceeds the k'-th bound (no outside item can enter the top k), uncertified users "
              "are re-run by the exact split-bf16 kernel; returned scores are f32 dot products for any data (both exact kernels are "
              "timed beside it)",
    "filter_f32_chain": "filtered (see `filter`), exact pass = the f32 fma chain",
    "split_bf16": "f32 scores as six-term split-bf16 MFMA products with f32 accumulation (item planes split on the fly; as close to "
                  "fp64 as the f32 fma chain, which is timed beside it)",
    "f32_chain": "exact k-ordered f32 fma chain on the f32 MFMA pipe"}


def profiles_ref(kernel, workload="deepfm"):
    """Mean duration (ms) of one call of a C-ABI entry point in the committed `rocprofv3 --kernel-trace --stats` summary of
    THIS tree's bench command (profiles/r05_<workload>_kernel_trace.md, condensed by scripts/profiles_from_run.py)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        with open(os.path.join(here, PROFILES_TIMES)) as fh:
            return json.load(fh).get(workload, {}).get(kernel, {}).get("avg_ms")
    except OSError:
        return None


def with_profiles(d, kernel, workload):
    """Beside the live HIP-event numbers of a `roofline` object: the same fraction from the committed rocprof average, so that
    the line can be recomputed from profiles/ alone (boxes differ by several per cent; the committed one does not move)."""
    avg = profiles_ref(kernel, workload)
    per_launch = d.get("algorithmic_bytes_per_launch", d.get("flops_per_launch"))
    d["profiles_avg_ms"] = avg
    d["frac_from_profiles"] = None
    if avg and per_launch:
        scale = 1e9 if d["bound"] == "hbm" else 1e12
        d["frac_from_profiles"] = round(per_launch / (avg * 1e-3) / scale / d["peak"], 4)
        d["profiles_source"] = f"profiles/{PROFILES_TIMES} [{workload}][{kernel}] (rocprofv3 --kernel-trace --stats of this tree's bench command)"
        if d.get("traffic"):
            d["frac_by_traffic_from_profiles"] = round(d["traffic"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return d


def make_parallel_net(args, cfg, dev, world, probe_batch, n_rows):
    """Multi-GPU DeepFM.  Default (`--parallel row`, north_star's scheme): batch data-parallel,
    tables row-sharded round-robin, RCCL all-to-all of the de-duplicated ids / rows / row gradients, one
    all-reduce of the dense gradients.  `--parallel field` is the named alternative for many-field
    models (fields partitioned, first MLP layer tensor-parallel, nets/field_parallel.py); it is never
    chosen silently and a failure of it is an error, not a switch of scheme."""
    Fs, K = cfg["n_sparse_fields"], cfg["embed_size"]
    if args.parallel == "field":
        from librecommender_amd.nets.field_parallel import FieldParallelDeepFMNet

        net = FieldParallelDeepFMNet(field_row_start(cfg), embed_size=K, hidden_units=cfg["hidden_units"], lr=1e-3, epsilon=1e-5,
                                     seed=42, device=dev)
        return net, (f"fields partitioned {world}-way (tables, fused backward + Adam local), first MLP layer tensor-parallel "
                     f"(reduce-scatter / all-gather of [global batch, 128]), all-reduce of the FM sums and of the replicated "
                     f"dense gradients, global-batch BatchNorm; dp{world} for the rest")
    from librecommender_amd.nets import ShardedDeepFMNet

    net = ShardedDeepFMNet(n_rows, Fs, embed_size=K, hidden_units=cfg["hidden_units"], lr=1e-3, epsilon=1e-5,
                           seed=42, device=dev, field_row_start=field_row_start(cfg))
    return net, (f"dp{world} batch + tables row-sharded {world}-way (RCCL all-to-all of de-duplicated ids / rows / row gradients, "
                 f"exchange plans prefetched one step ahead, all-reduce of dense grads)")


def bench_train(args, rank, world, dev):
    from librecommender_amd import ops
    from librecommender_amd.nets import DeepFMNet

    cfg = dict(CFG)
    if args.small:
        cfg.update(n_users=50_000, n_items=50_000, n_sparse_fields=20, vocab=2_000, batch=2_048)
    Fs, K, B = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"]
    mlp_dtype = torch.bfloat16 if args.mlp_dtype == "bf16" else torch.float32
    n_rows = cfg["n_users"] + 1 + cfg["n_items"] + 1 + Fs * (cfg["vocab"] + 1)
    if world == 1 and not args.force_sharded:
        net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), Fs, embed_size=K,
                        hidden_units=cfg["hidden_units"], lr=1e-3, epsilon=1e-5, seed=42, device=dev,
                        mlp_dtype=mlp_dtype, sparse_offsets=np.arange(Fs) * (cfg["vocab"] + 1),
                        fused_l1=not args.unfused)
    from bench_workloads import Pool

    pool = Pool(device_batch_maker(cfg, dev, seed=42 + rank))       # a fresh batch every step (never trained on twice)
    first = [pool.peek(k) for k in range(12)]                         # the first batches: also the CPU baseline's sample
    parallelism = "single"
    if world > 1 or args.force_sharded:
        net, parallelism = make_parallel_net(args, cfg, dev, world, first[0], n_rows)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = bool(world == 1 and not args.force_sharded and not args.no_graph and getattr(net, "fused_l1", False))
    if graphed:
        net.enable_graph(True)                 # first 2 steps eager, third captured, then replays
    row_sharded = (world > 1 or args.force_sharded) and args.parallel == "row"

    def one_step():
        cur = pool.next()
        if row_sharded:      # the next (resident) batch's exchange plan is built beside this step
            return net.train_step(*cur, next_idx=pool.peek(0)[0])
        return net.train_step(*cur)

    n_warm = max(args.warmup, 4 if graphed else 0)
    pool.ensure(n_warm + args.steps + min(args.steps, 10) + 2)        # drawn before the timed region
    for s in range(n_warm):
        one_step()
    timed = ("lr_fm_embed_fwd_f32", "lr_fm_embed_bwd_adam_f32", "lr_fm_embed_bwd_rows_f32",
             "lr_segments_build", "lr_embed_scatter_adam_f32", "lr_embed_gather_f32", "lr_adam_dense_f32",
             "lr_deepfm_l1_fwd_f32", "lr_deepfm_l1_wgrad_f32", "lr_deepfm_l1_dgrad_f32", "lr_fm_rows_adam_f32",
             "lr_deepfm_l1_fwd_sb_f32", "lr_deepfm_l1_wgrad_sb_f32", "lr_deepfm_l1_dgrad_sb_f32", "lr_deepfm_l1_sb_pack",
             "lr_deepfm_l1_sb_gz_pack", "lr_mlp_tail3_f32", "lr_reduce_partials_multi_f32", "lr_deepfm_l1_fold_stats_f32",
             "lr_deepfm_l1_fold_bias_f32", "lr_deepfm_l1_fold_bwd_f32", "lr_reduce_partials_f32", "lr_deepfm_l1_fold_stats_bias_f32",
             "lr_segments_build_fields", "lr_fm_field_stats_f32", "lr_deepfm_l1_pack_f32", "lr_idx_transpose_i32",
             "lr_fm_rows_grad_f32", "lr_fm_field_stats_slots_f32", "lr_embed_scatter_adam_lin_f32")
    if not graphed:
        ops.TIMER.enable(*timed)
    clk = ClockProbe(dev)
    clk.mark(0)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    clk.mark(1)
    ops.TIMER.disable()
    final_loss = float(loss)
    from librecommender_amd.layers.tail import check_all as _tail_check

    _tail_check()           # a one-launch tail that gave up on a grid barrier raises here (its losses are NaN)
    clock_mhz = clk.mhz()
    if world > 1:
        tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_note = "HIP events around every C-ABI launch of the timed steps"
    if graphed:
        # events cannot be read back from inside a replayed graph: the same kernels are timed on the same
        # batches in a short eager pass AFTER the timed region (not part of `value`)
        net.enable_graph(False)
        ops.TIMER.enable(*timed)
        for s in range(min(args.steps, 10)):
            one_step()
        torch.cuda.synchronize()
        ops.TIMER.disable()
        kernel_note = (f"HIP events around every C-ABI launch in {min(args.steps, 10)} eager steps run after the timed "
                       f"region (the timed steps are hipGraph replays of the same launches)")
    ms = dt / args.steps * 1e3
    l1_arith = getattr(net, "l1_arith", "f32_chain") if getattr(net, "fused_l1", False) or row_sharded else "f32_chain"
    kern = ops.TIMER.summary()
    F = 2 + Fs
    H1 = cfg["hidden_units"][0]
    ab = algorithmic_bytes_per_sample(F, K)
    # per launch: algorithmic HBM bytes (SURVEY 8d convention: every position counted as its own row) for
    # the gather / scatter / Adam kernels, flops for the kernels that sit on the f32 MFMA pipe
    hbm = {"lr_fm_embed_fwd_f32": ab["fwd"] * B, "lr_fm_embed_bwd_adam_f32": ab["bwd_adam"] * B,
           "lr_fm_embed_bwd_rows_f32": ab["bwd"] * B, "lr_fm_rows_adam_f32": ab["bwd_adam"] * B,
           "lr_fm_rows_grad_f32": ab["bwd"] * B}
    l1_flops = 2.0 * B * F * K * H1
    mfma = {"lr_deepfm_l1_fwd_f32": l1_flops, "lr_deepfm_l1_wgrad_f32": l1_flops, "lr_deepfm_l1_dgrad_f32": l1_flops}
    # the split-bf16 forms issue SIX bf16 MFMA products per f32 product: priced against the bf16 MFMA peak by the flops the
    # pipe actually executes, with the f32-equivalent rate beside it
    mfma_sb = {"lr_deepfm_l1_fwd_sb_f32": l1_flops, "lr_deepfm_l1_wgrad_sb_f32": l1_flops, "lr_deepfm_l1_dgrad_sb_f32": l1_flops}
    kinfo = {}
    for name, (n, mean_ms) in kern.items():
        kinfo[name] = {"launches": n, "mean_ms": round(mean_ms, 4)}
        if name in hbm:
            kinfo[name]["algorithmic_GBps"] = round(hbm[name] / (mean_ms * 1e-3) / 1e9, 1)
            kinfo[name]["frac_hbm_peak"] = round(hbm[name] / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if name in mfma:
            kinfo[name]["TFLOPs"] = round(mfma[name] / (mean_ms * 1e-3) / 1e12, 2)
            kinfo[name]["frac_mfma_f32_peak"] = round(mfma[name] / (mean_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
        if name in mfma_sb:
            kinfo[name]["f32_equivalent_TFLOPs"] = round(mfma_sb[name] / (mean_ms * 1e-3) / 1e12, 2)
            kinfo[name]["bf16_mfma_TFLOPs"] = round(6 * mfma_sb[name] / (mean_ms * 1e-3) / 1e12, 1)
            kinfo[name]["frac_mfma_bf16_peak"] = round(6 * mfma_sb[name] / (mean_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)
            kinfo[name]["gathered_or_written_GBps"] = round(B * F * K * 4 / (mean_ms * 1e-3) / 1e9, 1)   # the 847 MB of rows
    sum_kernel_ms = sum(m for _, m in kern.values())      # one launch of each per step

    full_size = not (args.small or world > 1 or args.force_sharded)
    distinct_rows = float(np.mean([torch.unique(b[0]).numel() for b in first[:4]])) if full_size else None

    def prof(d, name):
        return with_profiles(d, name, "deepfm") if full_size else d

    def roof(name):
        mean_ms = kern[name][1]
        if name in hbm:
            a = hbm[name] / (mean_ms * 1e-3) / 1e9
            tr = None if (args.small or world > 1) else pmc_traffic(name)
            d = {"kernel": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK_GBS, 4), "traffic": tr,
                 "traffic_source": "rocprofv3 PMC pass committed under profiles/ (not this run)",
                 "algorithmic_bytes_per_launch": hbm[name], "mean_launch_ms": round(mean_ms, 4),
                 "convention": "SURVEY 8(d): every position counted as its own row (the kernel de-duplicates rows, so "
                               "the bytes that cross the fabric are fewer: see achieved_by_traffic)"}
            if tr:      # what the memory system actually moved / time: the fabric rate, free of the convention
                d["achieved_by_traffic"] = round(tr / (mean_ms * 1e-3) / 1e9, 1)
                d["frac_by_traffic"] = round(tr / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if name in ("lr_fm_rows_adam_f32", "lr_fm_embed_bwd_adam_f32") and distinct_rows:
                # what the kernel cannot avoid once rows are de-duplicated: every position's row gradient read once, every
                # DISTINCT row's (row, m, v) read and written once (K values + the linear weight each)
                dmin = (B * F + 6.0 * distinct_rows) * (K * 4 + 4)
                d["dedup_min_bytes_per_launch"] = int(dmin)
                d["distinct_rows_per_batch"] = int(distinct_rows)
                d["positions_per_batch"] = B * F
                d["frac_dedup_min"] = round(dmin / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            return prof(d, name)
        if name in mfma_sb:
            a = 6 * mfma_sb[name] / (mean_ms * 1e-3) / 1e12
            return prof({"kernel": name, "bound": "mfma", "achieved": round(a, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(a / MFMA_BF16_PEAK_TF, 4), "traffic": None if (args.small or world > 1) else pmc_traffic(name),
                    "traffic_source": "rocprofv3 PMC pass committed under profiles/ (not this run)",
                    "flops_per_launch": 6 * mfma_sb[name], "f32_equivalent_flops_per_launch": mfma_sb[name],
                    "mean_launch_ms": round(mean_ms, 4),
                    "note": "six bf16 MFMA products per f32 product (split-bf16, f32 accumulate): flops the pipe executes"}, name)
        a = mfma[name] / (mean_ms * 1e-3) / 1e12
        return prof({"kernel": name, "bound": "mfma", "achieved": round(a, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(a / MFMA_F32_PEAK_TF, 4), "traffic": None if (args.small or world > 1) else pmc_traffic(name),
                "traffic_source": "rocprofv3 PMC pass committed under profiles/ (not this run)",
                "flops_per_launch": mfma[name], "mean_launch_ms": round(mean_ms, 4)}, name)

    # dominant hand-written kernel of the step (longest mean launch among those with a roofline)
    dom = max((n for n in kern if n in hbm or n in mfma or n in mfma_sb), key=lambda n: kern[n][1])
    roofline = roof(dom)
    scatter = next((n for n in ("lr_fm_rows_adam_f32", "lr_fm_embed_bwd_adam_f32", "lr_fm_rows_grad_f32",
                                "lr_fm_embed_bwd_rows_f32") if n in kern), None)
    result = {
        "metric": "train samples/sec", "value": round(B * world * args.steps / dt, 1),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (split-bf16 x6 MFMA products, f32 accumulate)" if l1_arith == "split_bf16" else "f32")
                 if args.mlp_dtype == "fp32" else "f32 tables+FM, bf16 MLP GEMMs",
        "data": "synthetic",
        "config": {"workload": "DeepFM ranking train step, 1M users x 1M items x 200 sparse fields "
                               "(10,000,200 sparse rows), embed_size=64, hidden=(128,64,32), "
                               "Zipf(1.05) ids" if not args.small else "DeepFM small (smoke)",
                   "per_gpu_batch": B, "global_batch": B * world, "fields": F, "embed_size": K,
                   "table_rows": n_rows, "optimizer": "row-wise Adam on the touched embedding rows (TF1 moves every row: equal at step 1, "
                                "diverges afterwards; dense_adam=True reproduces TF1) + dense Adam (MLP)",
                   "first_layer": ("lookup fused with the first Dense layer; " + (
                                       "its three contractions as six-term split-bf16 MFMA products with f32 accumulation (every f32 operand split "
                                       "exactly into three bf16 values; as close to fp64 as the f32 fma chain: tests/test_l1_split_bf16_gpu.py; the "
                                       "exact f32 chain stays selectable: LIBRECO_L1_ARITH=f32_chain, timed beside as f32_chain_ms_per_step)"
                                       if l1_arith == "split_bf16" else "f32 MFMA (exact f32 fma chain)"))
                                  if (getattr(net, "fused_l1", False) or getattr(net, "field_row_start", None) is not None)
                                  else "materialised deep_embed + library GEMM",
                   "parallelism": parallelism, "final_loss": round(final_loss, 5),
                   "stream": "a fresh batch every step, drawn on the device before the timed region (exact Zipf(1.05) ids, "
                             "Bernoulli(0.5) labels); no batch is trained on twice",
                   "launch": "one hipGraph replay per step" if graphed else "eager launches",
                   "shader_clock_mhz": clock_mhz, "realtime_counter_mhz": round(clk.rt_hz / 1e6, 2)},
        "roofline": roofline, "kernels": kinfo, "sum_kernel_ms": round(sum_kernel_ms, 4),
        "kernel_timing": kernel_note,
    }
    if scatter is not None and scatter != dom:      # the gather+scatter-add+Adam kernel against the HBM roof
        result["roofline_scatter"] = roof(scatter)
    # whole step against both roofs (SURVEY 8d cfg 2): 365 KB/sample of gather + scatter + Adam traffic and the
    # first layer's three B x (F K) x H1 contractions + the tail's small ones
    step_bytes = float(ab["fwd"] + ab["bwd_adam"]) * B
    hid = cfg["hidden_units"]
    tail_fl = sum(2.0 * B * a_ * b_ * 3 for a_, b_ in zip(hid[:-1], hid[1:]))
    step_flops = 3 * l1_flops + tail_fl
    result["roofline_step"] = {
        "algorithmic_bytes_per_step": step_bytes, "hbm_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1),
        "frac_hbm_peak": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "flops_per_step": step_flops, "TFLOPs": round(step_flops / (ms * 1e-3) / 1e12, 2),
        "frac_mfma_f32_peak": round(step_flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
        "serial_ideal_ms": round(step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3 + step_flops / (MFMA_F32_PEAK_TF * 1e12) * 1e3, 4),
        "note": "the HBM-bound kernels (row update, statistics, segment build) and the MFMA-bound first-layer kernels run "
                "back to back: the step's floor is the SUM of the two ideals"}
    if args.steady_seconds > 0 and world == 1:      # >= 1 s of steady-state replays next to the driver's short region
        n = max(args.steps, int(args.steady_seconds / max(dt / args.steps, 1e-6)) + 1)
        pool.ensure(n + 6)
        if graphed:
            net.enable_graph(True)
            for s in range(4):
                one_step()
        barrier()
        t1 = time.perf_counter()
        for s in range(n):
            loss_ss = one_step()
        barrier()
        result["steady_state"] = {"steps": n, "ms_per_step": round((time.perf_counter() - t1) / n * 1e3, 4),
                                  "final_loss": round(float(loss_ss), 5), "distinct_batches": pool.cursor}
        result["steady_ms_per_step"] = result["steady_state"]["ms_per_step"]
        result["steady_steps"] = n
        # (the driver's record keeps `config`, `roofline` and `cpu_baseline` whole and only the NAMES of other keys)
        result["config"]["steady_ms_per_step"] = result["steady_ms_per_step"]
        result["config"]["steady_steps"] = n
    host = [host_batch(cfg, b) for b in first]
    del net
    return result, cfg, host


def bench_f32_chain(args, cfg, dev):
    """The same step with the first layer on the EXACT f32 fma chain (`ops.set_l1_arith("f32_chain")`: csrc/deepfm_l1.hip,
    rounds 1-4's arithmetic), hipGraph replays on fresh batches — timed beside the default so the line shows what the
    split-bf16 products buy."""
    from bench_workloads import Pool
    from librecommender_amd import ops
    from librecommender_amd.nets import DeepFMNet

    torch.cuda.empty_cache()
    Fs, K, B = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"]
    prev = ops.set_l1_arith("f32_chain")
    try:
        net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), Fs, embed_size=K, hidden_units=cfg["hidden_units"],
                        lr=1e-3, epsilon=1e-5, seed=42, device=dev, sparse_offsets=np.arange(Fs) * (cfg["vocab"] + 1))
        assert net.l1_arith == "f32_chain"
    finally:
        ops.set_l1_arith(prev)
    pool = Pool(device_batch_maker(cfg, dev, seed=4242))
    n = max(args.steps, 20)
    pool.ensure(n + 8)
    if not args.no_graph:
        net.enable_graph(True)
    for _ in range(6):
        net.train_step(*pool.next())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss = net.train_step(*pool.next())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    out = {"ms_per_step": round(ms, 4), "steps": n, "value": round(B / ms * 1e3, 1), "unit": "samples/s", "final_loss": round(float(loss), 5),
           "first_layer": "v_mfma_f32_32x32x2_f32: exact k-ordered f32 fma chain (LIBRECO_L1_ARITH=f32_chain)"}
    del net
    torch.cuda.empty_cache()
    return out


def bench_dense_adam(args, cfg, host, dev):
    """The same training step with TF1's optimiser semantics (`dense_adam=True`: every row of every table decays its
    moments and moves every step, training/tf_trainer.py:120) — what the CPU port beside it computes.  The default
    line above uses row-wise Adam on the touched rows (equal at step 1, cheaper afterwards).  Same fused step (lookup + first
    layer, hand-written tail, one hipGraph replay per step); the row update is replaced by the per-row gradient kernel and ONE
    streaming pass over both tables (`lr_adam_dense_rows_dc_f32`), which has a roofline of its own."""
    from librecommender_amd import ops
    from librecommender_amd.nets import DeepFMNet

    torch.cuda.empty_cache()
    Fs, K, B = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"]
    net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), Fs, embed_size=K, hidden_units=cfg["hidden_units"],
                    lr=1e-3, epsilon=1e-5, seed=42, device=dev, sparse_offsets=np.arange(Fs) * (cfg["vocab"] + 1),
                    dense_adam=True)
    fused = bool(getattr(net, "fused_l1", False))
    batches = []
    for users, items, sparse, labels in host[:8]:
        batches.append((torch.from_numpy(global_rows(cfg, users, items, sparse)).to(dev).contiguous(),
                        torch.from_numpy(labels).to(dev)))
    graphed = fused and not args.no_graph
    if graphed:
        net.enable_graph(True)
    for s_ in range(5):
        net.train_step(*batches[s_ % len(batches)])
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for s_ in range(n):
        net.train_step(*batches[s_ % len(batches)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    res = {"value": round(B / ms * 1e3, 1), "unit": "samples/s", "ms_per_step": round(ms, 4), "steps": n,
           "semantics": "TF1 dense Adam over all table rows (reference-exact optimiser); "
                        + ("fused first layer + hand-written tail, " + ("one hipGraph replay per step" if graphed else "eager launches")
                           if fused else "unfused first layer, eager launches")}
    if fused:       # the table pass on its own: HIP events around the launches of a few eager steps
        net.enable_graph(False)
        names = ("lr_adam_dense_rows_f32", "lr_adam_dense_rows_dc_f32", "lr_fm_rows_grad_compact_f32")
        ops.TIMER.enable(*names)
        for s_ in range(5):
            net.train_step(*batches[s_ % len(batches)])
        torch.cuda.synchronize()
        ops.TIMER.disable()
        kern = ops.TIMER.summary()
        res["kernels"] = {k: {"launches": c, "mean_ms": round(m_, 4)} for k, (c, m_) in kern.items()}
        name = next((k for k in ("lr_adam_dense_rows_f32", "lr_adam_dense_rows_dc_f32") if k in kern), None)
        if name is not None:
            V = net.tables.V
            distinct = float(np.mean([torch.unique(b[0]).numel() for b in batches[:4]]))
            # every row: (w, m, v) read + written (K values + the linear weight each) and its slot word; the batch's rows: their gradient
            nbytes = V * (6.0 * (K * 4 + 4) + 4) + distinct * (K * 4 + 4)
            mean_ms = kern[name][1]
            a = nbytes / (mean_ms * 1e-3) / 1e9
            d = {"kernel": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None if args.small else pmc_traffic(name, "dense_adam"),
                 "algorithmic_bytes_per_launch": int(nbytes), "mean_launch_ms": round(mean_ms, 4), "table_rows": int(V),
                 "note": "one pass over (row, m, v) of the embedding and the linear table + the slot word of every row + the compact "
                         "per-row gradients of the batch (the mark / clear kernels of the slot words are inside the timed call)"}
            res["roofline"] = d if args.small else with_profiles(d, name, "dense_adam")
    del net
    torch.cuda.empty_cache()
    return res


def bench_cpu_baseline(cfg, host, seconds_budget=25.0):
    """Oracle (reference TF graph restated in PyTorch-CPU, TF1 dense Adam) on the host cores."""
    from oracle.models_torch import DeepFMOracle

    Fs, K, vocab = cfg["n_sparse_fields"], cfg["embed_size"], cfg["vocab"]
    U, N, S = cfg["n_users"] + 1, cfg["n_items"] + 1, Fs * (vocab + 1)
    g = torch.Generator().manual_seed(0)
    hidden = cfg["hidden_units"]
    W = {"user_embeds_var": torch.rand((U, K), generator=g) * 0.02 - 0.01,
         "item_embeds_var": torch.rand((N, K), generator=g) * 0.02 - 0.01,
         "sparse_embeds_var": torch.rand((S, K), generator=g) * 0.02 - 0.01,
         "user_linear_var": torch.zeros((U, 1)), "item_linear_var": torch.zeros((N, 1)),
         "sparse_linear_var": torch.zeros(S),
         "linear/kernel": torch.rand((2 + Fs, 1), generator=g) * 0.1, "linear/bias": torch.zeros(1),
         "out/kernel": torch.rand((1 + K + hidden[-1], 1), generator=g) * 0.1, "out/bias": torch.zeros(1)}
    d = (2 + Fs) * K
    W.update({"mlp/bn_in/gamma": torch.ones(d), "mlp/bn_in/beta": torch.zeros(d),
              "mlp/bn_in/moving_mean": torch.zeros(d), "mlp/bn_in/moving_var": torch.ones(d)})
    for i, h in enumerate(hidden, start=1):
        W[f"mlp/mlp_layer{i}/kernel"] = (torch.rand((d, h), generator=g) - 0.5) * 0.05
        W[f"mlp/mlp_layer{i}/bias"] = torch.zeros(h)
        if i != len(hidden):
            W.update({f"mlp/bn{i}/gamma": torch.ones(h), f"mlp/bn{i}/beta": torch.zeros(h),
                      f"mlp/bn{i}/moving_mean": torch.zeros(h), f"mlp/bn{i}/moving_var": torch.ones(h)})
        d = h
    model = DeepFMOracle(W, hidden, lr=1e-3, epsilon=1e-5, dtype=torch.float32)
    del W
    cores = torch.get_num_threads()
    steps, t_total = 0, 0.0
    B = cfg["batch"]
    # 10 timed full-size steps unless they would take more than ~100 s of host time
    while steps < 11 and (t_total < seconds_budget or (steps < 11 and t_total < 100.0)):
        users, items, sparse, labels = host[steps % len(host)]
        args = (torch.from_numpy(users).long(), torch.from_numpy(items).long(),
                torch.from_numpy(sparse).long(), torch.from_numpy(labels))
        t0 = time.perf_counter()
        model.train_step(*args)
        dt = time.perf_counter() - t0
        if steps > 0 or dt > seconds_budget:  # first step pays allocation / page faults
            t_total += dt
        steps += 1
    timed = max(steps - 1, 1)
    return {"value": round(B * timed / max(t_total, 1e-9), 1), "unit": "samples/s", "cores": cores,
            "kind": "port", "host_logical_cpus": os.cpu_count(),
            "cores_note": "cores = torch.get_num_threads(), the threads the oracle's torch ops actually ran on (torch's default: one "
                          "per physical core); host_logical_cpus = os.cpu_count() counts SMT siblings",
            "sample": f"{timed} full-size training steps (B={B}, same tables/ids) of the PyTorch-CPU "
                      f"oracle restatement of the reference TF graph incl. TF1 dense Adam; first step untimed"}


def bench_recommend_cpu_baseline(seconds_budget=12.0):
    """CPU baseline of the recommend leg: `recommend_from_embedding` + `rank_recommendations`
    (recommendation/recommend.py:57-78, ranking.py:10-56) on the host cores — the reference's own
    functions where the checkout is present (build container), the numpy restatement (oracle/ops_np.py)
    on the GPU box where it is not.  Bounded sample: 128-dim embeddings, 1 M items, users in chunks of 64
    until the budget is spent; items-scored/s is size-independent for this O(B*N*D) + O(B*N) path, the
    12.5 M-item figure is a linear extrapolation."""
    from oracle import ref_loader

    rng = np.random.default_rng(0)
    N, D, k, chunk = 1_000_000, 128, 100, 64
    I = rng.standard_normal((N, D)).astype(np.float32)
    consumed = {u: sorted(rng.integers(0, N, 50).tolist()) for u in range(chunk)}
    kind = "port"
    if ref_loader.available():
        try:
            ref_loader.load()
            from libreco.recommendation.ranking import rank_recommendations as ref_rank

            def run(U, users):
                preds = U @ I.T                                              # recommend.py:66-68
                return ref_rank("ranking", users, preds, k, N, consumed, True, False)
            kind = "reference"
        except Exception:  # noqa: BLE001
            kind = "port"
    if kind == "port":
        from oracle import ops_np

        def run(U, users):          # recommend.py:66-68 + the reference's partition-based selection
            return ops_np.rank_recommendations_partition(users, U @ I.T, k, N, consumed, True)
    users = list(range(chunk))
    t_total, n_users = 0.0, 0
    while t_total < seconds_budget and n_users < 4096:
        U = rng.standard_normal((chunk, D)).astype(np.float32)
        t0 = time.perf_counter()
        run(U, users)
        t_total += time.perf_counter() - t0
        n_users += chunk
    return {"value": round(n_users * N / t_total, 1), "unit": "items/s", "cores": torch.get_num_threads(), "kind": kind,
            "sample": f"{n_users} users x {N} items x {D} dims, k={k}, 50 consumed ids per user, float32 numpy GEMM + "
                      f"per-user ranking ({'the reference functions' if kind == 'reference' else 'numpy restatement of the reference functions'})"}


def bench_recommend(args, dev, rank=0, world=1):
    """Second half of the metric: recommend_user items-scored/sec (SURVEY §8d cfg 4): every GPU
    holds a 12.5M x 128 item shard (100M items / 8), scores 1,024 users against it with the
    fused score+top-k kernel (k=100, consumed lists of 50 global ids per user); with N>1 the
    [B,k] candidates are all-gathered and merged (item-sharded scoring, weak scaling)."""
    from librecommender_amd import ops

    # one GPU: the FULL cfg 4 catalogue (100 M x 128 f32 = 51.2 GB, fits the 288 GB of one MI355X); N > 1: the catalogue
    # item-sharded, 100 M / 8 = 12.5 M items per GPU (weak scaling in the catalogue, as cfg 4 shards it)
    n_full = 100_000_000 if world == 1 else 12_500_000
    B, N, D, k = (1024, n_full, 128, 100) if not args.small else (256, 200_000, 128, 10)
    g = torch.Generator(device=dev).manual_seed(42)          # same users / consumed on every rank
    U = torch.randn((B, D), device=dev, generator=g)
    cons = torch.sort(torch.randint(0, N * world, (B, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values
    gi = torch.Generator(device=dev).manual_seed(43 + rank)
    I = torch.empty((N, D), device=dev)
    for lo in range(0, N, 10_000_000):                        # filled in slices: one generator call stays below 2^31 elements
        I[lo:lo + 10_000_000].normal_(generator=gi)
    ptr = (torch.arange(B + 1, device=dev, dtype=torch.int64) * 50)
    flag = torch.ones(B, dtype=torch.uint8, device=dev)
    cidx = cons.reshape(-1).contiguous()
    arith = ops.TOPK_ARITH                                     # the filtered form unless LIBRECO_TOPK_ARITH says otherwise
    failed = torch.zeros(B, dtype=torch.uint8, device=dev)    # (filtered form: users ranked by the exact pass)
    if world == 1:
        lib = ops._lib.load()
        ws = torch.empty(max(lib.lr_score_topk_ws_bytes(B, N, D, k), lib.lr_score_topk_filter_ws_bytes(B, N, D, k)), dtype=torch.uint8, device=dev)
        run_with = lambda a: ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith=a,  # noqa: E731
                                            **({"failed_out": failed} if a.startswith("filter") else {}))
    else:
        from librecommender_amd.parallel import HipKernels, sharded_score_topk
        kern = HipKernels()
        run_with = lambda a: sharded_score_topk(kern, U, I, k, rank * N, ptr, cidx, flag)  # noqa: E731  (ops.TOPK_ARITH)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    clocks = {}

    def timed(a):
        """(seconds per pass by the wall clock between barriers, max over ranks; mean launch ms by HIP events)"""
        name = TOPK_ENTRY[a]
        run_with(a)
        reps = 3 if N <= 20_000_000 else 2
        ops.TIMER.enable(name)
        clk = ClockProbe(dev)
        clk.mark(0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_with(a)
        barrier()
        dt_ = (time.perf_counter() - t0) / reps
        clk.mark(1)
        clocks[a] = clk.mhz()
        ops.TIMER.disable()
        if world > 1:
            tt = torch.tensor([dt_], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, ops.TIMER.summary()[name][1], name

    dt, mean_ms, kname = timed(arith)
    # the timed result is checked before it is reported (size-independent properties; the same shape is compared with an
    # fp64 GEMM in tests/test_fullsize_parity_gpu.py::test_score_topk_100m_vs_fp64, both arithmetics): every returned score is
    # the fp32 dot product of its (user, item) pair, lists are sorted, no consumed id is returned, ids are in range
    s_out, i_out = run_with(arith)
    lo = rank * N
    mine = (i_out >= lo) & (i_out < lo + N)                     # (N > 1: the pairs whose item row this rank holds)
    rec = (U[:, None, :] * I[(i_out - lo).clamp(0, N - 1)]).sum(-1)
    err = float(((rec - s_out).abs() * mine).max())
    tol = 1e-4 + 1e-5 * float(s_out.abs().max())
    hit = bool((i_out[:, :, None] == cons.long()[:, None, :]).any())
    ok = (err <= tol and bool((s_out[:, :-1] >= s_out[:, 1:]).all()) and not hit
          and int(i_out.min()) >= 0 and int(i_out.max()) < N * world and (world > 1 or bool(mine.all())))
    if not ok:
        raise RuntimeError(f"recommend leg: the timed result failed its self-check (max |score - dot| {err:.3e} > {tol:.3e}, "
                           f"consumed id returned: {hit})")
    flops = 2.0 * B * N * D
    if arith.startswith("filter"):  # ONE bf16 MFMA product per f32 product, then f32 rescoring of k' candidates per user: the dense bf16 peak
        tf = flops / (mean_ms * 1e-3) / 1e12
        roof = {"kernel": "lr_score_topk_filter_f32 (one-term bf16 score + fused top-k' + merge, f32 rescoring and certification, masked exact pass)",
                "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
                "flops_note": "2 B N D bf16 flop per launch; the pass is bound by instruction issue (staging, conversion, threshold tests), "
                              "not by the matrix pipe: see DESIGN.md section 3",
                "f32_equivalent_TFLOPs": round(tf, 2), "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 3),
                "hbm_frac_of_one_catalogue_read": round(N * D * 4 / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "users_ranked_by_the_exact_pass": int(failed.sum())}
    elif arith == "split_bf16":     # six bf16 MFMA products per f32 product: priced against the dense bf16 peak by what the pipe executes
        tf = 6 * flops / (mean_ms * 1e-3) / 1e12
        roof = {"kernel": "lr_score_topk_sb_f32 (score + fused top-k + merge)", "bound": "mfma", "achieved": round(tf, 1),
                "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
                "flops_note": "6 x 2 B N D bf16 flop per launch (six-term split-bf16 products, f32 accumulation); "
                              "f32-equivalent rate in f32_equivalent_TFLOPs",
                "f32_equivalent_TFLOPs": round(flops / (mean_ms * 1e-3) / 1e12, 2)}
    else:
        tf = flops / (mean_ms * 1e-3) / 1e12
        roof = {"kernel": "lr_score_topk_f32 (score + fused top-k + merge)", "bound": "mfma", "achieved": round(tf, 2),
                "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4)}
    full = not (args.small or world > 1)
    roof.update({"traffic": (pmc_traffic(kname, "recommend_100m") or pmc_traffic("lr_score_topk_f32", "recommend_100m")) if full else None,
                 "traffic_source": "rocprofv3 PMC pass committed under profiles/ (not this run)",
                 "algorithmic_item_bytes": int(N) * D * 4, "flops_per_launch": flops, "mean_launch_ms": round(mean_ms, 3)})
    if full:
        roof = with_profiles(roof, kname, "recommend_100m")
    out = {"metric": "recommend_user items-scored/sec", "value": round(B * N * world / dt, 1), "unit": "items/s",
           "config": {"workload": f"{B} users x {N * world} items ({N} per GPU) x {D} dims, k={k}, "
                                  f"50 consumed/user, f32" + (", item-sharded + all-gather/merge of candidates" if world > 1 else ""),
                      "arithmetic": TOPK_ARITH_NOTE[arith],
                      "shader_clock_mhz": clocks.get(arith)},
           "ms_per_pass": round(dt * 1e3, 3),
           "verified": {"max_abs_score_minus_fp32_dot": err, "tolerance": tol, "sorted": True, "consumed_filtered": True,
                        "pairs_checked": int(mine.sum()), "what": "every returned (user, item, score) of the timed launch"},
           "roofline": roof}
    if world == 1:      # the exact arithmetics, same inputs, same line
        for other in ("split_bf16", "f32_chain"):
            if other == arith or (arith == "f32_chain"):
                continue
            dt2, ms2, _ = timed(other)
            s2, i2 = run_with(other)
            o = {"ms_per_pass": round(dt2 * 1e3, 3), "value": round(B * N / dt2, 1), "unit": "items/s",
                 "shader_clock_mhz": clocks.get(other),
                 "ids_equal_to_default": round(float((i2 == i_out).float().mean()), 6),
                 "id_sets_equal_to_default": round(float((torch.sort(i2, 1).values == torch.sort(i_out, 1).values).float().mean()), 6),
                 "max_abs_score_diff": float((s2 - s_out).abs().max())}
            if other == "f32_chain":
                o["frac_mfma_f32_peak"] = round(flops / (ms2 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
            else:
                o["frac_mfma_bf16_peak"] = round(6 * flops / (ms2 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)
            out[other] = o
            out[other + "_ms_per_pass"] = o["ms_per_pass"]
        # serving-sized batches against the same catalogue: the pass is then bound by reading the catalogue once
        small = {}
        for Bs in (1, 32):
            args_s = (U[:Bs].contiguous(), I, k, ptr[:Bs + 1].contiguous(), cidx, flag[:Bs].contiguous())
            ops.score_topk(*args_s, ws=ws)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                s_s, i_s = ops.score_topk(*args_s, ws=ws)
            torch.cuda.synchronize()
            ms_s = (time.perf_counter() - t0) / 3 * 1e3
            small[str(Bs)] = {"ms_per_pass": round(ms_s, 3), "catalogue_GBps": round(N * D * 4 / (ms_s * 1e-3) / 1e9, 1),
                              "frac_hbm_peak": round(N * D * 4 / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "ids_equal_to_the_batch_of_1024": bool(torch.equal(i_s, i_out[:Bs]))}
        out["small_batches"] = small
    return out


def _emit(result, rank, stdout_fd=None):
    """Rank 0 prints the ONE JSON line on the real stdout (restored first if the run had pointed fd 1 at stderr)."""
    sys.stdout.flush()
    if stdout_fd is not None:
        import ctypes

        ctypes.CDLL(None).fflush(None)      # the banner sits in C stdio's buffer when fd 1 is a file or a pipe
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-batches", type=int, default=8)
    ap.add_argument("--mlp-dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--workload", choices=["deepfm", "din", "twotower", "lightgcn", "deepfm_recommend"], default="deepfm",
                    help="deepfm = BASELINE cfg 2 (the configuration the metric is quoted on); din / twotower / lightgcn = "
                         "cfg 3 / 4 / 5 at full size on one GPU; deepfm_recommend = recommend_user of a DeepFM over cfg 2's "
                         "full catalogue (SURVEY 8 f2)")
    ap.add_argument("--steady-seconds", type=float, default=1.0,
                    help="additionally report ms/step over at least this many seconds of steady-state steps (0 = off)")
    ap.add_argument("--dense-adam-line", action="store_true", help="(default at N=1; kept for older command lines)")
    ap.add_argument("--no-dense-adam-line", action="store_true",
                    help="skip the second GPU line that times the step with TF1's dense Adam over every table row (the CPU "
                         "port's optimiser semantics)")
    ap.add_argument("--small", action="store_true", help="tiny shapes (functional check only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph replay per step")
    ap.add_argument("--unfused", action="store_true",
                    help="materialised deep_embed + library GEMMs for the first layer (round-1 path)")
    ap.add_argument("--no-recommend", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the cfg 3 / 4 / 5 lines (`workloads` object) that the default single-GPU run appends")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU net even at world size 1 (measures the exchange glue)")
    ap.add_argument("--parallel", choices=["row", "field"], default="row",
                    help="multi-GPU scheme: row-sharded tables (north_star) or field-partitioned + tensor-parallel first layer")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = functional check of the N>1 path with ranks sharing one GPU "
                         "(collectives staged through host; not a measurement)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by torch.distributed.run: re-launch this command line as --gpus ranks on this node
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with nproc-per-node {args.gpus}")
    if args.workload != "deepfm" and not (world > 1 or args.force_sharded):
        import bench_workloads

        torch.cuda.set_device(0)
        print(json.dumps(bench_workloads.run(args, torch.device("cuda", 0))))
        return
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    stdout_fd = None
    if world > 1 or args.force_sharded:
        # RCCL writes its banner ("Librccl path : ...") to file descriptor 1 when the first communicator is created: point
        # fd 1 at stderr for the length of the run, so that stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from datetime import timedelta
        limit = timedelta(minutes=5)            # a wedged collective should end the run, not sit out the default 10-30 min
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            torch.distributed.init_process_group("gloo", timeout=limit)

    if args.workload in ("twotower", "lightgcn", "din"):
        # cfg 4's train half (table row-sharded over the ranks) / cfg 5 (node table + Laplacian row-partitioned): strong scaling;
        # cfg 3 with its table row-sharded: weak scaling
        import bench_workloads

        fn = {"twotower": bench_workloads.bench_twotower_sharded, "lightgcn": bench_workloads.bench_lightgcn_sharded,
              "din": bench_workloads.bench_din_sharded}[args.workload]
        res = fn(args, rank, world, dev)
        torch.distributed.destroy_process_group()
        _emit(res, rank, stdout_fd)
        return
    result, cfg, host = bench_train(args, rank, world, dev)
    _release(dev)
    if not args.no_recommend:
        # (world > 1: the leg holds collectives — a rank that swallowed its own failure would leave the others blocked in
        # them; there a failure ends the run, round-4 advisor finding)
        rec = _guard(lambda: bench_recommend(args, dev, rank, world)) if world == 1 else bench_recommend(args, dev, rank, world)
        _release(dev)
        if rank == 0:
            result["recommend"] = rec
    if rank == 0 and world == 1 and not args.force_sharded and not args.unfused and "split-bf16" in str(result.get("dtype")):
        f32c = _guard(lambda: bench_f32_chain(args, cfg, dev))
        result["f32_chain"] = f32c
        result["f32_chain_ms_per_step"] = f32c.get("ms_per_step") if isinstance(f32c, dict) else None
        _release(dev)
    if rank == 0 and world == 1 and not args.no_dense_adam_line and not args.small:
        result["dense_adam"] = _guard(lambda: bench_dense_adam(args, cfg, host, dev))
        _release(dev)
    if rank == 0 and world == 1:
        result["host_cores"] = os.cpu_count()
        result["reference_checkout"] = _reference_note()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = bench_cpu_baseline(cfg, host)
            if isinstance(result.get("recommend"), dict) and "error" not in result["recommend"]:
                result["recommend"]["cpu_baseline"] = bench_recommend_cpu_baseline()
        if not args.no_workloads and not args.force_sharded:
            # the other BASELINE.json configurations at full size on this one GPU, each with its own `roofline` +
            # `cpu_baseline` (the same functions `--workload {din,twotower,lightgcn}` prints as stand-alone lines): the
            # driver runs exactly this default command, so every number DESIGN.md quotes is in the line it records
            import bench_workloads

            result["workloads"] = {}
            for name in ("din", "twotower", "lightgcn", "deepfm_recommend"):
                wargs = argparse.Namespace(**{**vars(args), "workload": name, "no_recommend": True})
                result["workloads"][name] = _guard(lambda: bench_workloads.run(wargs, dev))
                _release(dev)
    if rank == 0:
        result["config"]["other_legs"] = _legs_summary(result)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    _emit(result, rank, stdout_fd)


def _legs_summary(result):
    """The numbers of the secondary legs in a few scalars INSIDE `config` (the driver's record keeps `config`, `roofline` and
    `cpu_baseline` whole, of every other key only the name): recommend leg, f32-chain / dense-Adam variants, cfg 3 / 4 / 5."""
    out = {}

    def g(d, *path):
        for p_ in path:
            if not isinstance(d, dict) or p_ not in d:
                return None
            d = d[p_]
        return d

    rec = result.get("recommend")
    if isinstance(rec, dict) and "error" not in rec:
        out["recommend_items_per_s"], out["recommend_ms_per_pass"] = rec.get("value"), rec.get("ms_per_pass")
        out["recommend_roofline_frac"], out["recommend_roofline_peak"] = g(rec, "roofline", "frac"), g(rec, "roofline", "peak")
        out["recommend_shader_clock_mhz"] = g(rec, "config", "shader_clock_mhz")
        out["recommend_f32_chain_shader_clock_mhz"] = g(rec, "f32_chain", "shader_clock_mhz")
        out["recommend_f32_chain_ms_per_pass"] = rec.get("f32_chain_ms_per_pass")
        out["recommend_split_bf16_ms_per_pass"] = rec.get("split_bf16_ms_per_pass")
        out["recommend_split_bf16_shader_clock_mhz"] = g(rec, "split_bf16", "shader_clock_mhz")
        out["recommend_users_ranked_by_the_exact_pass"] = g(rec, "roofline", "users_ranked_by_the_exact_pass")
        out["recommend_id_sets_equal_f32_chain"] = g(rec, "f32_chain", "id_sets_equal_to_default")
        out["recommend_1_user_ms_per_pass"] = g(rec, "small_batches", "1", "ms_per_pass")
        out["recommend_32_users_ms_per_pass"] = g(rec, "small_batches", "32", "ms_per_pass")
        out["recommend_cpu_items_per_s"] = g(rec, "cpu_baseline", "value")
    out["f32_chain_ms_per_step"] = result.get("f32_chain_ms_per_step")
    out["dense_adam_ms_per_step"] = g(result, "dense_adam", "ms_per_step")
    for name, w in (result.get("workloads") or {}).items():
        if isinstance(w, dict) and "error" not in w:
            out[name] = {"ms_per_step": w.get("ms_per_step"), "samples_per_s": w.get("value"),
                         "shader_clock_mhz": g(w, "config", "shader_clock_mhz"),
                         "roofline_kernel": g(w, "roofline", "kernel"), "roofline_frac": g(w, "roofline", "frac"),
                         "frac_by_traffic": g(w, "roofline", "frac_by_traffic"), "cpu_samples_per_s": g(w, "cpu_baseline", "value")}
            if isinstance(w.get("recommend"), dict):
                out[name]["recommend_items_per_s"] = g(w, "recommend", "value")
            if name == "deepfm_recommend":
                out[name] = {"items_per_s": w.get("value"), "ms_per_pass": w.get("ms_per_pass"),
                             "roofline_frac": g(w, "roofline", "frac"), "cpu_items_per_s": g(w, "cpu_baseline", "value")}
        elif isinstance(w, dict):
            out[name] = {"error": w.get("error")}
    return out


def _release(dev):
    import gc

    gc.collect()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()


def _guard(fn):
    """A failing secondary leg is reported inside the line (`{"error": ...}`); it does not take the headline number with it."""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(e).__name__}: {e}"[:400]}


def _reference_note():
    """Where the CPU baselines come from on this box (`cpu_baseline.kind`): the reference checkout exists only in the build
    container; its own functions were timed there (8 cores) and the numbers are committed next to this file."""
    from oracle import ref_loader

    note = {"present": bool(ref_loader.available()),
            "meaning": "kind 'reference' = the reference's own functions ran on this box; kind 'port' = the oracle restatement "
                       "(the checkout /root/reference is absent here; TensorFlow is absent everywhere, so the FM / DeepFM / DIN / "
                       "TwoTower graphs are always the restatement)"}
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_cpu_reference_baselines.json")) as fh:
            note["reference_run_build_container"] = json.load(fh)
    except OSError:
        pass
    return note


if __name__ == "__main__":
    main()
