"""The other BASELINE.json configurations as `bench.py --workload {din,twotower,lightgcn}` lines (one MI355X each,
full size, synthetic inputs resident in HBM before the timed region; same JSON contract as the default DeepFM line:
`roofline` of the dominant hand-written kernel from HIP events, a whole-step roofline, `cpu_baseline` on a bounded
sample).  SURVEY 8(d):

  din       cfg 3: DIN, 1 M users, 10 M items, K = 128, L = 50 (lengths U{1..50}), B = 8,192, MLP (128, 64, 32)
  twotower  cfg 4 on ONE GPU: 100 M items x 128 (51 GB + Adam moments) + 1 M users, towers (128,), in-batch softmax
            B = 65,536, and the recommend leg against the full 100 M-item table
  lightgcn  cfg 5 on ONE GPU: 10 M x 10 M nodes, 200 M distinct interactions (400 M nnz), K = 64, 3 layers, BPR B = 65,536; the
            Laplacian is built on the device (`lr_csr_laplacian_build`)
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3


def zipf_ids_device(n, vocab, gen, dev, a=1.05):
    """`(numpy.random.Generator.zipf(a) - 1) % vocab` drawn on the device: the same law (P(X = k) ~ k^-a exactly, not the
    floor-of-Pareto approximation of the earlier rounds, whose head is lighter: P(1) = 0.034 instead of 0.049 at a = 1.05),
    by the rejection sampler numpy itself uses (Devroye X.6.1: X = floor(U^(-1/(a-1))), accept with probability
    T / b * (b - 1) / (X (T - 1)), T = (1 + 1/X)^(a-1), b = 2^(a-1); candidates beyond int64 are rejected as numpy does)."""
    am1 = a - 1.0
    b = 2.0 ** am1
    out = torch.empty(n, dtype=torch.int64, device=dev)
    have = 0
    while have < n:
        m = int((n - have) * 1.7) + 256
        u = 1.0 - torch.rand(m, device=dev, generator=gen, dtype=torch.float64)
        v = torch.rand(m, device=dev, generator=gen, dtype=torch.float64)
        x = torch.floor(u.pow_(-1.0 / am1))
        t = (1.0 + 1.0 / x).pow_(am1)
        ok = (x >= 1.0) & (x <= 9.2233720368547e18) & (v * x * (t - 1.0) / (b - 1.0) <= t / b)
        acc = x[ok].to(torch.int64)
        k = min(int(acc.numel()), n - have)
        out[have:have + k] = acc[:k]
        have += k
    return (out - 1).remainder_(vocab).to(torch.int32)


def distinct_interactions(E, n_users, n_items, gen, dev):
    """E DISTINCT (user, item) pairs with Zipf(1.05) endpoints, as int32 device arrays.  The reference builds its
    Laplacian from `user_consumed` — de-duplicated lists, binary weights (`lightgcn_module.py:36-61`) — so a repeated draw of
    a pair adds no nonzero: draws are topped up until E distinct pairs exist (nnz = 2 E, SURVEY 8(d) cfg 5), then a
    random subset of exactly E is kept."""
    key = torch.empty(0, dtype=torch.int64, device=dev)
    while key.numel() < E:
        n_draw = int((E - key.numel()) * 1.5) + 1024
        k = zipf_ids_device(n_draw, n_users, gen, dev).to(torch.int64) * n_items + zipf_ids_device(n_draw, n_items, gen, dev)
        key = torch.unique(torch.cat([key, k]))
        del k
    if key.numel() > E:
        keep = torch.randperm(key.numel(), generator=gen, device=dev)[:E]
        key = key[keep]
        del keep
    eu = torch.div(key, n_items, rounding_mode="floor").to(torch.int32)
    ei = key.remainder(n_items).to(torch.int32)
    return eu, ei


class Pool:
    """The synthetic interaction STREAM of a workload: distinct batches drawn on the device from one seeded generator and
    kept resident in HBM.  `ensure(n)` tops the pool up OUTSIDE the timed regions; a timed loop then takes the next unseen
    batch every step, so no batch is trained on twice (round 3 cycled through 8 resident batches and the nets memorised
    them: final losses of 0.09 / 3e-5 on Bernoulli(0.5) labels)."""

    def __init__(self, make_one):
        self.make_one, self.items, self.cursor = make_one, [], 0

    def ensure(self, n_more):
        while len(self.items) - self.cursor < n_more:
            self.items.append(self.make_one())

    def next(self):
        if self.cursor >= len(self.items):          # not reached by the timed loops (they call `ensure` first)
            self.items.append(self.make_one())
        b = self.items[self.cursor]
        self.items[self.cursor] = None              # a consumed batch is not needed again: free it
        self.cursor += 1
        return b

    def peek(self, ahead=0):
        while self.cursor + ahead >= len(self.items):
            self.items.append(self.make_one())
        return self.items[self.cursor + ahead]


LAST_CLOCK_MHZ = None       # shader clock of the last `_timed` region (bench.ClockProbe); `_base` puts it into `config`


def _timed(step, steps, warmup, min_seconds=0.0, pool=None, per_step=1):
    global LAST_CLOCK_MHZ
    if pool is not None:
        pool.ensure((warmup + steps) * per_step + 1)
    for _ in range(warmup):
        step()
    clk = None
    try:
        from bench import ClockProbe

        clk = ClockProbe(torch.device("cuda", torch.cuda.current_device()))
        clk.mark(0)
    except Exception:  # noqa: BLE001  (the probe is an aid: never the reason a line is lost)
        clk = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    LAST_CLOCK_MHZ = None
    if clk is not None:
        try:
            clk.mark(1)
            LAST_CLOCK_MHZ = clk.mhz()
        except Exception:  # noqa: BLE001
            LAST_CLOCK_MHZ = None
    extra = None
    if min_seconds > 0:           # steady-state figure over >= min_seconds of steps on fresh batches (not `value`)
        n = max(steps, int(min_seconds / max(dt / steps, 1e-6)) + 1)
        if pool is not None:
            pool.ensure(n * per_step + 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        extra = {"steps": n, "ms_per_step": round((time.perf_counter() - t1) / n * 1e3, 4)}
    return dt, out, extra


def _kernel_table(ops, names, fn, reps, pool=None):
    if pool is not None:
        pool.ensure(reps + 1)
    ops.TIMER.enable(*names)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ops.TIMER.disable()
    return ops.TIMER.summary()


def _traffic(d, name, workload, mean_ms):
    """`traffic` = fabric bytes per launch of the committed PMC pass of this workload at FULL size (profiles/r03_pmc_traffic.json)."""
    if workload is None:
        return d
    from bench import pmc_traffic, with_profiles

    tr = pmc_traffic(name, workload)
    if tr:
        d["traffic"] = tr
        d["traffic_source"] = "rocprofv3 PMC pass committed under profiles/ (not this run)"
        d["achieved_by_traffic"] = round(tr / (mean_ms * 1e-3) / 1e9, 1)          # GB/s at the fabric
        if d["bound"] == "hbm":
            d["frac_by_traffic"] = round(tr / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return with_profiles(d, name.split(" ")[0], workload)


def _roof_hbm(name, nbytes, mean_ms, extra=None, workload=None):
    a = nbytes / (mean_ms * 1e-3) / 1e9
    d = {"kernel": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(nbytes),
         "mean_launch_ms": round(mean_ms, 4)}
    if extra:
        d.update(extra)
    return _traffic(d, name, workload, mean_ms)


def _roof_mfma(name, flops, mean_ms, extra=None, workload=None, traffic_key=None):
    a = flops / (mean_ms * 1e-3) / 1e12
    d = {"kernel": name, "bound": "mfma", "achieved": round(a, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
         "frac": round(a / MFMA_F32_PEAK_TF, 4), "traffic": None, "flops_per_launch": float(flops),
         "mean_launch_ms": round(mean_ms, 4)}
    if extra:
        d.update(extra)
    return _traffic(d, traffic_key or name, workload, mean_ms)


def _base(metric_value, B, steps, warmup, ms, dtype, workload, config_extra):
    return {"metric": "train samples/sec", "value": round(metric_value, 1), "unit": "samples/s", "n_gpus": 1,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": B, "global_batch": B, "shader_clock_mhz": LAST_CLOCK_MHZ, **config_extra}}


# ======================================================================================================
# cfg 3: DIN
# ======================================================================================================
DIN_CFG = dict(n_users=1_000_000, n_items=10_000_000, embed_size=128, max_seq_len=50, batch=8192, hidden_units=(128, 64, 32))


def din_batch_maker(cfg, dev, seed=42):
    g = torch.Generator(device=dev).manual_seed(seed)
    nu, ni, L, B = cfg["n_users"], cfg["n_items"], cfg["max_seq_len"], cfg["batch"]
    ar = torch.arange(L, device=dev)[None, :]

    def one():
        users = zipf_ids_device(B, nu, g, dev)
        items = zipf_ids_device(B, ni, g, dev)
        lens = torch.randint(1, L + 1, (B,), device=dev, generator=g, dtype=torch.int32)
        seqs = zipf_ids_device(B * L, ni, g, dev).view(B, L)
        seqs = torch.where(ar < lens[:, None], seqs, torch.full_like(seqs, ni))         # pad id = n_items (sequence.py:56-58)
        labels = torch.randint(0, 2, (B,), device=dev, generator=g).float()
        return (users, items, seqs.contiguous(), lens, labels)
    return one


def din_batches(cfg, n_batches, dev, seed=42):
    one = din_batch_maker(cfg, dev, seed)
    return [one() for _ in range(n_batches)]


def bench_din(args, dev):
    from librecommender_amd import ops
    from librecommender_amd.nets import FeatDINNet, FeatSpec

    cfg = dict(DIN_CFG)
    if args.small:
        cfg.update(n_users=20_000, n_items=100_000, batch=2048)
    K, L, B = cfg["embed_size"], cfg["max_seq_len"], cfg["batch"]
    net = FeatDINNet(FeatSpec(cfg["n_users"], cfg["n_items"]), K, cfg["hidden_units"], use_bn=True, max_seq_len=L, lr=1e-3,
                     device=dev, graph_step=not args.no_graph)
    assert net._fstep is not None, "the fused DIN step is not active"
    pool = Pool(din_batch_maker(cfg, dev))
    batches = [pool.peek(k) for k in range(4)]          # the first batches, also handed to the CPU baseline
    last = [None]

    def step():
        last[0] = pool.next()
        u, i, s, ln, lab = last[0]
        return net.train_step(u, i, lab, seqs=s, seq_lens=ln)

    dt, loss, steady = _timed(step, args.steps, max(args.warmup, 3), min_seconds=args.steady_seconds, pool=pool)
    ms = dt / args.steps * 1e3
    # per-kernel HIP-event times: eager launches of the same kernels after the timed region
    net.graph_step = False
    names = ("lr_embed_gather_f32", "lr_din_attn_pool_fwd_f32", "lr_din_attn_pool_bwd_parts_f32", "lr_table_colstats_f32",
             "lr_deepfm_l1_fwd_f32", "lr_deepfm_l1_wgrad_f32", "lr_deepfm_l1_dgrad_f32", "lr_bn_remainder_f32",
             "lr_segments_build", "lr_embed_scatter_adam_f32", "lr_adam_dense_f32", "lr_mlp_colstats_f32",
             "lr_mlp_bn_finalize_f32", "lr_mlp_layer_fwd_f32", "lr_mlp_head_f32", "lr_mlp_layer_bwd_f32", "lr_mlp_first_bwd_f32",
             "lr_reduce_partials_f32", "lr_reduce_partials_multi_f32", "lr_deepfm_l1_fold_stats_f32", "lr_deepfm_l1_fold_stats_bias_f32",
             "lr_deepfm_l1_pack_scaled_f32", "lr_deepfm_l1_fold_bias_f32", "lr_deepfm_l1_fold_bwd_f32")
    kern = _kernel_table(ops, names, step, min(args.steps, 10), pool=pool)
    net.graph_step = not args.no_graph
    # algorithmic bytes (SURVEY 8d cfg 3): rows of K * 4 bytes, counted on the last batch the eager pass ran
    u, i, s, ln, lab = last[0]
    row = K * 4
    n_valid = int(ln.sum().item())
    bset = net._fstep.sets[(B, L)]
    n_distinct = int(bset.seg.n_seg.item())
    n_pos = 3 * B + n_valid                                   # user, item (MLP), item (query), window rows
    by = {"lr_din_attn_pool_fwd_f32": (n_valid + 2 * B) * row,                       # keys + query read, output written
          "lr_din_attn_pool_bwd_parts_f32": (n_valid + 2 * B) * row + (n_valid + B) * row,   # keys + query + gout read, gkey + gq written
          "lr_embed_scatter_adam_f32": n_pos * row + 6 * n_distinct * row}            # gradient rows + RMW of w, m, v
    kinfo = {}
    for name, (n, mean_ms) in kern.items():
        kinfo[name] = {"launches": n, "mean_ms": round(mean_ms, 4)}
        if name in by:
            kinfo[name]["algorithmic_GBps"] = round(by[name] / (mean_ms * 1e-3) / 1e9, 1)
            kinfo[name]["frac_hbm_peak"] = round(by[name] / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    dom = max((n for n in kern if n in by), key=lambda n: kern[n][1])
    step_bytes = (n_valid + 2 * B) * row * 3 + 4 * n_distinct * row          # fwd read + bwd RMW (8d: 42.3 KB/sample) + Adam m, v
    res = _base(B * args.steps / dt, B, args.steps, args.warmup, ms, "f32",
                "DIN train step (cfg 3): 1M users x 10M items, embed_size=128, L=50 (lengths U{1..50}), MLP (128,64,32), "
                "Zipf(1.05) ids" if not args.small else "DIN small (smoke)",
                {"embed_size": K, "max_seq_len": L, "table_rows": net.tables.V, "mean_seq_len": round(n_valid / B, 2),
                 "distinct_rows_per_step": n_distinct, "final_loss": round(float(loss), 5),
                 "stream": f"{pool.cursor} distinct batches drawn on the device (exact Zipf(1.05) ids), none trained on twice",
                 "optimizer": "row-wise Adam on the touched rows + dense Adam (attention MLP, MLP, BatchNorm)",
                 "launch": "one hipGraph replay per step (dedicated stream)" if not args.no_graph else "eager launches"})
    res["roofline"] = _roof_hbm(dom, by[dom], kern[dom][1], workload=None if args.small else "din")
    res["roofline_step"] = {"bound": "hbm", "algorithmic_bytes_per_step": int(step_bytes),
                            "achieved": round(step_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "SURVEY 8(d) cfg 3: (L+2) rows read forward + read-modify-write backward + Adam moments"}
    n_eager = min(args.steps, 10)
    res["kernels"], res["sum_kernel_ms"] = kinfo, round(sum(c * m for c, m in kern.values()) / n_eager, 4)
    res["kernel_timing"] = "HIP events around every C-ABI launch in eager steps run after the timed region"
    if steady:
        res["steady_state"] = steady
    return res, cfg, batches, net


def cpu_baseline_din(cfg, batches, net, budget=40.0):
    """`DINOracle` (fp32, TF1 dense Adam over every table row) from the HIP model's weights on the host cores."""
    from oracle.models_torch import DINOracle, export_net_weights

    B = cfg["batch"]
    W = export_net_weights(net)
    o = DINOracle(W, cfg["hidden_units"], True, cfg["max_seq_len"], lr=1e-3, dtype=torch.float32)
    del W
    t_tot, n = 0.0, 0
    for k in range(4):
        u, i, s, ln, lab = [x.cpu() for x in batches[k % len(batches)]]
        t0 = time.perf_counter()
        o.train_step(u.long(), i.long(), None, None, s.long(), ln.long(), lab)
        dt = time.perf_counter() - t0
        if k > 0:
            t_tot += dt
            n += 1
        if t_tot > budget or (k == 0 and dt > budget):
            break
    n = max(n, 1)
    return {"value": round(B * n / max(t_tot, 1e-9), 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full-size training steps (B={B}, same tables / batches) of the PyTorch-CPU oracle restatement of "
                      f"algorithms/din.py incl. TF1 dense Adam over all {net.tables.V} rows; first step untimed"}


# ======================================================================================================
# cfg 4: TwoTower on one GPU
# ======================================================================================================
TT_CFG = dict(n_users=1_000_000, n_items=100_000_000, embed_size=128, hidden_units=(128,), batch=65_536)


def bench_twotower(args, dev):
    from librecommender_amd import ops
    from librecommender_amd.nets import TwoTowerNet

    cfg = dict(TT_CFG)
    if args.small:
        cfg.update(n_users=50_000, n_items=500_000, batch=4096)
    nu, ni, K, B = cfg["n_users"], cfg["n_items"], cfg["embed_size"], cfg["batch"]
    net = TwoTowerNet(nu, ni, 0, 0, 0, [], [], 0, embed_size=K, hidden_units=cfg["hidden_units"], use_bn=False, lr=1e-3,
                      device=dev)
    g = torch.Generator(device=dev).manual_seed(42)

    def one():
        users = zipf_ids_device(B, nu, g, dev)
        items = zipf_ids_device(B, ni, g, dev)
        corr = torch.rand(B, device=dev, generator=g) * 1e-3 + 1e-6          # sampling probabilities Q(item)
        return (users, items, corr)
    pool = Pool(one)
    batches = [pool.peek(k) for k in range(6)]           # the first batches, also handed to the CPU baseline

    def step():
        u, i, c = pool.next()
        return net.train_step("softmax", u, i, corrections=c)

    dt, loss, steady = _timed(step, args.steps, max(args.warmup, 2), min_seconds=args.steady_seconds, pool=pool)
    ms = dt / args.steps * 1e3
    names = ("lr_softmax_ce_fwd_f32", "lr_softmax_ce_bwd_cols_f32", "lr_embed_gather_f32", "lr_segments_build",
             "lr_embed_scatter_adam_f32", "lr_adam_dense_f32")
    kern = _kernel_table(ops, names, step, min(args.steps, 5), pool=pool)
    D = net.out_dim
    fl = {"lr_softmax_ce_fwd_f32": 4.0 * B * B * D, "lr_softmax_ce_bwd_cols_f32": 4.0 * B * B * D}
    kinfo = {}
    for name, (n, mean_ms) in kern.items():
        kinfo[name] = {"launches": n, "mean_ms": round(mean_ms, 4)}
        if name in fl:
            kinfo[name]["TFLOPs"] = round(fl[name] / (mean_ms * 1e-3) / 1e12, 2)
            kinfo[name]["frac_mfma_f32_peak"] = round(fl[name] / (mean_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
    dom = max((n for n in kern if n in fl), key=lambda n: kern[n][1])
    sb = ops.SCE_ARITH == "split_bf16"
    if sb:
        from bench import MFMA_BF16_PEAK_TF
        for name in fl:
            if name in kinfo:
                kinfo[name]["f32_equivalent_TFLOPs"] = kinfo[name].pop("TFLOPs")
                kinfo[name]["bf16_mfma_TFLOPs"] = round(6 * fl[name] / (kern[name][1] * 1e-3) / 1e12, 1)
                kinfo[name]["frac_mfma_bf16_peak"] = round(6 * fl[name] / (kern[name][1] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)
    res = _base(B * args.steps / dt, B, args.steps, args.warmup, ms,
                "f32 (split-bf16 x6 MFMA products, f32 accumulate)" if sb else "f32",
                f"TwoTower train step (cfg 4 on one GPU): {nu} users + {ni} items x {K} in one table ({net.tables.V} rows, "
                f"{net.tables.bytes() / 1e9:.0f} GB with Adam moments), towers {cfg['hidden_units']}, in-batch softmax with logQ "
                f"correction, B={B}, Zipf(1.05) ids" if not args.small else "TwoTower small (smoke)",
                {"embed_size": K, "table_rows": net.tables.V, "final_loss": round(float(loss), 5),
                 "stream": f"{pool.cursor} distinct batches drawn on the device (exact Zipf(1.05) ids), none trained on twice",
                 "loss": "streaming softmax cross-entropy (no B x B logits): " + (
                     "its four contractions as six-term split-bf16 MFMA products with f32 accumulation (every f32 operand, the "
                     "probabilities included, split exactly into three bf16 values; as close to fp64 as the f32 fma chain: "
                     "tests/test_softmax_ce_gpu.py runs every case under both; LIBRECO_SCE_ARITH=f32_chain selects the chain, timed "
                     "beside as f32_chain_ms_per_step)" if sb else "exact f32 MFMA fma chain"),
                 "optimizer": "row-wise Adam on the touched rows + dense Adam (towers)", "launch": "eager launches"})
    if sb:
        from bench import MFMA_BF16_PEAK_TF
        a6 = 6 * fl[dom] / (kern[dom][1] * 1e-3) / 1e12
        res["roofline"] = _roof_mfma(dom, 6 * fl[dom], kern[dom][1],
                                     {"achieved": round(a6, 1), "peak": MFMA_BF16_PEAK_TF, "frac": round(a6 / MFMA_BF16_PEAK_TF, 4),
                                      "f32_equivalent_flops_per_launch": fl[dom],
                                      "note": "2*B*B*D forward scores + 2*B*B*D W = P Y in one sweep, six bf16 MFMA products per f32 "
                                              "product (split-bf16, f32 accumulate): flops the pipe executes, against the dense bf16 peak"},
                                     workload=None if args.small else "twotower")
    else:
        res["roofline"] = _roof_mfma(dom, fl[dom], kern[dom][1], {"note": "2*B*B*D forward scores + 2*B*B*D W = P Y in one sweep"},
                                     workload=None if args.small else "twotower")
    step_fl = 8.0 * B * B * D + 2 * 2 * 3 * B * K * D          # softmax-CE (4 contractions) + towers fwd/bwd
    if sb:      # the pipe executes six bf16 products per f32 product: priced against the dense bf16 peak (never the f32 one)
        from bench import MFMA_BF16_PEAK_TF
        a6 = 6 * step_fl / (ms * 1e-3) / 1e12
        res["roofline_step"] = {"bound": "mfma", "flops_per_step": 6 * step_fl, "f32_equivalent_flops_per_step": step_fl,
                                "achieved": round(a6, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(a6 / MFMA_BF16_PEAK_TF, 4),
                                "f32_equivalent_TFLOPs": round(step_fl / (ms * 1e-3) / 1e12, 2)}
    else:
        res["roofline_step"] = {"bound": "mfma", "flops_per_step": step_fl, "achieved": round(step_fl / (ms * 1e-3) / 1e12, 2),
                                "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(step_fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)}
    res["kernels"], res["sum_kernel_ms"] = kinfo, round(sum(m for _, m in kern.values()), 4)
    res["kernel_timing"] = "HIP events around every C-ABI launch in eager steps run after the timed region"
    if steady:
        res["steady_state"] = steady
    if sb and not args.small:       # the exact f32 chain on the same net and stream, timed in the same line
        prev = ops.set_sce_arith("f32_chain")
        try:
            dt2, _, _ = _timed(step, max(args.steps // 2, 3), 2, min_seconds=0, pool=pool)
            res["f32_chain_ms_per_step"] = round(dt2 / max(args.steps // 2, 3) * 1e3, 4)
        finally:
            ops.set_sce_arith(prev)
    return res, cfg, batches, net


def bench_twotower_sharded(args, rank, world, dev):
    """cfg 4's train half as north_star splits it: the [1 M users | 100 M items] x 128 table ROW-SHARDED over the ranks
    (row r on rank r % W), ids / rows / row gradients exchanged by RCCL all-to-all (`ShardedFieldTables`), the GLOBAL
    in-batch softmax at B = 65,536 (item-tower outputs all-gathered, each rank scores its B / W users against all B
    items with the streaming softmax-CE kernels, gradient of the gathered block back by reduce-scatter), dense
    parameters all-reduced.  STRONG scaling: the global batch is fixed, each rank takes B / W samples."""
    import torch.distributed as dist

    from librecommender_amd.nets import ShardedTwoTowerNet

    cfg = dict(TT_CFG)
    if args.small:
        cfg.update(n_users=50_000, n_items=500_000, batch=4096)
    nu, ni, K, B = cfg["n_users"], cfg["n_items"], cfg["embed_size"], cfg["batch"]
    Bl = B // world
    V = nu + 1 + ni
    net = ShardedTwoTowerNet(V, 1, 1, embed_size=K, hidden_units=cfg["hidden_units"], use_bn=False, lr=1e-3, device=dev)
    g = torch.Generator(device=dev).manual_seed(42)        # the same global batch on every rank; each takes its slice

    def one():
        users = zipf_ids_device(B, nu, g, dev)[rank * Bl:(rank + 1) * Bl]
        items = zipf_ids_device(B, ni, g, dev)[rank * Bl:(rank + 1) * Bl]
        corr = (torch.rand(B, device=dev, generator=g) * 1e-3 + 1e-6)[rank * Bl:(rank + 1) * Bl]
        u_idx = users.view(-1, 1).contiguous()
        i_idx = (items + (nu + 1)).view(-1, 1).contiguous()
        return (u_idx, i_idx, items.contiguous(), corr.contiguous(), torch.cat([u_idx, i_idx], dim=1).contiguous())
    pool = Pool(one)                # a fresh global batch every step (the same stream on every rank)
    pool.ensure(max(args.warmup, 2) + args.steps + 2)

    def step():
        u, i, it, c, cat = pool.next()
        return net.train_step("softmax", u, i, items=it, corrections=c, idx=cat, next_idx=pool.peek(0)[4])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    res = _base(B * args.steps / dt, Bl, args.steps, args.warmup, ms, "f32",
                f"TwoTower train step (cfg 4, row-sharded): {nu} users + {ni} items x {K}, table sharded {world}-way "
                f"({net.tables.V_local} rows per rank), towers {cfg['hidden_units']}, GLOBAL in-batch softmax B={B}, Zipf(1.05) ids",
                {"embed_size": K, "table_rows": V, "final_loss": round(float(loss), 5),
                 "parallelism": f"dp{world} batch + table row-sharded {world}-way (RCCL all-to-all of de-duplicated ids / rows / row "
                                f"gradients, all-gather of item-tower outputs + reduce-scatter of their gradient, all-reduce of dense grads)",
                 "launch": "eager launches"})
    res["n_gpus"], res["scaling"] = world, "strong"
    res["config"]["global_batch"] = B
    D = net.user_tower.n_out
    from librecommender_amd import ops as _ops
    from bench import MFMA_BF16_PEAK_TF

    sb = _ops.SCE_ARITH == "split_bf16"
    fl = 8.0 * Bl * B * D * (6 if sb else 1)
    peak = MFMA_BF16_PEAK_TF if sb else MFMA_F32_PEAK_TF
    res["dtype"] = "f32 (split-bf16 x6 MFMA products, f32 accumulate)" if sb else "f32"
    res["roofline_step"] = {"bound": "mfma", "flops_per_step_per_gpu": fl, "achieved": round(fl / (ms * 1e-3) / 1e12, 2),
                            "peak": peak, "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4),
                            "note": "this rank's share of the global softmax (4 contractions of [B/W, B, D]"
                                    + (", six bf16 MFMA products per f32 product: flops the pipe executes)" if sb else ")")}
    res["roofline"] = dict(res["roofline_step"], kernel="lr_softmax_ce_fwd_f32 + lr_softmax_ce_bwd_cols_f32 (whole step / their flops)",
                           traffic=None)
    return res


def bench_din_sharded(args, rank, world, dev):
    """cfg 3 with the [users | items] x 128 table ROW-SHARDED over the ranks (`nets/feat_nets.py:ShardedDINNet`; row r on rank
    r % W): the global rows of [user, item, the window's items] of every sample are looked up through the all-to-all exchange
    of de-duplicated ids / rows, attention + MLP run on the fetched rows (replicated dense parameters, global-batch BatchNorm),
    row gradients go back to their owners, dense gradients are all-reduced.  WEAK scaling: 8,192 samples per GPU."""
    import torch.distributed as dist

    from librecommender_amd.nets import ShardedDINNet

    cfg = dict(DIN_CFG)
    if args.small:
        cfg.update(n_users=20_000, n_items=100_000, batch=1024)
    nu, ni, K, L, B = cfg["n_users"], cfg["n_items"], cfg["embed_size"], cfg["max_seq_len"], cfg["batch"]
    V = nu + 1 + ni + 1
    net = ShardedDINNet(V, K, cfg["hidden_units"], use_bn=True, max_seq_len=L, lr=1e-3, device=dev)
    maker = din_batch_maker(cfg, dev, seed=42 + rank)

    def one():
        users, items, seqs, lens, labels = maker()
        idx = torch.cat([users.view(-1, 1), items.view(-1, 1) + (nu + 1), seqs + (nu + 1)], dim=1).to(torch.int32).contiguous()
        return idx, lens, labels
    pool = Pool(one)
    pool.ensure(max(args.warmup, 2) + args.steps + 2)

    def step():
        idx, lens, labels = pool.next()
        return net.train_step(idx, lens, labels, next_idx=pool.peek(0)[0])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    res = _base(B * world * args.steps / dt, B, args.steps, args.warmup, ms, "f32",
                f"DIN train step (cfg 3, row-sharded): {nu} users + {ni} items x {K} in one table sharded {world}-way "
                f"({net.tables.V_local} rows per rank), max_seq_len={L}, hidden={cfg['hidden_units']}, Zipf(1.05) ids"
                if not args.small else "DIN small, row-sharded (smoke)",
                {"embed_size": K, "table_rows": V, "final_loss": round(float(loss), 5),
                 "stream": f"{pool.cursor} distinct batches per rank drawn on the device, none trained on twice",
                 "parallelism": f"dp{world} batch + table row-sharded {world}-way (RCCL all-to-all of de-duplicated ids / rows / row "
                                f"gradients, all-reduce of dense grads, BatchNorm over the global batch)",
                 "launch": "eager launches (the general sharded step: attention in its dense form on the fetched rows, torch autograd for "
                           "the MLP — the single-GPU line's fused kernels address the table directly)"})
    res["n_gpus"], res["scaling"] = world, "weak"
    res["config"]["global_batch"] = B * world
    n_pos = B * (2 + L)
    by = n_pos * (K * 4 + 4) * 2.0 + B * (2 + L) * K * 4 * 2.0      # rows gathered into the exchange + the per-sample block, forward and backward
    res["roofline_step"] = {"bound": "hbm", "algorithmic_bytes_per_step_per_gpu": int(by), "achieved": round(by / (ms * 1e-3) / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "row traffic of the exchange path only; the step is launch- and exchange-bound"}
    res["roofline"] = dict(res["roofline_step"], kernel="whole step / exchange-path row bytes", traffic=None)
    return res


def bench_recommend_full(args, dev, net):
    """recommend_user leg at the FULL cfg 4 catalogue on one GPU: 1,024 users against the 100 M x 128 item rows of the
    training table (resident: the exported embeddings never leave the device)."""
    from librecommender_amd import ops

    t = net.tables
    I = t.variable("item_embeds_var")
    N, D = I.shape
    B, k = (1024, 100) if not args.small else (256, 10)
    g = torch.Generator(device=dev).manual_seed(7)
    U = torch.randn((B, D), device=dev, generator=g) * 0.05
    cons = torch.sort(torch.randint(0, N, (B, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values
    ptr = torch.arange(B + 1, device=dev, dtype=torch.int64) * 50
    flag = torch.ones(B, dtype=torch.uint8, device=dev)
    from bench import MFMA_BF16_PEAK_TF, TOPK_ARITH_NOTE, TOPK_ENTRY

    lib = ops._lib.load()
    ws = torch.empty(max(lib.lr_score_topk_ws_bytes(B, N, D, k), lib.lr_score_topk_filter_ws_bytes(B, N, D, k)), dtype=torch.uint8, device=dev)
    arith = ops.TOPK_ARITH
    filt, sb = arith.startswith("filter"), arith == "split_bf16"
    failed = torch.zeros(B, dtype=torch.uint8, device=dev)
    run = lambda: ops.score_topk(U, I, k, ptr, cons.reshape(-1).contiguous(), flag, ws=ws,  # noqa: E731
                                 **({"failed_out": failed} if filt else {}))
    run()
    kname = TOPK_ENTRY[arith]
    ops.TIMER.enable(kname)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ops.TIMER.disable()
    _, mean_ms = ops.TIMER.summary()[kname]
    fl = 2.0 * B * N * D
    if filt:    # one bf16 MFMA product per f32 product (+ f32 rescoring of k' candidates per user): against the dense bf16 peak
        tf = fl / (mean_ms * 1e-3) / 1e12
        extra = {"achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
                 "f32_equivalent_TFLOPs": round(tf, 2), "algorithmic_item_bytes": int(N) * D * 4,
                 "users_ranked_by_the_exact_pass": int(failed.sum())}
        fl_exec = fl
    elif sb:    # six bf16 MFMA products per f32 product: against the dense bf16 peak
        tf = 6 * fl / (mean_ms * 1e-3) / 1e12
        extra = {"achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
                 "f32_equivalent_TFLOPs": round(fl / (mean_ms * 1e-3) / 1e12, 2), "algorithmic_item_bytes": int(N) * D * 4}
        fl_exec = 6 * fl
    else:
        extra = {"achieved": round(fl / (mean_ms * 1e-3) / 1e12, 2), "algorithmic_item_bytes": int(N) * D * 4}
        fl_exec = fl
    return {"metric": "recommend_user items-scored/sec", "value": round(B * N / dt, 1), "unit": "items/s",
            "config": {"workload": f"{B} users x {N} items x {D} dims (the full cfg 4 catalogue on one GPU), k={k}, 50 consumed/user, f32",
                       "arithmetic": TOPK_ARITH_NOTE[arith]},
            "ms_per_pass": round(dt * 1e3, 3),
            "roofline": _roof_mfma(f"{kname} (score + fused top-k + merge)", fl_exec, mean_ms, extra,
                                   workload=None if args.small else "twotower", traffic_key=kname)}


def cpu_baseline_twotower(cfg, batches, net, budget=30.0):
    """`TwoTowerOracle` (fp32, TF1 dense Adam) on a bounded sample: tables cut to 1 M items / 100 k users, in-batch
    softmax at B = 8,192 (the B x B logits are materialised on the CPU path: cost per sample grows with B)."""
    from oracle.models_torch import TwoTowerOracle, export_net_weights

    Bc, ni_c, nu_c = 8192, 1_000_000, 100_000
    W = {}
    t = net.tables
    W["user_embeds_var"] = t.embed[t.user_off: t.user_off + nu_c + 1].cpu().clone()
    W["item_embeds_var"] = t.embed[t.item_off: t.item_off + ni_c].cpu().clone()
    for name, p in net.P.params.items():
        W[name] = p.detach().cpu().clone()
    o = TwoTowerOracle(W, cfg["hidden_units"], user_dense_cols=[], item_dense_cols=[], lr=1e-3, dtype=torch.float32, use_bn=False)
    t_tot, n = 0.0, 0
    for k in range(6):
        u, i, c = [x.cpu() for x in batches[k % len(batches)]]
        d = dict(users=(u[:Bc].long() % nu_c), items=(i[:Bc].long() % ni_c), corrections=c[:Bc])
        t0 = time.perf_counter()
        o.train_step("softmax", **d)
        dt = time.perf_counter() - t0
        if k > 0:
            t_tot += dt
            n += 1
        if t_tot > budget:
            break
    n = max(n, 1)
    return {"value": round(Bc * n / max(t_tot, 1e-9), 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} training steps of the PyTorch-CPU oracle restatement of algorithms/two_tower.py (TF1 dense Adam) at "
                      f"B={Bc} on tables cut to {nu_c} users / {ni_c} items (the full tables' dense Adam and the B=65,536 "
                      f"B x B logits do not fit a bounded CPU sample); first step untimed"}


# ======================================================================================================
# cfg 5: LightGCN on one GPU
# ======================================================================================================
LG_CFG = dict(n_users=10_000_000, n_items=10_000_000, n_edges=200_000_000, embed_size=64, n_layers=3, batch=65_536)


def bench_lightgcn(args, dev):
    from librecommender_amd import ops
    from librecommender_amd.nets.graph_nets import LightGCNNet

    cfg = dict(LG_CFG)
    if args.small:
        cfg.update(n_users=100_000, n_items=100_000, n_edges=2_000_000, batch=8192)
    nu, ni, E, K, L, B = (cfg[k] for k in ("n_users", "n_items", "n_edges", "embed_size", "n_layers", "batch"))
    g = torch.Generator(device=dev).manual_seed(42)
    eu, ei = distinct_interactions(E, nu, ni, g, dev)    # degrees ~ Zipf on both sides, mean E / n_users
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net = LightGCNNet(nu, ni, K, L, 0.0, None, dev, lr=1e-3, interactions=(eu, ei), want_tperm=False, torch_init=False)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    del eu, ei
    nnz = int(net.val.numel())
    pool = Pool(lambda: (zipf_ids_device(B, nu, g, dev), zipf_ids_device(B, ni, g, dev), zipf_ids_device(B, ni, g, dev)))
    batches = None

    def step():
        u, p, n = pool.next()
        return net.train_step("bpr", u, p, items_neg=n)[0]

    dt, loss, steady = _timed(step, args.steps, max(args.warmup, 2), min_seconds=args.steady_seconds, pool=pool)
    ms = dt / args.steps * 1e3
    names = ("lr_spmm_csr_bucketed_f32", "lr_spmm_csr_masked_f32", "lr_spmm_csr_adam_f32", "lr_spmm_csr_f32", "lr_adam_dense_f32",
             "lr_embed_gather_f32", "lr_embed_scatter_add_f32")
    kern = _kernel_table(ops, names, step, min(args.steps, 3), pool=pool)
    n = nu + ni
    spmm_bytes = nnz * (8 + K * 4) + n * K * 4 + (n + 1) * 8              # col + val + gathered rows (no reuse) + Y write + rowptr (no accumulator pass)
    by = {"lr_spmm_csr_bucketed_f32": spmm_bytes, "lr_spmm_csr_f32": spmm_bytes, "lr_adam_dense_f32": 7 * n * K * 4}
    kinfo = {}
    for name, (cnt, mean_ms) in kern.items():
        kinfo[name] = {"launches": cnt, "mean_ms": round(mean_ms, 4)}
        if name in by:
            kinfo[name]["algorithmic_GBps"] = round(by[name] / (mean_ms * 1e-3) / 1e9, 1)
            kinfo[name]["frac_hbm_peak"] = round(by[name] / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    dom = max((k_ for k_ in kern if k_ in by), key=lambda k_: kern[k_][1] * kern[k_][0])
    res = _base(B * args.steps / dt, B, args.steps, args.warmup, ms, "f32",
                f"LightGCN train step (cfg 5 on one GPU): {nu} users x {ni} items, {E} distinct interactions ({nnz} nnz), embed_size={K}, "
                f"{L} layers, BPR, Zipf(1.05) endpoints" if not args.small else "LightGCN small (smoke)",
                {"embed_size": K, "nnz": nnz, "laplacian_build_s": round(build_s, 3), "final_loss": round(float(loss), 5),
                 "stream": f"{pool.cursor} distinct batches drawn on the device (exact Zipf(1.05) ids), none trained on twice",
                 "laplacian": "built on the device from the interaction list (lr_csr_laplacian_build: radix sort + scan)",
                 "optimizer": "torch-style Adam over the whole node table, applied as the epilogue of the last backward product",
                 "products": "2 L per step: the last forward one computes the batch's rows only, the first backward one skips the zero "
                             "rows of its operand (row bitmaps), the last backward one ends in the optimiser step",
                 "launch": "eager launches"})
    res["roofline"] = _roof_hbm(dom, by[dom], kern[dom][1], {"note": "no-reuse byte count of SURVEY 8(d) cfg 5 (gathered rows counted once per nonzero)"},
                                workload=None if args.small else "lightgcn")
    step_bytes = 2 * L * spmm_bytes
    res["roofline_step"] = {"bound": "hbm", "algorithmic_bytes_per_step": int(step_bytes),
                            "achieved": round(step_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "SURVEY 8(d) convention: 6 full SpMM per step (664 GB).  This tree restricts two of the six to the "
                                    "batch's rows (the last forward product's output, the first backward product's operand), so the "
                                    "fraction by this convention exceeds what any one kernel reaches: per-kernel fractions are in `kernels`"}
    res["kernels"], res["sum_kernel_ms"] = kinfo, round(sum(m * c for c, m in kern.values()) / max(min(args.steps, 3), 1), 4)
    res["kernel_timing"] = "HIP events around every C-ABI launch in eager steps run after the timed region"
    if steady:
        res["steady_state"] = steady
    return res, cfg, batches, net


def bench_lightgcn_sharded(args, rank, world, dev):
    """cfg 5 as BASELINE names it (8 x MI355X): the node table and the Laplacian 1-D ROW-PARTITIONED over the ranks
    (`ShardedLightGCNNet`): per layer one all-gather of the layer's rows + the local bucketed SpMM on this rank's row
    slice, the batch's rows of the layer mean read from the all-gathered layer inputs (the last layer's by one all-to-all),
    row gradients routed to their owners, local torch-style Adam.  STRONG scaling: graph and global batch are fixed, each
    rank owns 1 / W of the rows and takes B / W samples.  Every rank builds the Laplacian from the same seeded interaction
    list on its device and keeps its row slice."""
    import torch.distributed as dist

    from librecommender_amd import ops
    from librecommender_amd.nets.graph_nets import ShardedLightGCNNet

    cfg = dict(LG_CFG)
    if args.small:
        cfg.update(n_users=100_000, n_items=100_000, n_edges=2_000_000, batch=8192)
    nu, ni, E, K, L, B = (cfg[k] for k in ("n_users", "n_items", "n_edges", "embed_size", "n_layers", "batch"))
    Bl = B // world
    g = torch.Generator(device=dev).manual_seed(42)        # the same graph and global batch on every rank
    eu, ei = distinct_interactions(E, nu, ni, g, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net = ShardedLightGCNNet(nu, ni, K, L, None, dev, lr=1e-3, interactions=(eu, ei), torch_init=False)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    del eu, ei
    torch.cuda.empty_cache()
    nnz_local = int(net.val.numel())
    sl = slice(rank * Bl, (rank + 1) * Bl)
    pool = Pool(lambda: (zipf_ids_device(B, nu, g, dev)[sl], zipf_ids_device(B, ni, g, dev)[sl], zipf_ids_device(B, ni, g, dev)[sl]))
    pool.ensure(max(args.warmup, 2) + args.steps + 1)      # a fresh global batch every step (the same stream on every rank)

    def step():
        u, p, n = pool.next()
        return net.train_step("bpr", u, p, items_neg=n)[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):
        step()
    barrier()
    ops.TIMER.enable("lr_spmm_csr_bucketed_f32", "lr_spmm_csr_masked_f32", "lr_spmm_csr_adam_f32", "lr_adam_dense_f32")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.TIMER.disable()
    kern = ops.TIMER.summary()
    if world > 1:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    n = nu + ni
    res = _base(B * args.steps / dt, Bl, args.steps, args.warmup, ms, "f32",
                f"LightGCN train step (cfg 5, node table and Laplacian row-partitioned {world}-way): {nu} users x {ni} items, "
                f"{E} distinct interactions ({nnz_local} nnz on this rank), embed_size={K}, {L} layers, BPR, global batch {B}, Zipf(1.05) endpoints",
                {"embed_size": K, "nnz_local": nnz_local, "rows_local": int(net.hi - net.lo), "laplacian_build_s": round(build_s, 3),
                 "final_loss": round(float(loss), 5),
                 "parallelism": f"1-D row partition over {world} ranks: {2 * L} all-gathers of [n, K] per step (RCCL), all-to-all of "
                                f"the batch's rows / row gradients, local bucketed SpMM + torch-style Adam",
                 "launch": "eager launches"})
    res["n_gpus"], res["scaling"] = world, "strong"
    res["config"]["global_batch"] = B
    if "lr_spmm_csr_bucketed_f32" in kern:
        cnt, mean_ms = kern["lr_spmm_csr_bucketed_f32"]
        by = nnz_local * (8 + K * 4) + net.per * K * 4 + (net.per + 1) * 8
        res["roofline"] = _roof_hbm("lr_spmm_csr_bucketed_f32", by, mean_ms,
                                    {"note": "this rank's row slice; no-reuse byte count (gathered rows counted once per nonzero)"})
        res["kernels"] = {k: {"launches": c, "mean_ms": round(m_, 4)} for k, (c, m_) in kern.items()}
    gather_bytes = 2 * L * n * K * 4 * (world - 1) / max(world, 1)
    res["exchange"] = {"all_gather_bytes_received_per_step": int(gather_bytes),
                       "note": "2 L all-gathers of the [n, K] layer rows: on a random bipartite graph a row slice references nearly every column"}
    return res


def cpu_baseline_lightgcn(cfg, budget=25.0):
    """The reference module restated for the CPU (`oracle.models_torch.LightGCNOracle`: torch.sparse.mm propagation,
    BPR, torch Adam — lightgcn_module.py:66-88, training/torch_trainer.py:77-121) on a 1/100-scale graph of the same
    law (the reference's dok-matrix Laplacian build does not finish at 10^8 interactions)."""
    from oracle.models_torch import LightGCNOracle

    s = 100
    nu, ni, E, K, L, B = (cfg["n_users"] // s, cfg["n_items"] // s, cfg["n_edges"] // s, cfg["embed_size"], cfg["n_layers"],
                          cfg["batch"])
    rng = np.random.default_rng(0)
    eu = (rng.zipf(1.05, E) - 1) % nu
    ei = (rng.zipf(1.05, E) - 1) % ni
    o = LightGCNOracle(nu, ni, K, L, eu, ei, lr=1e-3)
    t_tot, n = 0.0, 0
    for k in range(8):
        u, p, ng = rng.integers(0, nu, B), rng.integers(0, ni, B), rng.integers(0, ni, B)
        t0 = time.perf_counter()
        o.train_step(u, p, ng)
        dt = time.perf_counter() - t0
        if k > 0:
            t_tot += dt
            n += 1
        if t_tot > budget:
            break
    n = max(n, 1)
    return {"value": round(B * n / max(t_tot, 1e-9), 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} BPR training steps (B={B}) of the torch-CPU restatement of the reference LightGCN module on a 1/{s}-scale "
                      f"graph ({nu} x {ni} nodes, {o.nnz} nnz, K={K}, {L} layers); step cost is dominated by the 6 SpMMs and scales "
                      f"with nnz: the full-size figure is ~{s}x lower; first step untimed"}


# ---- f2: full-catalogue ranking of a feature model (DeepFM) at cfg-2 scale ------------------------------------------------
class _Cols:
    def __init__(self, name, index):
        self.name, self.index = list(name), list(index)


def make_feat_catalog(dev, small=False, seed=7):
    """A DeepFM at BASELINE cfg 2's shape behind the product's own ranking path, without a pandas pass over 10^6 x 202 values:
    1 M users x 1 M items, 200 sparse fields (the first 100 user-side, the last 100 item-side; vocabulary 50 000 + OOV each),
    embed_size 64, hidden (128, 64, 32).  Returns (model, cfg): `model` is a `DeepFM` object whose attributes are set directly
    (net, data_info with the per-user / per-item feature tables the reference keeps as `user_sparse_unique` /
    `item_sparse_unique`, consumed lists) — everything `FeatBase._recommend_inner` reads (reference:
    `recommendation/recommend.py:81-105`, `recommendation/preprocess.py:110-172`)."""
    import types

    from bench import CFG
    from librecommender_amd.algorithms.fm import DeepFM
    from librecommender_amd.nets import DeepFMNet
    from librecommender_amd.recommendation.recommend import ConsumedIndex

    cfg = dict(CFG)
    if small:
        cfg.update(n_users=20_000, n_items=30_000, n_sparse_fields=20, vocab=500)
    nu, ni, Fs, vocab, K = cfg["n_users"], cfg["n_items"], cfg["n_sparse_fields"], cfg["vocab"], cfg["embed_size"]
    Fu = Fs // 2
    net = DeepFMNet(nu, ni, Fs * (vocab + 1), Fs, embed_size=K, hidden_units=cfg["hidden_units"], lr=1e-3, device=dev,
                    sparse_offsets=np.arange(Fs) * (vocab + 1), seed=seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():                      # trained-looking parameters: non-trivial BatchNorm statistics, linear weights
        for name, p_ in net.P.params.items():
            if name.endswith("gamma"):
                p_.add_(torch.randn(p_.shape, device=dev, generator=g) * 0.1)
            elif name.endswith("beta") or name.endswith("bias"):
                p_.add_(torch.randn(p_.shape, device=dev, generator=g) * 0.05)
        for bn in [net.mlp.bn_in, *net.mlp.bns]:
            if bn is not None:
                bn.moving_mean.add_(torch.randn(bn.moving_mean.shape, device=dev, generator=g) * 0.01)
                bn.moving_var.mul_(1.0 + 0.2 * torch.rand(bn.moving_var.shape, device=dev, generator=g))
        net.tables.lin.add_(torch.randn(net.tables.lin.shape, device=dev, generator=g) * 0.05)
    off = (torch.arange(Fs, device=dev, dtype=torch.int64) * (vocab + 1)).to(torch.int32)
    usu = (zipf_ids_device((nu + 1) * Fu, vocab, g, dev).view(nu + 1, Fu) + off[None, :Fu]).cpu().numpy()
    isu = (zipf_ids_device((ni + 1) * (Fs - Fu), vocab, g, dev).view(ni + 1, Fs - Fu) + off[None, Fu:]).cpu().numpy()
    names = [f"s{c}" for c in range(Fs)]
    info = types.SimpleNamespace(
        sparse_col=_Cols(names, range(Fs)), dense_col=_Cols([], []),
        user_sparse_col=_Cols(names[:Fu], range(Fu)), item_sparse_col=_Cols(names[Fu:], range(Fu, Fs)),
        user_dense_col=_Cols([], []), item_dense_col=_Cols([], []),
        user_sparse_unique=usu, item_sparse_unique=isu, user_dense_unique=None, item_dense_unique=None,
        n_users=nu, n_items=ni, feat_version=0)
    model = object.__new__(DeepFM)
    model.net, model.data_info, model.device = net, info, dev
    model.n_users, model.n_items = nu, ni
    model.task = "ranking"
    rng = np.random.default_rng(seed)
    n_q = 1024 if not small else 64
    model._consumed_index = ConsumedIndex({u: np.unique(rng.integers(0, ni, 50)).tolist() for u in range(n_q)}, nu)
    cfg["query_users"] = n_q
    return model, cfg


def feat_rows(model, users, items):
    """The materialised feature rows of (user, item) pairs as the reference builds them (`_extract_feats`,
    `recommendation/preprocess.py:203-212`): (users, items, sparse [n, Fs]) host arrays."""
    from librecommender_amd.bases.feat_base import merge_user_item_feats

    sparse, _ = merge_user_item_feats(model.data_info, users, items)
    return np.asarray(users), np.asarray(items), sparse


def bench_deepfm_recommend(args, dev):
    """SURVEY 8 row f2 / a18 at cfg-2 scale: `recommend_user` of a DeepFM over the FULL catalogue — 1,024 users x 1 M items x
    202 fields, k = 100, consumed lists of 50 ids filtered — through the product path itself (`FeatBase._recommend_inner`:
    the factorised scorer of recommendation/catalog.py with the item side cached, `lr_pair_mlp_f32` for the MLP tail of every
    pair, consumed filter, top-k).  The reference materialises B x N feature rows and runs the whole model on each
    (`recommendation/recommend.py:81-105`)."""
    from bench import MFMA_F32_PEAK_TF
    from librecommender_amd import ops

    model, cfg = make_feat_catalog(dev, small=args.small)
    N, K, n_q, k = cfg["n_items"], cfg["embed_size"], cfg["query_users"], (100 if not args.small else 10)
    hid = cfg["hidden_units"]
    users = list(range(n_q))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = model._catalog_scorer()
    sc._item_side()                                    # the item-side cache (once per fit): not part of the timed passes
    torch.cuda.synchronize()
    cache_s = time.perf_counter() - t0
    model._recommend_inner(users[: max(n_q // 8, 1)], k, None, None, True, False)      # warm-up (one block)
    torch.cuda.synchronize()
    reps = 2
    sb = ops.PAIR_MLP_ARITH == "split_bf16"
    kname = "lr_pair_mlp_sb_f32" if sb else "lr_pair_mlp_f32"
    ops.TIMER.enable(kname)
    t0 = time.perf_counter()
    for _ in range(reps):
        recs = model._recommend_inner(users, k, None, None, True, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ops.TIMER.disable()
    cnt, mean_ms = ops.TIMER.summary()[kname]
    ub = max(1, (1 << 29) // max(1, N * 4))            # users per score block (FeatBase._recommend_inner)
    ub = min(ub, n_q)
    H1, H2 = hid[0], hid[1]
    fl = 2.0 * ub * N * (H1 * H2 + H2)                 # per launch: relu(P + Q) @ W2' (H1 x H2), relu, @ v3
    peak = MFMA_F32_PEAK_TF
    if sb:      # six bf16 MFMA products per f32 product of the H1 x H2 contraction: against the dense bf16 peak
        from bench import MFMA_BF16_PEAK_TF as peak  # noqa: N811
        fl_exec = 6 * 2.0 * ub * N * H1 * H2
    else:
        fl_exec = fl
    tf = fl_exec / (mean_ms * 1e-3) / 1e12
    # self-check of the timed result: no consumed id, ids in range, and the returned items' scores (recomputed by the model's
    # own forward on the materialised rows) are sorted and not below the same user's score of 1,000 random other items
    rng = np.random.default_rng(0)
    chk_users = users[:: max(n_q // 8, 1)][:8]
    worst = 0.0
    for u in chk_users:
        got = recs[u]
        cons = model.consumed_index.consumed(u)
        assert cons is None or not np.isin(got, cons).any(), "a consumed id was recommended"
        others = rng.integers(0, N, 1000)
        ids = np.concatenate([got, others])
        uu, ii, sp = feat_rows(model, np.full(len(ids), u), ids)
        idx = model.net._idx(torch.from_numpy(uu), torch.from_numpy(ii), torch.from_numpy(sp))
        s = model.net.forward(idx).float().cpu().numpy()
        top, rest = s[:k], s[k:][~np.isin(others, got) & ~(np.isin(others, cons) if cons is not None else False)]
        assert np.all(np.diff(top) <= 1e-4), "returned scores are not sorted"
        worst = max(worst, float(rest.max() - top.min()))
    ok = worst <= 1e-4
    if not ok:
        raise RuntimeError(f"deepfm_recommend: a non-returned item outscores a returned one by {worst:.3e}")
    res = {"metric": "recommend_user items-scored/sec", "value": round(n_q * N / dt, 1), "unit": "items/s", "n_gpus": 1,
           "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"DeepFM recommend_user over the full catalogue (SURVEY 8 f2 / a18 at cfg-2 scale): {n_q} users x {N} "
                                  f"items x {2 + cfg['n_sparse_fields']} fields, embed_size={K}, hidden={hid}, k={k}, 50 consumed ids per user filtered",
                      "path": "FeatBase._recommend_inner: factorised scorer (item side cached) + lr_pair_mlp_f32 + consumed filter + top-k, "
                              f"{ub} users per [B, N] score block",
                      "item_side_cache_s": round(cache_s, 3),
                      "item_side_cache_bytes": int(N * (H1 + K + 2) * 4)},
           "ms_per_pass": round(dt * 1e3, 3), "ms_per_user": round(dt * 1e3 / n_q, 4),
           "verified": {"users_checked": len(chk_users), "max_margin_violation": worst,
                        "what": "returned ids vs the model's own forward on materialised rows: sorted, consumed filtered, no sampled other item scores higher"},
           "roofline": {"kernel": kname, "bound": "mfma", "achieved": round(tf, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(tf / peak, 4), "flops_per_launch": fl_exec,
                        "f32_equivalent_TFLOPs": round(fl / (mean_ms * 1e-3) / 1e12, 2),
                        "mean_launch_ms": round(mean_ms, 3), "launches": cnt, "traffic": None,
                        "traffic_source": "rocprofv3 PMC pass committed under profiles/ (not this run)",
                        "item_cache_and_score_block_bytes_per_launch": int(N * H1 * 4 + 2 * ub * N * 4),
                        "note": "MLP tail of every (user, item) pair: 2 (H1 H2 + H2) flop per pair (split-bf16: six bf16 MFMA products per "
                                "f32 product, f32 accumulation); bytes = the item-side cache Q [N, H1] once + the [B, N] score block "
                                "read and written"},
           "kernels": {kname: {"launches": cnt, "mean_ms": round(mean_ms, 4)},
                       "pair_mlp_share_of_pass": round(mean_ms * (n_q / ub) / (dt * 1e3), 4)}}
    if not args.small:
        from bench import pmc_traffic, with_profiles

        res["roofline"]["traffic"] = pmc_traffic(kname, "deepfm_recommend")
        res["roofline"] = with_profiles(res["roofline"], kname, "deepfm_recommend")
    return res, cfg, None, model


def cpu_baseline_deepfm_recommend(cfg, model, budget=15.0):
    """The reference's path restated for the CPU: per user, materialise the feature rows of (user, every item) and run the whole
    model on them (`recommendation/recommend.py:81-105`, `preprocess.py:110-172`; `oracle.models_torch.DeepFMOracle.forward`
    from the SAME weights), then rank — timed on chunks of 50 000 items of the full-size catalogue until the budget is spent."""
    from oracle.models_torch import DeepFMOracle, export_fieldnet_weights

    o = DeepFMOracle(export_fieldnet_weights(model.net), cfg["hidden_units"], dtype=torch.float32)
    N = cfg["n_items"]
    chunk = min(50_000, N)
    t_tot, n_pairs, u, s0 = 0.0, 0, 0, 0
    while t_tot < budget:
        items = np.arange(s0, min(N, s0 + chunk))
        t0 = time.perf_counter()
        uu, ii, sp = feat_rows(model, np.full(len(items), u), items)
        with torch.no_grad():
            o.forward(torch.from_numpy(uu).long(), torch.from_numpy(ii).long(), torch.from_numpy(sp).long())
        t_tot += time.perf_counter() - t0
        n_pairs += len(items)
        s0 += chunk
        if s0 >= N:
            s0, u = 0, u + 1
    return {"value": round(n_pairs / t_tot, 1), "unit": "items/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_pairs} (user, item) pairs of the same catalogue and weights: feature rows materialised per pair + the whole "
                      f"DeepFM forward (PyTorch-CPU restatement of the reference TF graph), the reference's own algorithm for this path"}



def run(args, dev):
    which = args.workload
    if which == "din":
        res, cfg, batches, net = bench_din(args, dev)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_din(cfg, batches, net)
    elif which == "twotower":
        res, cfg, batches, net = bench_twotower(args, dev)
        if not args.no_recommend:
            res["recommend"] = bench_recommend_full(args, dev, net)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_twotower(cfg, batches, net)
    elif which == "lightgcn":
        res, cfg, batches, net = bench_lightgcn(args, dev)
        del net
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_lightgcn(cfg)
    elif which == "deepfm_recommend":
        res, cfg, batches, model = bench_deepfm_recommend(args, dev)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_deepfm_recommend(cfg, model)
        del model
        torch.cuda.empty_cache()
    else:
        raise SystemExit(f"unknown workload {which}")
    res["host_cores"] = os.cpu_count()
    return res
