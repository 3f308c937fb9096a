#!/usr/bin/env python
"""Per-kernel measurements at the SURVEY §8(d) shapes that are not part of bench.py's step:
  din     cfg 3: DIN attention pooling fwd/bwd, B=8192, L=50 (len ~ U{1..50}), K'=128, N=10M items
  spmm    cfg 5 (one GPU's 1/8 row slice): 2.5M rows x 20M cols, 50M nnz, K=64
  gather  plain embedding lookup, 3.3M uniform-random rows of a 12M x 64 table
  scatter row-wise Adam scatter of 3.3M gradient rows (uniform ids) into the same table
Prints achieved algorithmic GB/s (bytes as defined in DESIGN.md §3) from HIP-event timings."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["din", "spmm", "gather", "scatter", "pair_mlp"]
g = torch.Generator(device=dev).manual_seed(42)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if "din" in which:
    B, L, K, N, H = 8192, 50, 128, 10_000_000, 16
    tab = torch.randn((N + 1, K), device=dev, generator=g) * 0.05
    item = torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)
    seq = torch.randint(0, N, (B, L), device=dev, generator=g, dtype=torch.int32)
    ln = torch.randint(1, L + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    W1 = torch.randn((4 * K, H), device=dev, generator=g) * 0.05
    b1 = torch.zeros(H, device=dev); W2 = torch.randn((H, 1), device=dev, generator=g) * 0.1; b2 = torch.zeros(1, device=dev)
    out, attn = ops.din_attn_pool_fwd(tab, item, seq, ln, W1, b1, W2, b2)
    gout = torch.randn_like(out)
    rows = float(ln.sum().item()) + B                 # gathered key rows + the query row
    t_f = timeit(lambda: ops.din_attn_pool_fwd(tab, item, seq, ln, W1, b1, W2, b2))
    t_b = timeit(lambda: ops.din_attn_pool_bwd(tab, item, seq, ln, W1, b1, W2, b2, attn, gout))
    fwd_bytes = rows * K * 4 + B * K * 4 + B * L * 8
    bwd_bytes = rows * K * 4 * 2 + B * K * 4 * 2            # re-gather + write per-position grads
    print(f"din fwd  {t_f:.3f} ms  {fwd_bytes / t_f / 1e6:8.1f} GB/s  ({B / t_f * 1e3:.3e} samples/s)")
    print(f"din bwd  {t_b:.3f} ms  {bwd_bytes / t_b / 1e6:8.1f} GB/s")
    del tab

if "spmm" in which:
    rows, cols, nnz, K = 2_500_000, 20_000_000, 50_000_000, 64
    deg = torch.full((rows,), nnz // rows, dtype=torch.int64, device=dev)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(deg, 0)])
    col = torch.randint(0, cols, (nnz,), device=dev, generator=g, dtype=torch.int32)
    val = torch.rand(nnz, device=dev, generator=g)
    X = torch.randn((cols, K), device=dev, generator=g)
    Y = torch.empty((rows, K), device=dev)
    acc = torch.zeros((rows, K), device=dev)
    t = timeit(lambda: ops.spmm_csr(rowptr, col, val, X, out=Y, acc=acc), reps=3)
    by = nnz * (8 + K * 4) + rows * K * 4 * 3 + rows * 8
    print(f"spmm     {t:.3f} ms  {by / t / 1e6:8.1f} GB/s  (nnz={nnz}, no-reuse upper bound of gathered rows)")
    del col, val
    # Zipf-degree graph (a recommender's item side): same nnz, Pareto degrees, a head of very long rows
    degz = torch.from_numpy(__import__("numpy").random.default_rng(0).pareto(1.05, rows) * 4).to(torch.int64).clamp_(0, 2_000_000)
    degz = (degz.double() * (nnz / float(degz.sum()))).to(torch.int64).to(dev)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(degz, 0)])
    nz = int(rowptr[-1])
    col = torch.randint(0, cols, (nz,), device=dev, generator=g, dtype=torch.int32)
    val = torch.rand(nz, device=dev, generator=g)
    t = timeit(lambda: ops.spmm_csr(rowptr, col, val, X, out=Y, acc=acc), reps=3)
    by = nz * (8 + K * 4) + rows * K * 4 * 3 + rows * 8
    print(f"spmm zipf {t:.3f} ms  {by / t / 1e6:8.1f} GB/s  (nnz={nz}, max degree {int(degz.max())}, rows > 128 nnz: {int((degz > 128).sum())})")
    t0 = timeit(lambda: ops._call("lr_spmm_csr_f32", rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), rows, X.data_ptr(), K,
                                  Y.data_ptr(), acc.data_ptr(), ops._stream()), reps=3)
    print(f"spmm zipf, one row group per row (round-1 kernel) {t0:.3f} ms")
    del X, Y, acc, col, val

if "gather" in which or "scatter" in which:
    V, K, n = 12_000_202, 64, 3_309_568
    tab = torch.randn((V, K), device=dev, generator=g) * 0.01
    idx = torch.randint(0, V, (n,), device=dev, generator=g, dtype=torch.int32)
    if "gather" in which:
        t = timeit(lambda: ops.embed_gather(tab, idx))
        print(f"gather   {t:.3f} ms  {n * (K * 4 * 2 + 4) / t / 1e6:8.1f} GB/s")
    if "scatter" in which:
        m, v = torch.zeros_like(tab), torch.zeros_like(tab)
        grad = torch.randn((n, K), device=dev, generator=g) * 0.01
        seg = ops.build_segments(idx, V)
        nd = seg.count()
        hp = ops.adam_hp(1e-3, 1)
        t = timeit(lambda: ops.embed_scatter_adam(tab, m, v, grad, seg, hp))
        print(f"scatter  {t:.3f} ms  {(n * K * 4 + nd * 6 * K * 4) / t / 1e6:8.1f} GB/s  ({nd} distinct rows)")
        t = timeit(lambda: ops.build_segments(idx, V))
        print(f"segments {t:.3f} ms  (radix sort + scan of {n} ids)")

if "pair_mlp" in which:
    # DeepFM full-catalogue tail (row f2): 1,024 users x 1 M items, hidden stack (128, 64, 32) collapsed to one
    # H1 x H2 product per pair (csrc/pair_mlp.hip)
    B, N, H1, H2 = 1024, 1_000_000, 128, 64
    P = torch.randn((B, H1), device=dev, generator=g)
    Q = torch.randn((N, H1), device=dev, generator=g)
    W2 = torch.randn((H1, H2), device=dev, generator=g) * 0.1
    b2 = torch.randn(H2, device=dev, generator=g) * 0.1
    v3 = torch.randn(H2, device=dev, generator=g)
    out = torch.zeros((B, N), device=dev)
    t = timeit(lambda: ops.pair_mlp(P, Q, W2, b2, v3, 0.25, out, accumulate=False), reps=3)
    fl = 2.0 * B * N * H1 * H2
    print(f"pair_mlp {t:.3f} ms  {B * N / t / 1e6:.2f} G pairs/s  {fl / t / 1e9:.1f} TFLOP/s = {fl / t / 1e9 / 157.3:.3f} of the f32 MFMA peak")
