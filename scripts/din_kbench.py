#!/usr/bin/env python
"""DIN attention kernels at cfg 3 (B = 8,192, L = 50, K = 128, 10 M-item table) under different length laws:
where does the time go — imbalance between waves (random lengths), row-read latency, or compute?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, K, N = 8192, 50, 128, 10_000_000
g = torch.Generator(device=dev).manual_seed(0)
table = torch.randn((N + 1, K), device=dev, generator=g) * 0.05
W1 = torch.randn((4 * K, 16), device=dev, generator=g) * 0.05
b1 = torch.zeros(16, device=dev)
W2 = torch.randn(16, device=dev, generator=g) * 0.1
b2 = torch.zeros(1, device=dev)
item = torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)
seq = torch.randint(0, N, (B, L), device=dev, generator=g, dtype=torch.int32)
gout = torch.randn((B, K), device=dev, generator=g)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


laws = {"U{1..50}": torch.randint(1, L + 1, (B,), device=dev, generator=g, dtype=torch.int32),
        "all 26": torch.full((B,), 26, device=dev, dtype=torch.int32),
        "all 16": torch.full((B,), 16, device=dev, dtype=torch.int32),
        "all 48": torch.full((B,), 48, device=dev, dtype=torch.int32)}
laws["U{1..50} sorted"] = torch.sort(laws["U{1..50}"]).values
small = torch.randn((1000, K), device=dev, generator=g) * 0.05          # L2-resident table: latency floor
for name, lens in laws.items():
    tiles = int(((lens + 15) // 16).sum())
    for tab, tn, it, sq in ((table, "10M-row table", item, seq), (small, "1k-row table", item % 1000, seq % 1000)):
        out, attn = ops.din_attn_pool_fwd(tab, it, sq, lens, W1, b1, W2, b2)
        f = t(lambda: ops.din_attn_pool_fwd(tab, it, sq, lens, W1, b1, W2, b2))
        bw = t(lambda: ops.din_attn_pool_bwd(tab, it, sq, lens, W1, b1, W2, b2, attn, gout))
        print(f"{name:16s} {tn:14s} tiles {tiles:6d}  fwd {f:7.1f} us ({f / tiles * 1e3:6.2f} ns/tile)  bwd {bw:7.1f} us ({bw / tiles * 1e3:6.2f} ns/tile)")
