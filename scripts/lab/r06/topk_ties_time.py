"""Degenerate ties: n all-zero users among 1,024 (every score of theirs is 0) — time per pass of the three forms at N x 128."""
import sys
import time

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
B, D, k = 1024, 128, 100
g = torch.Generator(device=dev).manual_seed(42)
U0 = torch.randn((B, D), device=dev, generator=g)
I = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I[lo:lo + 10_000_000].normal_(generator=g)
for n_bad in (0, 1, 64, 1024):
    U = U0.clone()
    bad = torch.randperm(B, device=dev, generator=g)[:n_bad]
    U[bad] = 0
    line = []
    for arith in ("filter", "split_bf16", "f32_chain"):
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s, i = ops.score_topk(U, I, k, arith=arith)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        ok = bool((i[bad] == torch.arange(k, device=dev)[None, :]).all()) if n_bad else True
        line.append(f"{arith} {ms:.1f} ms ({'ids 0..k-1' if ok else 'WRONG'})")
    print(f"{n_bad} all-zero users: " + ", ".join(line), flush=True)
