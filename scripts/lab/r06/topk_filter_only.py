"""Time the filtered `ops.score_topk` alone (1,024 users x N x 128, k = 100) — for builds of the library with parts taken out."""
import sys
import time

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
B, D = 1024, 128
g = torch.Generator(device=dev).manual_seed(42)
U = torch.randn((B, D), device=dev, generator=g)
cons = torch.sort(torch.randint(0, N, (B, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values
I = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I[lo:lo + 10_000_000].normal_(generator=g)
ptr = torch.arange(B + 1, device=dev, dtype=torch.int64) * 50
flag = torch.ones(B, dtype=torch.uint8, device=dev)
cidx = cons.reshape(-1).contiguous()
ws = torch.empty(ops._lib.load().lr_score_topk_filter_ws_bytes(B, N, D, k), dtype=torch.uint8, device=dev)
failed = torch.zeros(B, dtype=torch.uint8, device=dev)
for rep in range(2):
    ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith="filter", failed_out=failed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith="filter", failed_out=failed)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"filter: {ms:.2f} ms per pass, exact-pass users {int(failed.sum())}", flush=True)
