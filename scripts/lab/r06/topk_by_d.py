"""The three forms of ops.score_topk by embedding width at 1,024 users x N items, k = 100."""
import sys
import time

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
B, k = 1024, 100
g = torch.Generator(device=dev).manual_seed(42)
for D in (16, 32, 64, 96, 128):
    I = torch.empty((N, D), device=dev)
    for lo in range(0, N, 10_000_000):
        I[lo:lo + 10_000_000].normal_(generator=g)
    U = torch.randn((B, D), device=dev, generator=g)
    failed = torch.zeros(B, dtype=torch.uint8, device=dev)
    line = []
    for arith in ("filter", "split_bf16", "f32_chain"):
        kw = {"failed_out": failed} if arith == "filter" else {}
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s, i = ops.score_topk(U, I, k, arith=arith, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        line.append(f"{arith} {ms:.1f} ms" + (f" ({int(failed.sum())} uncertified)" if kw else ""))
    print(f"D = {D}: " + ", ".join(line), flush=True)
    del I
