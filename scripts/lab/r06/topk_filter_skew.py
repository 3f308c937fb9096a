"""How the filter's certification fares on catalogues whose row norms are skewed (lognormal norm factor, sigma given):
users sent to the exact pass and time per pass at 1,024 users x N x 128, k = 100."""
import sys
import time

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dev = torch.device("cuda:0")
B, D, k = 1024, 128, 100
g = torch.Generator(device=dev).manual_seed(42)
U = torch.randn((B, D), device=dev, generator=g)
I0 = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I0[lo:lo + 10_000_000].normal_(generator=g)
failed = torch.zeros(B, dtype=torch.uint8, device=dev)
for sigma in (0.0, 0.1, 0.25, 0.5, 1.0):
    f = torch.exp(sigma * torch.randn((N, 1), device=dev, generator=g))
    I = I0 * f
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s, i = ops.score_topk(U, I, k, arith="filter", failed_out=failed)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    se, ie = ops.score_topk(U, I, k, arith="split_bf16")
    torch.cuda.synchronize()
    ms_e = (time.perf_counter() - t0) * 1e3
    same = (torch.sort(i, 1).values == torch.sort(ie, 1).values).float().mean().item()
    print(f"sigma {sigma}: max norm / median norm {float(f.max() / f.median()):.1f}, exact-pass users {int(failed.sum())} of {B}, "
          f"filter {ms:.1f} ms, split_bf16 {ms_e:.1f} ms, id sets equal {same * 100:.3f} %", flush=True)
    del I, f
