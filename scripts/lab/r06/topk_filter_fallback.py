"""Cost of the filter's exact re-run by the number of users it cannot certify at 1,024 users x N x 128, k = 100.  An uncertified
user is made by planting 400 items u (1 + 1e-5 j) in the catalogue: its best 400 scores lie within 0.4 % of each other, inside the
bound's width (0.4 % + the 0.4 % of the bound itself), so the 256th bound is above the 100th exact score."""
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
failed = torch.zeros(B, dtype=torch.uint8, device=dev)
ws = torch.empty(ops._lib.load().lr_score_topk_filter_ws_bytes(B, N, D, k), dtype=torch.uint8, device=dev)
I_clean = None
for n_bad in (0, 1, 16, 128, 129, 512, 1024):
    U = U0
    bad = torch.randperm(B, device=dev, generator=g)[:n_bad]
    rows = torch.randperm(N, device=dev, generator=g)[: n_bad * 400].view(n_bad, 400)
    saved = I[rows.view(-1)].clone()
    if n_bad:
        I[rows.view(-1)] = (U[bad][:, None, :] * (1 + 1e-5 * torch.arange(400, device=dev)[None, :, None])).view(-1, D)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s, i = ops.score_topk(U, I, k, ws=ws, arith="filter", failed_out=failed)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    se, ie = ops.score_topk(U, I, k, arith="split_bf16")
    same = bool((torch.sort(i[bad], 1).values == torch.sort(ie[bad], 1).values).all()) if n_bad else True
    want = bool((torch.sort(i[bad], 1).values == torch.sort(rows[:, 300:], 1).values).all()) if n_bad else True
    print(f"{n_bad} uncertified users (reported {int(failed.sum())}): {ms:.1f} ms per pass; their rows equal the exact kernel's: {same}, "
          f"are the planted winners: {want}", flush=True)
    I[rows.view(-1)] = saved
