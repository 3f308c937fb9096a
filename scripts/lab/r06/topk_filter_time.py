"""Time the filtered form of `ops.score_topk` against the exact arithmetics at the bench's recommend shape
(1,024 users x N items x 128, k = 100): per-pass time, fallback users, agreement of the ids."""
import sys
import time

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
B, D, k = 1024, 128, 100
g = torch.Generator(device=dev).manual_seed(42)
U = torch.randn((B, D), device=dev, generator=g)
cons = torch.sort(torch.randint(0, N, (B, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values
I = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I[lo:lo + 10_000_000].normal_(generator=g)
ptr = torch.arange(B + 1, device=dev, dtype=torch.int64) * 50
flag = torch.ones(B, dtype=torch.uint8, device=dev)
cidx = cons.reshape(-1).contiguous()
lib = ops._lib.load()
ws = torch.empty(max(lib.lr_score_topk_ws_bytes(B, N, D, k), lib.lr_score_topk_filter_ws_bytes(B, N, D, k)), dtype=torch.uint8, device=dev)
failed = torch.zeros(B, dtype=torch.uint8, device=dev)
res = {}
for arith in ("split_bf16", "filter", "split_bf16", "filter", "f32_chain"):
    kw = {"failed_out": failed} if arith == "filter" else {}
    ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith=arith, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith=arith, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res[arith] = out
    print(f"{arith}: {ms:.2f} ms per pass, {B * N / ms / 1e6:.1f} G items/s" + (f", users sent to the exact pass: {int(failed.sum())}" if kw else ""), flush=True)
for other in ("split_bf16", "f32_chain"):
    a, b = res["filter"], res[other]
    same = (a[1] == b[1]).float().mean().item()
    sets = (torch.sort(a[1], 1).values == torch.sort(b[1], 1).values).float().mean().item()
    print(f"filter vs {other}: ids equal at {same * 100:.3f} % of positions, as sets {sets * 100:.3f} %; max |score diff| {float((a[0] - b[0]).abs().max()):.3e}")
