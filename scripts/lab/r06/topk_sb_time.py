"""Time `ops.score_topk` under both arithmetics at the bench's recommend shape (1,024 users x N items x 128, k = 100)."""
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
ws = torch.empty(ops._lib.load().lr_score_topk_ws_bytes(B, N, D, k), dtype=torch.uint8, device=dev)
res = {}
for arith in ("f32_chain", "split_bf16", "f32_chain", "split_bf16"):
    ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith=arith)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ops.score_topk(U, I, k, ptr, cidx, flag, ws=ws, arith=arith)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res[arith] = out
    print(f"{arith}: {ms:.2f} ms per pass, {B * N / ms / 1e6:.1f} G items/s, f32-equivalent {2 * B * N * D / ms / 1e9:.1f} TFLOP/s", flush=True)
a, b = res["f32_chain"], res["split_bf16"]
same = (a[1] == b[1]).float().mean().item()
print(f"ids equal at {same * 100:.3f} % of the {B * k} positions; max |score diff| {float((a[0] - b[0]).abs().max()):.3e}")
