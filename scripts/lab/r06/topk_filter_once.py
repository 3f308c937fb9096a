"""One filtered score_topk pass (after one warm pass) for rocprofv3 --pmc / --kernel-trace runs."""
import sys

import torch

from librecommender_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
B, D, k = 1024, 128, 100
g = torch.Generator(device=dev).manual_seed(42)
U = torch.randn((B, D), device=dev, generator=g)
I = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I[lo:lo + 10_000_000].normal_(generator=g)
for _ in range(2):
    ops.score_topk(U, I, k, arith="filter")
torch.cuda.synchronize()
