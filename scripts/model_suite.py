#!/usr/bin/env python
"""Train-step measurements of the other BASELINE configs on one MI355X (synthetic data, §8d):
  din       cfg 3: DIN, 10M items, K=128, L=50 (len ~ U{1..50}), B=8192, MLP (128,64,32), pure ids
  twotower  cfg 4 (one GPU's share): 12.5M items + 1M users, K=128, towers (128,), in-batch softmax B=65536
  lightgcn  cfg 5 (1/8 scale): 1.25M users x 1.25M items, 25M interactions, K=64, 3 layers, BPR B=65536
Prints ms/step and samples/s (HIP events, steady state)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd.nets import FeatDINNet, FeatSpec, TwoTowerNet  # noqa: E402
from librecommender_amd.nets.graph_nets import LightGCNNet  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["din", "twotower", "lightgcn"]
rng = np.random.default_rng(0)


def run(step, warm=3, reps=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if "din" in which:
    nu, ni, K, L, B = 1_000_000, 10_000_000, 128, 50, 8192
    net = FeatDINNet(FeatSpec(nu, ni), K, (128, 64, 32), use_bn=True, max_seq_len=L, lr=1e-3, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    users = torch.randint(0, nu, (B,), device=dev, generator=g)
    items = torch.randint(0, ni, (B,), device=dev, generator=g)
    lens = torch.randint(1, L + 1, (B,), device=dev, generator=g)
    seqs = torch.randint(0, ni, (B, L), device=dev, generator=g)
    seqs = torch.where(torch.arange(L, device=dev)[None, :] < lens[:, None], seqs, torch.full_like(seqs, ni))
    labels = torch.randint(0, 2, (B,), device=dev, generator=g).float()
    ms = run(lambda: net.train_step(users, items, labels, seqs=seqs, seq_lens=lens))
    print(f"din      {ms:8.3f} ms/step  {B / ms * 1e3:.3e} samples/s  (fused attention kernels: {net.fused})")
    del net
    torch.cuda.empty_cache()

if "twotower" in which:
    nu, ni, K, B = 1_000_000, 12_500_000, 128, 65536
    net = TwoTowerNet(nu, ni, 0, 0, 0, [], [], 0, embed_size=K, hidden_units=(128,), use_bn=False, lr=1e-3, device=dev)
    g = torch.Generator(device=dev).manual_seed(2)
    users = torch.randint(0, nu, (B,), device=dev, generator=g)
    items = torch.randint(0, ni, (B,), device=dev, generator=g)
    corr = torch.rand(B, device=dev, generator=g) * 1e-3 + 1e-6
    ms = run(lambda: net.train_step("softmax", users, items, corrections=corr))
    print(f"twotower {ms:8.3f} ms/step  {B / ms * 1e3:.3e} samples/s  (in-batch softmax, streaming softmax-CE kernels: no [B,B] logits)")
    del net
    torch.cuda.empty_cache()

if "lightgcn" in which:
    nu, ni, ne, K, L, B = 1_250_000, 1_250_000, 25_000_000, 64, 3, 65536
    u = rng.integers(0, nu, ne)
    i = rng.integers(0, ni, ne)
    order = np.argsort(u, kind="stable")
    u, i = u[order], i[order]
    bounds = np.searchsorted(u, np.arange(nu + 1))
    consumed = {int(k): i[bounds[k]:bounds[k + 1]] for k in range(nu) if bounds[k + 1] > bounds[k]}
    t0 = time.perf_counter()
    net = LightGCNNet(nu, ni, K, L, 0.0, consumed, dev, lr=1e-3)
    print(f"lightgcn graph build {time.perf_counter() - t0:.1f} s, nnz {net.val.numel()}")
    bu, bp, bn = rng.integers(0, nu, B), rng.integers(0, ni, B), rng.integers(0, ni, B)
    ms = run(lambda: net.train_step("bpr", bu, bp, items_neg=bn), warm=2, reps=5)
    nnz = net.val.numel()
    by = 6 * (nnz * (8 + K * 4) + (nu + ni) * K * 4 * 3)
    print(f"lightgcn {ms:8.3f} ms/step  {B / ms * 1e3:.3e} samples/s  ({by / ms / 1e6:.0f} GB/s over the 6 SpMMs, no-reuse byte count)")
