#!/usr/bin/env python
"""Timing of the three big fp32 GEMMs of the DeepFM first layer (B=16384, D=12928, H=128) and
of split-K variants of the dW contraction (output 12928x128 has too few tiles for 256 CUs)."""
import torch

dev = torch.device("cuda:0")
B, D, H = 16384, 12928, 128
x = torch.randn(B, D, device=dev)
gz = torch.randn(B, H, device=dev)
W = torch.randn(D, H, device=dev)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


fl = 2.0 * B * D * H
for name, fn in [
    ("fwd  x@W", lambda: x @ W),
    ("dx   gz@W.t()", lambda: gz @ W.t()),
    ("dW   x.t()@gz", lambda: x.t() @ gz),
    ("dW   (gz.t()@x).t()", lambda: (gz.t() @ x).t()),
] + [(f"dW   bmm split-K S={S}", (lambda S: lambda: torch.bmm(x.view(S, B // S, D).transpose(1, 2), gz.view(S, B // S, H)).sum(0))(S))
     for S in (2, 4, 8, 16)]:
    ms = t(fn)
    print(f"{name:28s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
ref = x.t() @ gz
alt = torch.bmm(x.view(4, B // 4, D).transpose(1, 2), gz.view(4, B // 4, H)).sum(0)
print("max rel diff split-K:", ((ref - alt).abs().max() / ref.abs().max()).item())
# batch statistics of x
print("var_mean ms", t(lambda: torch.var_mean(x, dim=0, unbiased=False)))
print("sum+sumsq via GEMV ms", t(lambda: (torch.ones(1, B, device=dev) @ x, (x * x).sum(0))))
Wt = W.t().contiguous()
for name, fn in [
    ("fwd  F.linear(x, Wt)", lambda: torch.nn.functional.linear(x, Wt)),
    ("dx   gz@Wt_contig", lambda: gz @ Wt),
    ("dx   linear(gz, W)", lambda: torch.nn.functional.linear(gz, W)),
    ("dx   2 col chunks", lambda: torch.cat([gz @ Wt[:, : D // 2], gz @ Wt[:, D // 2:]], 1)),
    ("dx   (Wt.t()@gz.t()).t()", lambda: (W @ gz.t()).t()),
]:
    ms = t(fn)
    print(f"{name:28s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
