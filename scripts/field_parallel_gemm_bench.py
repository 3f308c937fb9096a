#!/usr/bin/env python
"""First-layer GEMM shapes of ONE rank of the field-parallel DeepFM (nets/field_parallel.py) at world
sizes 1/2/4/8 of the benchmark workload, measured on a single GPU: x [B*W, ceil(202/W)*64] against
W1_r [.., 128].  Same flops per rank for every W, but the dW contraction's output shrinks to
(F_r*64) x 128 — too few tiles for 256 CUs unless the reduction over the (8x longer) batch is split."""
import torch

dev = torch.device("cuda:0")
B, F, K, H = 16384, 202, 64, 128


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for W in (1, 2, 4, 8):
    Bg, D = B * W, -(-F // W) * K
    x, gz, W1 = torch.randn(Bg, D, device=dev), torch.randn(Bg, H, device=dev), torch.randn(D, H, device=dev)
    fl = 2.0 * Bg * D * H
    print(f"--- world {W}: x [{Bg}, {D}]")
    cases = [("fwd x@W1", lambda: x @ W1), ("dx gz@W1.t()", lambda: gz @ W1.t()), ("dW x.t()@gz", lambda: x.t() @ gz)]
    cases += [(f"dW bmm slabs S={S}", (lambda S: lambda: torch.bmm(x.view(S, Bg // S, D).transpose(1, 2), gz.view(S, Bg // S, H)).sum(0))(S))
              for S in (4, 16, 64)]
    for lib in ("cublas", "cublaslt"):
        torch.backends.cuda.preferred_blas_library(lib)
        for name, fn in cases:
            ms = t(fn)
            print(f"{lib:9s} {name:22s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
    print(f"var_mean {t(lambda: torch.var_mean(x, dim=0, unbiased=False)):7.3f} ms")
    del x, gz, W1
    torch.cuda.empty_cache()
