#!/usr/bin/env python
"""fp32 GEMM shapes of the DeepFM dense tail at B=16384 (forward, dx, dW) and split-K variants of
the weight-gradient contractions (tiny outputs, K = batch)."""
import torch

dev = torch.device("cuda:0")
B = 16384


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (i, o) in [(128, 64), (64, 32), (97, 1), (202, 1)]:
    x = torch.randn(B, i, device=dev); W = torch.randn(i, o, device=dev); gz = torch.randn(B, o, device=dev); b = torch.randn(o, device=dev)
    print(f"[{i}->{o}] fwd addmm {t(lambda: torch.addmm(b, x, W)):7.1f} us | dx {t(lambda: gz @ W.t()):7.1f} us | "
          f"dW x.t()@gz {t(lambda: x.t() @ gz):7.1f} us | dW (gz.t()@x).t() {t(lambda: (gz.t() @ x).t()):7.1f} us | "
          + " ".join(f"splitK{S} {t(lambda S=S: torch.bmm(x.view(S, B // S, i).transpose(1, 2), gz.view(S, B // S, o)).sum(0)):6.1f}" for S in (16, 64, 256))
          + f" | gb sum {t(lambda: gz.sum(0)):6.1f} us")
