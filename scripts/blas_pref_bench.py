#!/usr/bin/env python
"""Which BLAS backend serves the three big fp32 GEMMs of the DeepFM first layer best?"""
import torch

dev = torch.device("cuda:0")
B, D, H = 16384, 12928, 128
x = torch.randn(B, D, device=dev); gz = torch.randn(B, H, device=dev); W = torch.randn(D, H, device=dev); b = torch.randn(H, device=dev)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n


for lib in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:  # noqa: BLE001
        print(lib, "unavailable", ex); continue
    print(lib, "fwd addmm %.3f" % t(lambda: torch.addmm(b, x, W)), "dx gz@W.t() %.3f" % t(lambda: gz @ W.t()),
          "dW bmm16 %.3f" % t(lambda: torch.bmm(x.view(16, B // 16, D).transpose(1, 2), gz.view(16, B // 16, H)).sum(0)),
          "dW x.t()@gz %.3f" % t(lambda: x.t() @ gz))
