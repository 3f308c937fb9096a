"""Phase time stamps of one wave of the DIN attention backward (data kernel) inside the cfg 3 bench step (lab build with -DLR_DIN_MARKS)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
import argparse  # noqa: E402

import bench_workloads as bw  # noqa: E402

args = argparse.Namespace(small=False, steps=5, warmup=3, steady_seconds=0.0, no_graph=True, workload="din")
res, cfg, batches, net = bw.bench_din(args, torch.device("cuda:0"))
torch.cuda.synchronize()
lib = C.CDLL(os.environ["LIBRECO_HIP_LIB"])
buf = (C.c_ulonglong * 64)()
rc = lib.lr_din_debug_marks(buf)
m = list(buf)
print("rc", rc, "ms_per_step", res["ms_per_step"])
print("stage weights:", m[1] - m[0], "cycles")
for s in range(7):
    b = 2 + 8 * s
    if m[b] == 0:
        break
    print(f"sample {s}: len {m[b + 5]:3d}  load q/gout {m[b + 1] - m[b]:6d}  pass1 {m[b + 2] - m[b + 1]:6d}  pass2 {m[b + 3] - m[b + 2]:6d}  epilogue {m[b + 4] - m[b + 3]:6d}  total {m[b + 4] - m[b]:6d}")
print("kernel (this wave):", m[60] - m[0], "cycles")
