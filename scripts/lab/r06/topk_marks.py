"""Where the waves of the filter's one-term pass spend their time (lab build -DLR_TK_MARKS): shader-clock time per phase of the
stage loop, summed over all waves of the main pass, 1,024 users x N x 128, k = 100."""
import ctypes as C
import sys

import torch

from librecommender_amd import _lib, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
B, D, k = 1024, 128, 100
g = torch.Generator(device=dev).manual_seed(42)
U = torch.randn((B, D), device=dev, generator=g)
I = torch.empty((N, D), device=dev)
for lo in range(0, N, 10_000_000):
    I[lo:lo + 10_000_000].normal_(generator=g)
lib = _lib.load()
fn = lib._lib.lr_score_topk_debug_marks if hasattr(lib, "_lib") else C.CDLL(str(_lib.LIB_PATH if not __import__("os").environ.get("LIBRECO_HIP_LIB") else __import__("os").environ["LIBRECO_HIP_LIB"])).lr_score_topk_debug_marks
fn.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 8)()
ops.score_topk(U, I, k, arith="filter")
torch.cuda.synchronize()
fn(out, 1)
ops.score_topk(U, I, k, arith="filter")
torch.cuda.synchronize()
fn(out, 1)
v = list(out)
names = ["wait for the stage (full)", "MFMAs + epilogues", "wait for the ring slot (done)", "prefetch wait + stage write", "stages", "window-edge lockstep"]
tot = v[0] + v[1] + v[2] + v[3] + v[5]
for n_, x in zip(names, v):
    print(f"{n_}: {x:.3e}" + (f" = {100 * x / tot:.1f} %" if n_ != "stages" else ""))
print(f"of the MFMAs + epilogues phase, the MFMA chains (LDS reads, dot2, MFMAs, until the accumulators are read): {v[6]:.3e} = {100 * v[6] / max(v[1], 1):.1f} %")
print(f"per wave and stage: {tot / max(v[4], 1):.0f} clocks of s_memtime (100 MHz x ?): phases {[round(x / max(v[4], 1)) for x in (v[0], v[1], v[2], v[3], v[5])]}")
