#!/usr/bin/env python
"""Times lr_deepfm_l1_fwd_f32 against the experimental lr_deepfm_l1_fwd_sb_f32 at BASELINE cfg 2's shape (B = 16,384, F = 202,
K = 64, H1 = 128, Zipf ids over 12 M rows), HIP events around 20 launches each, and prints both errors against f64 on a
row subset."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, F, K, H1, V = 16384, 202, 64, 128, 12_000_202
g = torch.Generator(device=dev).manual_seed(0)
table = torch.randn((V, K), generator=g, device=dev) * 0.1
lin = torch.randn((V, 1), generator=g, device=dev) * 0.1
W = torch.randn((F * K, H1), generator=g, device=dev) * 0.05
bias = torch.randn(H1, generator=g, device=dev)
import bench_workloads as bw  # noqa: E402
per = V // F
idx = (torch.arange(F, device=dev)[None, :] * per + bw.zipf_ids_device(B * F, per, g, dev).view(B, F)).to(torch.int32)
WpA, _ = ops.deepfm_l1_pack(W, F, K)
Wsb = ops.deepfm_l1_sb_pack(W, F, K)


def timed(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


from librecommender_amd import _lib  # noqa: E402

out_a = ops.deepfm_l1_fwd(table, idx, WpA, bias, H1, lin=lin)
out_b = ops.deepfm_l1_fwd_sb(table, idx, Wsb, bias, H1, lin=lin)
ta = timed(lambda: ops.deepfm_l1_fwd(table, idx, WpA, bias, H1, lin=lin, out=out_a[:3]))
res = {}
for mode, name in ((4, "4 waves"), (8, "8 waves"), (20, "4 waves, B fragments direct"), (24, "8 waves, B fragments direct"), (40, "8 waves, 4 multiply + 4 stage"), (56, "128 samples per workgroup, 8 waves, B direct")):
    _lib.load().lr_deepfm_l1_sb_waves_override(mode)
    res[mode] = timed(lambda: ops.deepfm_l1_fwd_sb(table, idx, Wsb, bias, H1, lin=lin, out=out_b[:3]))
    print(f"split-bf16, {name}: {res[mode]:.4f} ms")
best = min(res, key=res.get)
_lib.load().lr_deepfm_l1_sb_waves_override(best)
z_best = ops.deepfm_l1_fwd_sb(table, idx, Wsb, bias, H1, lin=lin)[0]
_lib.load().lr_deepfm_l1_sb_waves_override(8)
out_b = ops.deepfm_l1_fwd_sb(table, idx, Wsb, bias, H1, lin=lin)
print("all modes give the same bits:", torch.equal(z_best, out_b[0]))
tb = res[best]
fl = 2.0 * B * F * K * H1
sub = slice(0, 512)
ref = table.double()[idx[sub].long()].reshape(512, F * K) @ W.double() + bias.double()
rms = float(ref.pow(2).mean().sqrt())
ea = float((out_a[0][sub].double() - ref).pow(2).mean().sqrt()) / rms
eb = float((out_b[0][sub].double() - ref).pow(2).mean().sqrt()) / rms
print(f"l1_fwd f32 MFMA   : {ta:.4f} ms  {fl / ta / 1e9:.1f} TFLOP/s  rel rms err vs f64 {ea:.3e}")
print(f"l1_fwd split-bf16 : {tb:.4f} ms  {fl / tb / 1e9:.1f} TFLOP/s (f32-equivalent)  rel rms err vs f64 {eb:.3e}")
print("pair / fsum / lin_out identical:", all(torch.equal(a, b) for a, b in zip(out_a[1:], out_b[1:])))
