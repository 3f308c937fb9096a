#!/usr/bin/env python
"""The fused first layer's three kernels, f32 fma chain (csrc/deepfm_l1.hip) against split-bf16 products (csrc/deepfm_l1_sb.hip),
at BASELINE cfg 2's shape (B = 16,384, F = 202, K = 64, H1 = 128, exact Zipf(1.05) ids over 12 M rows): HIP events around 20
launches each, every grid / staging variant of the split-bf16 kernels, and both families' errors against f64 on a row subset."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import _lib, ops  # noqa: E402

import os  # noqa: E402

QUICK = bool(os.environ.get("LR_KBENCH_QUICK"))      # ablation / counter passes: the automatic variants only, few launches
dev = torch.device("cuda")
B, F, K, H1, V = 16384, 202, 64, 128, 12_000_202
g = torch.Generator(device=dev).manual_seed(0)
table = torch.randn((V, K), generator=g, device=dev) * 0.1
lin = torch.randn((V, 1), generator=g, device=dev) * 0.1
W = torch.randn((F * K, H1), generator=g, device=dev) * 0.05
bias = torch.randn(H1, generator=g, device=dev)
gz = torch.randn((B, H1), generator=g, device=dev) * 0.01
if os.environ.get("LR_KBENCH_ZEROS"):      # operands of zeros: the same instruction streams, no switching in the MFMA datapath
    table.zero_(); W.zero_(); gz.zero_()
import bench_workloads as bw  # noqa: E402

per = V // F
idx = (torch.arange(F, device=dev)[None, :] * per + bw.zipf_ids_device(B * F, per, g, dev).view(B, F)).to(torch.int32)
idxT = ops.idx_transpose(idx)
frs = (torch.arange(F + 1, device=dev) * per).to(torch.int32)
frs[-1] = V
seg = ops.FieldSegmentBuilder(B, F, V, dev).build(idxT, frs)
lib = _lib.load()


def timed(fn, n=(3 if QUICK else 20)):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rel(a, ref):
    return float((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


fl = 2.0 * B * F * K * H1
sub = slice(0, 512)
rows = table.double()[idx[sub].long()]                                       # [512, F, K]
ref_z = rows.reshape(512, F * K) @ W.double() + bias.double()
Pf = ops.deepfm_l1_pack(W, F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="f32_chain"))
Ps = ops.deepfm_l1_pack(W, F, K, out=ops.deepfm_l1_pack_bufs(F, K, H1, dev, arith="split_bf16"))
print(f"pack (both plane sets): {timed(lambda: ops.deepfm_l1_pack(W, F, K, out=Ps)):.4f} ms;  f32 pack: {timed(lambda: ops.deepfm_l1_pack(W, F, K, out=Pf)):.4f} ms")

# ---- forward
of = ops.deepfm_l1_fwd(table, idx, Pf[0], bias, H1, lin=lin)
tf = timed(lambda: ops.deepfm_l1_fwd(table, idx, Pf[0], bias, H1, lin=lin, out=of[:3]))
print(f"l1_fwd   f32 chain      : {tf:.4f} ms  {fl / tf / 1e9:6.1f} TFLOP/s   rel rms err vs f64 {rel(of[0][sub], ref_z):.3e}")
FWD_MODES = (((0, 0), "automatic"), ((64, 1), "64-sample tiles"), ((128, 1), "128-sample tiles, 1 field group (half the chip)"),
             ((128, 2), "128-sample tiles, 2 field groups"), ((128, 4), "128-sample tiles, 4 field groups"), ((64, 2), "64-sample tiles, 2 field groups"),
             ((64, 4), "64-sample tiles, 4 field groups"))
for mode, name in (FWD_MODES[:1] if QUICK else FWD_MODES):
    lib.lr_deepfm_l1_sb_override(mode[0], mode[1], 0, 0)
    os_ = ops.deepfm_l1_fwd(table, idx, Ps[0], bias, H1, lin=lin)
    ts = timed(lambda: ops.deepfm_l1_fwd(table, idx, Ps[0], bias, H1, lin=lin, out=os_[:3]))
    print(f"l1_fwd   split-bf16, {name}: {ts:.4f} ms  {fl / ts / 1e9:6.1f} TFLOP/s (f32-equivalent)   err {rel(os_[0][sub], ref_z):.3e}"
          f"   fsum err {rel(os_[2][sub], rows.sum(1)):.2e}  lin_out equal {torch.equal(os_[3], of[3])}")
lib.lr_deepfm_l1_sb_override(0, 0, 0, 0)

# ---- weight gradient
ref_w = torch.einsum("bfk,bh->fkh", table.double()[idx.long()][:, :8], gz.double()).reshape(8 * K, H1)      # first 8 fields, whole batch
pf = ops.deepfm_l1_wgrad(table, idxT, gz, arith="f32_chain")
tf = timed(lambda: ops.deepfm_l1_wgrad(table, idxT, gz, out=pf, arith="f32_chain"))
print(f"l1_wgrad f32 chain ({pf.shape[0]} chunks): {tf:.4f} ms  {fl / tf / 1e9:6.1f} TFLOP/s   err {rel(pf.double().sum(0)[:8 * K], ref_w):.3e}")
print(f"gz pack: {timed(lambda: ops._call('lr_deepfm_l1_sb_gz_pack', ops._ptr(gz), B, H1, ops._ptr(ops._l1_ws(dev, 'gzp', lib.lr_deepfm_l1_sb_gz_pack_bytes(B, H1))), ops._stream())):.4f} ms")
for fg, cw in (((0, 0),) if QUICK else ((2, 8), (2, 4), (4, 4))):
    lib.lr_deepfm_l1_sb_override(0, 0, cw, fg)
    for nch in ((None,) if QUICK else (None, 10)):
        ps = ops.deepfm_l1_wgrad(table, idxT, gz, n_chunks=nch, arith="split_bf16")
        ts = timed(lambda: ops.deepfm_l1_wgrad(table, idxT, gz, out=ps, arith="split_bf16"))
        print(f"l1_wgrad split-bf16, {fg} fields / {cw} multiplying waves per workgroup, {ps.shape[0]} chunks (incl. gz pack): {ts:.4f} ms  {fl / ts / 1e9:6.1f} TFLOP/s   "
              f"err {rel(ps.double().sum(0)[:8 * K], ref_w):.3e}")
lib.lr_deepfm_l1_sb_override(0, 0, 0, 0)

# ---- row gradient
gl = torch.randn(B, generator=g, device=dev) * 0.01
wp = torch.randn(K, generator=g, device=dev)
ge_f = ops.deepfm_l1_dgrad(gz, Pf[1], K, F, seg.slotT, gl=gl, wp=wp, fsum=of[2])
tf = timed(lambda: ops.deepfm_l1_dgrad(gz, Pf[1], K, F, seg.slotT, gl=gl, wp=wp, fsum=of[2], out=ge_f))
sl = seg.slotT[:, sub].long()                                                # [F, 512]
ref_g = (torch.einsum("bh,fkh->fbk", gz[sub].double(), W.double().view(F, K, H1))
         + (gl[sub].double()[:, None] * wp.double()[None, :] * of[2][sub].double())[None])
ok = sl >= 0
print(f"l1_dgrad f32 chain      : {tf:.4f} ms  {fl / tf / 1e9:6.1f} TFLOP/s   err {rel(ge_f[sl.clamp(min=0)][ok], ref_g[ok]):.3e}")
DG_MODES = (((0, 0), "automatic field groups"), ((1, 0), "1 field group (half the chip)"), ((4, 0), "4 field groups"))
for mode, name in (DG_MODES[:1] if QUICK else DG_MODES):
    lib.lr_deepfm_l1_sb_override(0, mode[0], 0, 0)
    ge_s = ops.deepfm_l1_dgrad(gz, Ps[1], K, F, seg.slotT, gl=gl, wp=wp, fsum=of[2])
    ts = timed(lambda: ops.deepfm_l1_dgrad(gz, Ps[1], K, F, seg.slotT, gl=gl, wp=wp, fsum=of[2], out=ge_s))
    print(f"l1_dgrad split-bf16, {name}: {ts:.4f} ms  {fl / ts / 1e9:6.1f} TFLOP/s   err {rel(ge_s[sl.clamp(min=0)][ok], ref_g[ok]):.3e}")
lib.lr_deepfm_l1_sb_override(0, 0, 0, 0)
