#!/usr/bin/env python
"""Isolated launches of the fused DeepFM first-layer kernels at the bench workload's shapes
(BASELINE cfg 2): HIP-event times per launch, for `rocprofv3 --kernel-trace/--pmc` runs as well.
usage: python scripts/fused_kbench.py [all|seg|stats|fwd|wgrad|dgrad|adam|gather] [reps]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import CFG, global_rows, make_batches  # noqa: E402
from librecommender_amd import ops  # noqa: E402
from librecommender_amd.layers import FieldTables  # noqa: E402

tile = 0
if "--tile" in sys.argv:                       # pin the first-layer kernels' tiling (32 / 64 samples per workgroup)
    k_ = sys.argv.index("--tile")
    tile = int(sys.argv[k_ + 1])
    del sys.argv[k_:k_ + 2]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
ops._lib.load().lr_deepfm_l1_tile_override(tile)
print(f"first-layer tiling: {tile or 'automatic'}")
cfg = dict(CFG)
Fs, K, B, H1 = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"], cfg["hidden_units"][0]
F = Fs + 2
t = FieldTables(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), K, dev, sparse_offsets=np.arange(Fs) * (cfg["vocab"] + 1))
users, items, sparse, _ = make_batches(cfg, 1, 42)[0]
idx = torch.from_numpy(global_rows(cfg, users, items, sparse)).to(dev).contiguous()
g = torch.Generator(device=dev).manual_seed(1)
Wp = torch.randn((F * K, H1), device=dev, generator=g) * 0.01
bias = torch.randn(H1, device=dev, generator=g)
gz = torch.randn((B, H1), device=dev, generator=g) * 0.01
gl = torch.randn(B, device=dev, generator=g) * 1e-4
wp = torch.randn(K, device=dev, generator=g)
a = torch.randn(F * K, device=dev, generator=g) * 1e-3
c = torch.randn(F * K, device=dev, generator=g) * 1e-3
lin_scale = torch.randn(F, device=dev, generator=g)
sb = ops.FieldSegmentBuilder(B, F, t.V, dev)
idxT = ops.idx_transpose(idx)
seg = sb.build(idxT, t.field_row_start)
ns = seg.count()
ln = np.diff(seg.start[: ns + 1].cpu().numpy())
print(f"positions {idx.numel()}  distinct rows {ns}  run length mean {ln.mean():.2f} max {ln.max()}  >32: {(ln > 32).sum()} runs / {ln[ln > 32].sum()} positions")
WpA, WpB = ops.deepfm_l1_pack(Wp, F, K)
z1, pair, fsum, lin_out = ops.deepfm_l1_fwd(t.embed, idx, WpA, bias, H1, lin=t.lin)
ge = torch.empty((B * F + 1, K), device=dev)
nch = ops._lib.load().lr_deepfm_l1_wgrad_chunks(B, F)
part = torch.empty((nch, F * K, H1), device=dev)
ws = torch.empty(ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F), dtype=torch.uint8, device=dev)
step = [0]


def run(name):
    if name == "seg":
        ops.idx_transpose(idx, out=idxT)
        sb.build(idxT, t.field_row_start)
    elif name == "stats":
        ops.fm_field_stats(t.embed, seg, t.field_row_start, B)
    elif name == "pack":
        ops.deepfm_l1_pack(Wp, F, K, out=(WpA, WpB))
    elif name == "fwd":
        ops.deepfm_l1_fwd(t.embed, idx, WpA, bias, H1, lin=t.lin)
    elif name == "wgrad":
        ops.deepfm_l1_wgrad(t.embed, idxT, gz, n_chunks=nch, out=part)
    elif name == "dgrad":
        ops.deepfm_l1_dgrad(gz, WpB, K, F, seg.slotT, gl=gl, wp=wp, fsum=fsum, out=ge)
    elif name == "adam":
        step[0] += 1
        ops.fm_rows_adam(t.embed, t.m, t.v, ge, seg, ops.adam_hp(1e-3, step[0]), B, F, gl=gl, wp=wp, lin=t.lin,
                         lin_m=t.lin_m, lin_v=t.lin_v, bn_a=a, bn_c=c, lin_scale=lin_scale, ws=ws)
    elif name == "gather":          # the unfused forward, for comparison
        ops.fm_embed_fwd(t.embed, idx, lin=t.lin)


flops = 2.0 * B * F * K * H1
names = {"all": ["seg", "stats", "pack", "fwd", "wgrad", "dgrad", "adam", "gather"], "l1": ["fwd", "wgrad", "dgrad"]}.get(which, [which])
for name in names:
    run(name)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        run(name)
        ev[i][1].record()
    torch.cuda.synchronize()
    ms = [x.elapsed_time(y) for x, y in ev]
    best, mean = min(ms), sum(ms) / len(ms)
    extra = ""
    if name in ("fwd", "wgrad", "dgrad"):
        extra = f"  {flops / (mean * 1e-3) / 1e12:.1f} TFLOP/s ({flops / (mean * 1e-3) / 1e12 / 157.3:.1%} of f32 MFMA peak)"
    if name == "adam":
        by = 311_888 * B
        extra = f"  algorithmic {by / (mean * 1e-3) / 1e12:.2f} TB/s ({by / (mean * 1e-3) / 1e9 / 8000:.1%} of HBM peak)"
    print(f"{name:7s} ms: mean {mean:.3f}  min {best:.3f}  all {[round(x, 3) for x in ms]}{extra}")
