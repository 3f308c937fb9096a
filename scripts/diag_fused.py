#!/usr/bin/env python
"""Stage-by-stage fp64 check of the fused DeepFM step at BASELINE cfg 2's full size."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import CFG, global_rows, make_batches  # noqa: E402
from librecommender_amd import ops  # noqa: E402
from librecommender_amd.layers.dense import FusedL1IO  # noqa: E402
from librecommender_amd.nets import DeepFMNet  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(CFG)
Fs, K, B, vocab = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"], cfg["vocab"]
F_ = Fs + 2
net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), Fs, embed_size=K, hidden_units=cfg["hidden_units"],
                lr=1e-3, epsilon=1e-5, seed=42, device=dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
users, items, sparse, labels = make_batches(cfg, 1, seed=4242)[0]
idx = torch.from_numpy(global_rows(cfg, users, items, sparse)).to(dev).contiguous()
lab = torch.from_numpy(labels).to(dev)
t = net.tables


def rep(tag, a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    d = (a - b).abs()
    print(f"{tag:42s} rms ref {float(b.pow(2).mean().sqrt()):.3e}  rms diff {float(d.pow(2).mean().sqrt()):.3e}  "
          f"max diff {float(d.max()):.3e}  rel rms {float(d.pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-300)):.2e}")


e64 = t.embed[idx.long()].double()                     # [B,F,K]
x64 = e64.reshape(B, F_ * K)
idxT = ops.idx_transpose(idx)
sb = ops.FieldSegmentBuilder(B, F_, t.V, dev)
seg = sb.build(idxT, t.field_row_start)
mean, var = ops.fm_field_stats(t.embed, seg, t.field_row_start, B)
rep("stats mean", mean, x64.mean(0))
rep("stats var", var, x64.var(0, unbiased=False))
gen_seg = ops.build_segments(idx.reshape(-1), t.V)
mean2, var2 = ops.fm_field_stats(t.embed, gen_seg, t.field_row_start, B)
rep("stats var (general segments)", var2, x64.var(0, unbiased=False))
print("n_seg field", seg.count(), "general", gen_seg.count())

P, mlp = net.P, net.mlp
bn, l0 = mlp.bn_in, mlp.layers[0]
inv = torch.rsqrt(var + bn.eps)
io = FusedL1IO(t.embed, t.lin, idx, idxT, F_, K)
net.P.zero_grad()
z1 = mlp.fused_first(io, training=True, stats=(mean, var))
mean64, var64 = x64.mean(0), x64.var(0, unbiased=False)
inv64 = torch.rsqrt(var64 + bn.eps)
W64, b64, g64, be64 = P[l0.w].double(), P[l0.b].double(), P[bn.gamma].double(), P[bn.beta].double()
z1_64 = ((x64 - mean64) * (g64 * inv64) + be64) @ W64 + b64
rep("z1", z1.detach(), z1_64)
rep("fsum", io.fsum, e64.sum(1))
logits = net._fused_tail(z1, io, training=True)
logits.retain_grad()
loss = net.loss_fn(logits, lab, "cross_entropy")
loss.backward()
gz = io.gz
gl = logits.grad.contiguous()
gz64 = gz.double()
# first-layer parameter gradients from gz (fp64 BatchNorm backward on the materialised block)
xhat = (x64 - mean64) * inv64
Gy = gz64 @ W64.t()                                     # d loss / d BN output  [B, F*K]
dW_ref = ((xhat * g64 + be64)).t() @ gz64
rep("dW1", P[l0.w].grad, dW_ref)
dgamma_ref, dbeta_ref = (Gy * xhat).sum(0), Gy.sum(0)
rep("dgamma", P[bn.gamma].grad, dgamma_ref)
rep("dbeta", P[bn.beta].grad, dbeta_ref)
dx_ref = (g64 * inv64) * (Gy - dbeta_ref / B - xhat * (dgamma_ref / B))      # [B, F*K]
w_out = P[net.out.w]
wp = w_out[1:1 + K, 0].clone()
gp64 = gl.double()[:, None] * wp.double()[None, :]
dfm = gp64[:, None, :] * (e64.sum(1)[:, None, :] - e64)
ge_ref_pos = dx_ref.reshape(B, F_, K) + dfm             # per-position total gradient
# fused pieces
s = P[bn.gamma] * inv
c = s * inv * (P[bn.gamma].grad / B)
a = s * (P[bn.beta].grad / B) - c * mean
rep("bn_a (io)", io.bn_a, (g64 * inv64) * (dbeta_ref / B) - (g64 * inv64) * inv64 * (dgamma_ref / B) * mean64)
rep("bn_c (io)", io.bn_c, (g64 * inv64) * inv64 * (dgamma_ref / B))
ge = ops.deepfm_l1_dgrad(gz, io.WpB, K, F_, seg.slotT, gl=gl, wp=wp, fsum=io.fsum)
Wp64 = W64 * (g64 * inv64)[:, None]
G_ref = (gz64 @ Wp64.t()).reshape(B, F_, K) + gp64[:, None, :] * e64.sum(1)[:, None, :]
slot = seg.slotT.t().long()                             # [B,F]
rep("ge (per position, run order)", ge[slot.reshape(-1)], G_ref.reshape(-1, K))
# per-row gradient
rows = seg.rows[: seg.count()].long()
gref_rows = torch.zeros((t.V, K), dtype=torch.float64, device=dev)
gref_rows.index_add_(0, idx.reshape(-1).long(), ge_ref_pos.reshape(-1, K))
m0 = torch.zeros_like(t.m)
v0 = torch.zeros_like(t.v)
wcopy = t.embed.clone()
lin_scale = (w_out[0, 0] * P[net.linear.w][:, 0]).contiguous()
ws = torch.empty(ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F_), dtype=torch.uint8, device=dev)
ops.fm_rows_adam(wcopy, m0, v0, ge, seg, ops.adam_hp(1e-3, 1, eps=1e-5), B, F_, gl=gl, wp=wp, bn_a=io.bn_a, bn_c=io.bn_c, ws=ws)
omb1 = float(np.float32(1) - np.float32(0.9))
rep("row gradient (m / (1-b1))", m0[rows] / omb1, gref_rows[rows])
ln = (seg.start[1: seg.count() + 1] - seg.start[: seg.count()]).long()
for lo_, hi_ in ((1, 1), (2, 8), (9, 32), (33, 100000)):
    sel = (ln >= lo_) & (ln <= hi_)
    rep(f"  rows with run length {lo_}..{hi_} ({int(sel.sum())})", (m0[rows] / omb1)[sel], gref_rows[rows][sel])
