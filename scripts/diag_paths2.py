#!/usr/bin/env python
"""Fused vs unfused: z1 / gz1 element-level comparison at full size."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import CFG, global_rows, make_batches  # noqa: E402
from librecommender_amd.layers import dense as D  # noqa: E402
from librecommender_amd.nets import DeepFMNet  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(CFG)
Fs, K, B, vocab = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"], cfg["vocab"]
kw = dict(embed_size=K, hidden_units=cfg["hidden_units"], lr=1e-3, epsilon=1e-5, seed=42, device=dev,
          sparse_offsets=np.arange(Fs) * (vocab + 1))
users, items, sparse, labels = make_batches(cfg, 1, seed=4242)[0]
idx = torch.from_numpy(global_rows(cfg, users, items, sparse)).to(dev).contiguous()
lab = torch.from_numpy(labels).to(dev)


def rep(tag, a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    d = (a - b).abs()
    print(f"{tag:30s} rms ref {float(b.pow(2).mean().sqrt()):.3e}  rms diff {float(d.pow(2).mean().sqrt()):.3e}  "
          f"max diff {float(d.max()):.3e}  rel rms {float(d.pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-300)):.2e}")


cap = {}
f_fwd, f_bwd = D._FusedL1.forward, D._FusedL1.backward
u_fwd, u_bwd = D._FoldedBNDense.forward, D._FoldedBNDense.backward


def wrap(tag, fwd, bwd, cls):
    def fwd2(ctx, *a):
        out = fwd(ctx, *a)
        cap[tag + "_z1"] = out.detach().clone()
        return out

    def bwd2(ctx, gz):
        cap[tag + "_gz"] = gz.detach().clone()
        return bwd(ctx, gz)
    cls.forward, cls.backward = staticmethod(fwd2), staticmethod(bwd2)


wrap("f", f_fwd, f_bwd, D._FusedL1)
wrap("u", u_fwd, u_bwd, D._FoldedBNDense)
fused = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), Fs, **kw)
lf = fused.train_step(idx, lab)
del fused
torch.cuda.empty_cache()
plain = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), Fs, fused_l1=False, **kw)
lu = plain.train_step(idx, lab)
print("loss", float(lf), float(lu))
zf, zu, gf, gu = cap["f_z1"], cap["u_z1"], cap["f_gz"], cap["u_gz"]
rep("z1", zf, zu)
rep("gz1", gf, gu)
print("relu mask differs in", int(((zf > 0) != (zu > 0)).sum()), "of", zf.numel(), "entries;  z1 == 0 exactly:", int((zf == 0).sum()), int((zu == 0).sum()))
d = (gf.double() - gu.double()).abs()
per_sample = d.pow(2).sum(1).sqrt()
ref_sample = gu.double().pow(2).sum(1).sqrt()
rel = per_sample / (ref_sample + 1e-30)
print("per-sample relative gz1 difference: median %.2e  p99 %.2e  max %.2e" % (float(rel.median()), float(rel.quantile(0.99)), float(rel.max())))
per_col = d.pow(2).sum(0).sqrt() / gu.double().pow(2).sum(0).sqrt()
print("per-column relative gz1 difference: min %.2e median %.2e max %.2e" % (float(per_col.min()), float(per_col.median()), float(per_col.max())))
# activation statistics feeding BN1
for tag, z in (("fused", zf), ("unfused", zu)):
    a = torch.relu(z).double()
    print(tag, "relu(z1) column mean rms %.6e  var rms %.6e  frac(z1>0) %.6f" % (float(a.mean(0).pow(2).mean().sqrt()), float(a.var(0, unbiased=False).pow(2).mean().sqrt()), float((z > 0).double().mean())))
