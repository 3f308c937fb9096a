#!/usr/bin/env python
"""Row-gradient parity at BASELINE cfg 2's full size, through the first moments of Adam
(m = (1 - beta1) * g after step 1: linear in g): fused path vs unfused path vs the oracle."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import CFG, global_rows, make_batches  # noqa: E402
from librecommender_amd.nets import DeepFMNet  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(CFG)
Fs, K, B, vocab = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"], cfg["vocab"]
kw = dict(embed_size=K, hidden_units=cfg["hidden_units"], lr=1e-3, epsilon=1e-5, seed=42, device=dev,
          sparse_offsets=np.arange(Fs) * (vocab + 1))
users, items, sparse, labels = make_batches(cfg, 1, seed=4242)[0]
rows_np = global_rows(cfg, users, items, sparse)
idx = torch.from_numpy(rows_np).to(dev).contiguous()
lab = torch.from_numpy(labels).to(dev)
touched = torch.from_numpy(np.unique(rows_np.reshape(-1))).to(dev)


def run(fused):
    net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), Fs, fused_l1=fused, **kw)
    loss = float(net.train_step(idx, lab))
    m = net.tables.m[touched].clone()
    lm = net.tables.lin_m[touched].clone()
    pm = net.P.m.clone()
    names = dict(net.P.params)
    del net
    torch.cuda.empty_cache()
    return loss, m, lm, pm


lf, mf, lmf, pmf = run(True)
lu, mu, lmu, pmu = run(False)
print(f"loss fused {lf:.8f} unfused {lu:.8f}")


def report(tag, a, b):
    d = (a.double() - b.double()).abs()
    scale = float(b.abs().max())
    rel = d / (b.double().abs() + 1e-3 * scale)
    print(f"{tag}: scale {scale:.3e}  max abs diff {float(d.max()):.3e}  rms diff {float(d.pow(2).mean().sqrt()):.3e}  "
          f"rms ref {float(b.double().pow(2).mean().sqrt()):.3e}  frac(rel>1e-3) {float((rel > 1e-3).double().mean()):.2e}  "
          f"frac(rel>1e-2) {float((rel > 1e-2).double().mean()):.2e}")


report("table m   fused vs unfused", mf, mu)
report("lin m     fused vs unfused", lmf, lmu)
report("dense m   fused vs unfused", pmf, pmu)
u_end, i_end = cfg["n_users"] + 1, cfg["n_users"] + 1 + cfg["n_items"] + 1
for name, lo, hi in (("user", 0, u_end), ("item", u_end, i_end), ("sparse", i_end, 1 << 40)):
    sel = (touched.long() >= lo) & (touched.long() < hi)
    report(f"  {name:6s} rows fused vs unfused", mf[sel], mu[sel])

if "--wgrad" in sys.argv:
    # X^T gz at full size against an fp64 reference: the fused kernel (5 partial slabs) and the library GEMM
    from librecommender_amd import ops
    from librecommender_amd.layers import FieldTables

    t = FieldTables(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), K, dev, sparse_offsets=np.arange(Fs) * (vocab + 1))
    g = torch.Generator(device=dev).manual_seed(5)
    # gz with a realistic structure: per-sample scale ~1/B, small column means
    gz = (torch.randn((B, 128), device=dev, generator=g) * (1.0 / B)).contiguous()
    idxT = ops.idx_transpose(idx)
    part = ops.deepfm_l1_wgrad(t.embed, idxT, gz)
    x = t.embed[idx.long()].reshape(B, -1)
    ref = torch.zeros((x.shape[1], 128), dtype=torch.float64, device=dev)
    for s0 in range(0, B, 2048):
        ref += x[s0:s0 + 2048].double().t() @ gz[s0:s0 + 2048].double()
    lib = x.t() @ gz
    report("X^T gz  fused kernel (sum of partials in fp32) vs fp64", part.sum(0), ref.float())
    report("X^T gz  fused kernel (partials summed in fp64) vs fp64", part.double().sum(0).float(), ref.float())
    report("X^T gz  library fp32 GEMM vs fp64", lib, ref.float())
    del x, ref, lib, part, t
    torch.cuda.empty_cache()

if "--oracle" in sys.argv:
    from oracle.models_torch import DeepFMOracle, export_fieldnet_weights

    net = DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (vocab + 1), Fs, **kw)
    W = export_fieldnet_weights(net)
    del net
    for dt in (torch.float32, torch.float64):
        t0 = time.time()
        o = DeepFMOracle(W, cfg["hidden_units"], lr=1e-3, epsilon=1e-5, dtype=dt)
        lo_ = float(o.train_step(torch.from_numpy(users).long(), torch.from_numpy(items).long(),
                                 torch.from_numpy(sparse).long(), torch.from_numpy(labels)))
        st = o.opt.state
        om = torch.cat([st[id(o.V.v[f"{k}_embeds_var"])][0] for k in ("user", "item", "sparse")])[touched.cpu()].float().to(dev)
        # TF forms (1 - beta1) in the variable dtype: rescale the fp64 oracle's m to the fp32 constant
        if dt == torch.float64:
            om = (om.double() * (float(np.float32(1) - np.float32(0.9)) / (1.0 - 0.9))).float()
        print(f"oracle {dt}: loss {lo_:.8f} ({time.time() - t0:.1f}s)")
        report(f"table m   fused   vs oracle {dt}", mf, om)
        report(f"table m   unfused vs oracle {dt}", mu, om)
        del o
