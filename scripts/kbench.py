#!/usr/bin/env python
"""Isolated launches of one hot kernel on the bench workload's shapes (for rocprofv3 --pmc).
usage: python scripts/kbench.py {fwd|bwd|seg|score|scoref} [reps]   (scoref = with consumed filter)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import CFG, make_batches  # noqa: E402
from librecommender_amd import ops  # noqa: E402
from librecommender_amd.layers import FieldTables  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
cfg = dict(CFG)
Fs, K, B = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"]
F = Fs + 2
if which in ("fwd", "bwd", "seg"):
    t = FieldTables(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), K, dev)
    users, items, sparse, _ = make_batches(cfg, 1, 42)[0]
    idx = t.global_idx(torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(sparse).to(dev))
    seg = t.segments(idx)
    print("positions", idx.numel(), "distinct rows", seg.count())
    start = seg.start[: seg.count() + 1].cpu().numpy()
    ln = np.diff(start)
    print("run length: mean %.2f  max %d  >32: %d runs holding %d positions" % (ln.mean(), ln.max(), (ln > 32).sum(), ln[ln > 32].sum()))
    gdeep = torch.randn((B, F, K), device=dev) * 0.01
    gpair = torch.randn((B, K), device=dev) * 0.01
    glin = torch.randn((B, F), device=dev) * 0.01
    a = torch.randn((F, K), device=dev) * 0.01
    c = torch.randn((F, K), device=dev) * 0.01
    e, pair, fsum, lin = ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
    ws = torch.empty(ops._lib.load().lr_fm_embed_bwd_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    tab, lin_t = t.embed, t.lin
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        if which == "fwd":
            ops.fm_embed_fwd(t.embed, idx, lin=t.lin)
        elif which == "seg":
            t.segments(idx)
        else:
            ops.fm_embed_bwd_adam(tab, t.m, t.v, gdeep, gpair, fsum, B, F, seg, ops.adam_hp(1e-3, i + 1),
                                  lin=lin_t, lin_m=t.lin_m, lin_v=t.lin_v, glin=glin, bn_a=a, bn_c=c, ws=ws)
        ev[i][1].record()
    torch.cuda.synchronize()
    print(which, "ms:", [round(x.elapsed_time(y), 3) for x, y in ev])
else:
    Bu, N, D, k = 1024, 12_500_000, 128, 100
    g = torch.Generator(device=dev).manual_seed(42)
    U = torch.randn((Bu, D), device=dev, generator=g)
    I = torch.randn((N, D), device=dev, generator=g)
    ws = torch.empty(ops._lib.load().lr_score_topk_ws_bytes(Bu, N, D, k), dtype=torch.uint8, device=dev)
    cons = torch.sort(torch.randint(0, N, (Bu, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
    ptr = torch.arange(Bu + 1, device=dev, dtype=torch.int64) * 50
    flag = torch.ones(Bu, dtype=torch.uint8, device=dev)
    filt = which == "scoref"
    for i in range(reps):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        if filt:
            ops.score_topk(U, I, k, ptr, cons, flag, ws=ws)
        else:
            ops.score_topk(U, I, k, ws=ws)
        b_.record()
        torch.cuda.synchronize()
        print("score ms", round(a_.elapsed_time(b_), 3))
