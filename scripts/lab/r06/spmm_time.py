"""One full product Y = A X of cfg 5's Laplacian (10 M + 10 M nodes, 400 M nnz, K = 64) with the bucketed kernel: mean of 5 launches
by HIP events.  `LIBRECO_HIP_LIB` selects a lab build of csrc/spmm.hip (scripts/lab/r06/spmm_build.sh)."""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench_workloads as bw  # noqa: E402
from librecommender_amd import ops  # noqa: E402
from librecommender_amd.nets.graph_nets import LightGCNNet  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(bw.LG_CFG)
nu, ni, E, K = cfg["n_users"], cfg["n_items"], cfg["n_edges"], cfg["embed_size"]
g = torch.Generator(device=dev).manual_seed(42)
eu, ei = bw.distinct_interactions(E, nu, ni, g, dev)
net = LightGCNNet(nu, ni, K, 3, 0.0, None, dev, lr=1e-3, interactions=(eu, ei), want_tperm=False, torch_init=False)
del eu, ei
X = torch.randn((nu + ni, K), device=dev, generator=g) * 0.1
Y = torch.empty_like(X)
plan = ops.SpmmPlan(net.rowptr, net.col.numel(), K)
ops.spmm_csr(net.rowptr, net.col, net.val, X, out=Y, plan=plan)
torch.cuda.synchronize()
n = int(os.environ.get("SPMM_REPS", "5"))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ops.spmm_csr(net.rowptr, net.col, net.val, X, out=Y, plan=plan)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(f"{os.environ.get('LIBRECO_HIP_LIB', 'product build')}: {sum(ms) / n:.3f} ms per product (min {min(ms):.3f}), checksum {float(Y.double().sum()):.6e}")
