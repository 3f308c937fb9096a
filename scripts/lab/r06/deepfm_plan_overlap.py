"""Feasibility: how much of the DeepFM step (cfg 2) is the batch plan (idx transpose + field-wise segments), and how much of it
hides when the NEXT batch's plan runs on a second stream beside the replayed step.  One fixed batch (so that a stale plan is a valid
plan); three timings of 300 replayed steps: the product step, the step captured without its plan kernels, the same + the plan of
a batch launched on a side stream after every replay."""
import time

import numpy as np
import torch

import bench
from librecommender_amd import ops
from librecommender_amd.nets import DeepFMNet

dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
Fs, K, B = cfg["n_sparse_fields"], cfg["embed_size"], cfg["batch"]


def make_net():
    return DeepFMNet(cfg["n_users"], cfg["n_items"], Fs * (cfg["vocab"] + 1), Fs, embed_size=K, hidden_units=cfg["hidden_units"],
                     lr=1e-3, epsilon=1e-5, seed=42, device=dev, sparse_offsets=np.arange(Fs) * (cfg["vocab"] + 1), fused_l1=True)


one = bench.device_batch_maker(cfg, dev, seed=42)
batches = [one() for _ in range(4)]
idx, labels = batches[0]


def timed(net, after=None, n=300):
    for _ in range(6):
        net.train_step(idx, labels)
        if after:
            after()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.train_step(idx, labels)
        if after:
            after()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


net = make_net()
net.enable_graph(True)
print(f"product step: {timed(net):.4f} ms", flush=True)

# the step captured WITHOUT its plan kernels (the plan of this batch, built once, stays in the builder's buffers)
net2 = make_net()
net2.train_step(idx, labels)            # eager: allocates the builder, builds the plan
torch.cuda.synchronize()
seg_keep = net2._fseg.build(net2._idxT, net2.tables.field_row_start)
torch.cuda.synchronize()
real_T, real_build = ops.idx_transpose, net2._fseg.build
ops.idx_transpose = lambda i, out=None: net2._idxT
net2._fseg.build = lambda idxT, frs: seg_keep
net2.enable_graph(True)
print(f"step without the plan kernels: {timed(net2):.4f} ms", flush=True)

# + the plan of another batch on a side stream after every replay (own buffers)
side = torch.cuda.Stream(device=dev, priority=-1)
fseg2 = ops.FieldSegmentBuilder(B, 2 + Fs, net2.tables.V, dev)
idxT2 = torch.empty_like(net2._idxT)
nxt = batches[1][0]


def plan_next():
    with torch.cuda.stream(side):
        real_T(nxt, out=idxT2)
        fseg2.build(idxT2, net2.tables.field_row_start)


print(f"step without the plan kernels + a plan on a side stream per step: {timed(net2, plan_next):.4f} ms", flush=True)
side2 = torch.cuda.Stream(device=dev)


def plan_next_lo():
    with torch.cuda.stream(side2):
        real_T(nxt, out=idxT2)
        fseg2.build(idxT2, net2.tables.field_row_start)


print(f"the same, side stream of default priority: {timed(net2, plan_next_lo):.4f} ms", flush=True)
