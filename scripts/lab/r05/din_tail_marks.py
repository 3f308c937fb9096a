"""Phase marks of the fused tail inside the replayed DIN step (workgroup 0's s_memtime at every phase boundary)."""
import sys, types
import torch
sys.path.insert(0, ".")
import bench_workloads as bw
from librecommender_amd.nets import FeatDINNet, FeatSpec

dev = torch.device("cuda:0")
cfg = dict(bw.DIN_CFG)
K, L, B = cfg["embed_size"], cfg["max_seq_len"], cfg["batch"]
net = FeatDINNet(FeatSpec(cfg["n_users"], cfg["n_items"]), K, cfg["hidden_units"], use_bn=True, max_seq_len=L, lr=1e-3, device=dev,
                 graph_step=True)
pool = bw.Pool(bw.din_batch_maker(cfg, dev))
for _ in range(8):
    u, i, s, ln, lab = pool.next()
    net.train_step(u, i, lab, seqs=s, seq_lens=ln)
torch.cuda.synchronize()
b = net._fstep.sets[(B, L)]
w = b.tail.sync_words.cpu().numpy().astype("uint32")
m = w[2:18]
names = ["start", "colstats0", "bar0", "fin0", "fwd1", "bar1", "fin1", "fwd2(t0)", "head(t0)", "bwd2", "bar2", "sum1", "bwd1", "bar3", "sum0",
         "first_bwd"]
print("shader-clock cycles (s_memtime):", [int(x) for x in (m - m[0])])
prev = m[0]
for n, x in zip(names[1:], m[1:]):
    if x == 0:
        continue
    print(f"{n:12s} +{(int(x) - int(prev)) * 0.01:7.2f} (x100 cycles)   at {(int(x) - int(m[0])) * 0.01:7.2f}")
    prev = x
