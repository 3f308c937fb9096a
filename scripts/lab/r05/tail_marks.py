"""Phase time stamps of the fused tail (workgroup 0) at cfg 2 / cfg 3 batch sizes."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from librecommender_amd.layers.tail import DeepFMTail
from librecommender_amd.nets import DeepFMNet
dev = torch.device("cuda")
names = ["start", "colstats", "barrier1", "finalize0", "fwd 128->64", "barrier2", "finalize1", "fwd 64->32", "head", "bwd 64<-32", "barrier3",
         "reduce1", "bwd 128<-64", "barrier4", "reduce0", "first_bwd"]
for B, plain in ((16384, False), (8192, True)):
    Fs, K = 200, 64
    net = DeepFMNet(50, 60, Fs * 10, Fs, embed_size=K, hidden_units=(128, 64, 32), device=dev, sparse_offsets=np.arange(Fs) * 10)
    g = torch.Generator(device=dev).manual_seed(1)
    z1 = torch.randn((B, 128), device=dev, generator=g)
    pair = None if plain else torch.randn((B, K), device=dev, generator=g)
    lin = None if plain else torch.randn((B, Fs + 2), device=dev, generator=g)
    lab = (torch.rand(B, device=dev, generator=g) > 0.5).float()
    tail = DeepFMTail(net.P, net.mlp, None if plain else net.linear, net.out, 0 if plain else Fs + 2, 0 if plain else K, dev)
    for _ in range(5):
        tail.run(z1, pair, lin, lab)
    torch.cuda.synchronize()
    m = tail.sync_words.cpu().numpy().astype(np.uint32)[2:]
    d = (m[1:] - m[:-1]).astype(np.int64)
    print(f"B={B} plain={plain}: total {(int(m[len(names)-1]) - int(m[0])) & 0xffffffff} clocks (100 MHz ticks?)")
    for n, x in zip(names[1:], d):
        print(f"   {n:14s} {int(x) & 0xffffffff}")
