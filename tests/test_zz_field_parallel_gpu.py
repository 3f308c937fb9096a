"""Field-parallel DeepFM (nets/field_parallel.py) with the HIP kernels: one and two ranks sharing
cuda:0 (collectives over gloo, staged through host memory), against each other and against the
same run on CPU with the oracle kernels (tests/test_field_parallel_cpu.py pins that one to the
reference-graph oracle).  (Added after this round's GPU budget was spent: first run is the driver's.)"""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_field_parallel_cpu import BG, FRS, HID, K, make_batches
from tests.test_sharded_cpu import free_port

pytestmark = pytest.mark.gpu


def run_rank(rank, world, port, out_dir, on_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.nets.field_parallel import FieldParallelDeepFMNet
    from librecommender_amd.parallel import HipKernels
    if on_gpu:
        dev, kern = torch.device("cuda", 0), None
        torch.cuda.set_device(dev)
    else:
        from tests.oracle_kernels import OracleKernels
        dev, kern = torch.device("cpu"), OracleKernels()
    net = FieldParallelDeepFMNet(FRS, embed_size=K, hidden_units=HID, use_bn=True, lr=1e-2, device=dev, kern=kern, seed=42)
    assert (not on_gpu) or isinstance(net.kern, HipKernels)
    if on_gpu:       # device generators differ from the CPU ones: start from the CPU run's initial model
        init = torch.load(os.path.join(out_dir, "init.pt"))
        lo, hi = int(FRS[net.f_lo]), int(FRS[net.f_hi])
        net.embed.copy_(init["emb"][lo:hi])
        net.lin.copy_(init["lin"][lo:hi])
        with torch.no_grad():
            for k_, p in net.PL.params.items():
                width = p.shape[0] // net.Fr
                p.copy_(init["sharded"][k_][net.f_lo * width: net.f_hi * width])
            for k_, p in net.P.params.items():
                p.copy_(init["dense"][k_])
    else:
        emb, lin = net.gather_full()
        torch.save({"emb": emb, "lin": lin, "sharded": net.gather_sharded_dense(),
                    "dense": {k_: p.detach().clone() for k_, p in net.P.params.items()}}, os.path.join(out_dir, "init.pt"))
    per = BG // world
    sl = slice(rank * per, (rank + 1) * per)
    losses, trace = [None] * world, []
    for idx, labels in make_batches():
        loss = float(net.train_step(torch.from_numpy(idx[sl]).to(dev), torch.from_numpy(labels[sl]).to(dev)))
        dist.all_gather_object(losses, loss)
        trace.append(float(np.mean(losses)))
    logits = [None] * world
    dist.all_gather_object(logits, net.forward(torch.from_numpy(make_batches()[0][0][sl]).to(dev)).cpu())
    emb, lin = net.gather_full()
    sharded = net.gather_sharded_dense()
    if rank == 0:
        torch.save({"emb": emb, "lin": lin, "sharded": sharded, "losses": trace, "logits": torch.cat(logits),
                    "dense": {k_: p.detach().cpu() for k_, p in net.P.params.items()}},
                   os.path.join(out_dir, f"{'gpu' if on_gpu else 'cpu'}_w{world}.pt"))
    dist.destroy_process_group()


def test_field_parallel_hip_matches_oracle_kernels_and_rank_count(dev):
    out = tempfile.mkdtemp()
    mp.spawn(run_rank, args=(1, free_port(), out, False), nprocs=1, join=True)
    for world in (1, 2):
        mp.spawn(run_rank, args=(world, free_port(), out, True), nprocs=world, join=True)
    ref = torch.load(os.path.join(out, "cpu_w1.pt"))
    for world in (1, 2):
        got = torch.load(os.path.join(out, f"gpu_w{world}.pt"))
        np.testing.assert_allclose(got["losses"], ref["losses"], rtol=1e-4)
        torch.testing.assert_close(got["emb"], ref["emb"], rtol=1e-3, atol=5e-6)
        torch.testing.assert_close(got["lin"], ref["lin"], rtol=1e-3, atol=5e-6)
        for group in ("sharded", "dense"):
            for k_ in ref[group]:
                torch.testing.assert_close(got[group][k_], ref[group][k_], rtol=2e-3, atol=1e-5, msg=lambda m, n=k_: f"{n}: {m}")
        torch.testing.assert_close(got["logits"], ref["logits"], rtol=2e-3, atol=5e-5)
