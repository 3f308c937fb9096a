"""Row-partitioned LightGCN over 2 gloo ranks on CPU with the oracle kernels injected,
checked against the fixture produced by the REFERENCE module itself (tests/golden/lightgcn.npz:
propagated embeddings, BPR loss, gradients, one torch-Adam step)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import unflatten
from tests.oracle_kernels import OracleKernels
from tests.test_sharded_cpu import free_port

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lightgcn.npz")


def run_rank(rank, world, port, out_dir, chunks=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.nets.graph_nets import ShardedLightGCNNet

    g = np.load(GOLD)
    nu, ni, L = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    net = ShardedLightGCNNet(nu, ni, 16, L, unflatten(g["user_consumed_flat"]), torch.device("cpu"),
                             kern=OracleKernels(), seed=42, lr=1e-2, epsilon=1e-8, chunks=chunks)
    # more than one rank: the layer inputs travel in pieces and the slice is multiplied column block by column block (default 4)
    assert net.chunks == (chunks if chunks is not None else (1 if world == 1 else min(4, net.per)))
    ue, ie = net.embeddings()
    B = len(g["users"])
    sl = slice(rank * B // world, (rank + 1) * B // world)
    loss, G = net.train_step("bpr", g["users"][sl], g["pos"][sl], items_neg=g["neg"][sl])
    losses = [None] * world
    dist.all_gather_object(losses, float(loss))
    G_full = net._all_gather_rows(G)[: nu + ni]
    E_full = net._all_gather_rows(net.E)[: nu + ni]
    assert (net._blocks is not None) == (net.chunks > 1)
    if net.chunks > 1:      # the column blocks partition the slice's entries
        assert sum(int(rp[-1]) for rp, _, _ in net._blocks) == net.col.numel() and len(net._blocks) == net._nC
    if rank == 0:
        torch.save({"ue": ue, "ie": ie, "loss": float(np.mean(losses)), "G": G_full, "E": E_full},
                   os.path.join(out_dir, f"w{world}.pt"))
    dist.destroy_process_group()


# equal per-rank batches of the 20-sample fixture (loss = mean of local means); 10 ranks: 7-8 nodes each.  `chunks`: pieces of the
# all-gather = column blocks of the slice (None: the default, 1 for one rank / 4 otherwise; 3 does not divide the block sizes)
@pytest.mark.parametrize("world,chunks", [(1, None), (2, None), (4, None), (10, None), (1, 3), (2, 1), (2, 3), (4, 7)])
def test_sharded_lightgcn_matches_reference_fixture(world, chunks):
    out = tempfile.mkdtemp()
    mp.spawn(run_rank, args=(world, free_port(), out, chunks), nprocs=world, join=True)
    r = torch.load(os.path.join(out, f"w{world}.pt"))
    g = np.load(GOLD)
    nu = int(g["n_users"])
    np.testing.assert_allclose(r["ue"].numpy(), g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["ie"].numpy(), g["item_embeds"], rtol=1e-5, atol=1e-6)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(r["G"][:nu].numpy(), g["gU"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(r["G"][nu:].numpy(), g["gI"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(r["E"][:nu].numpy(), g["U1"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["E"][nu:].numpy(), g["I1"], rtol=1e-4, atol=2e-6)


def run_rank_dropout(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.nets.graph_nets import ShardedLightGCNNet

    g = np.load(GOLD)
    nu, ni, L = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    net = ShardedLightGCNNet(nu, ni, 16, L, unflatten(g["user_consumed_flat"]), torch.device("cpu"),
                             kern=OracleKernels(), seed=42, lr=1e-2, epsilon=1e-8, dropout=0.3, amsgrad=True)
    E0 = net._all_gather_rows(net.E)[: nu + ni].clone()
    B = len(g["users"])
    sl = slice(rank * B // world, (rank + 1) * B // world)
    losses = []
    for _ in range(3):
        loss, _ = net.train_step("bpr", g["users"][sl], g["pos"][sl], items_neg=g["neg"][sl])
        part = [None] * world
        dist.all_gather_object(part, float(loss))
        losses.append(float(np.mean(part)))
    E_full = net._all_gather_rows(net.E)[: nu + ni]
    vmax = net._all_gather_rows(net.vmax)[: nu + ni]
    # the dense Laplacian this rank's slice came from (for the reference below)
    rows = net._row_of_nnz
    A_loc = torch.zeros((net.per, nu + ni))
    A_loc[rows - net.lo, net.col.long()] = net.val
    parts = [None] * world
    dist.all_gather_object(parts, A_loc[: net.hi - net.lo])
    if rank == 0:
        torch.save({"E0": E0, "E": E_full, "vmax": vmax, "losses": losses, "A": torch.cat(parts), "seed": net._drop_seed},
                   os.path.join(out_dir, f"drop_w{world}.pt"))
    dist.destroy_process_group()


def test_sharded_lightgcn_edge_dropout_and_amsgrad():
    """Round 4: edge dropout (`lightgcn_module.py:90-96`) and AMSGrad (`torch_trainer.py:63-69`) under a process group.  The
    dropout mask is a counter-based function of (step, row, column), so 1, 2 and 4 ranks compute the same steps, and the
    steps are those of torch autograd through the DENSE dropped Laplacian (forward A_drop, backward A_drop^T) + torch's own
    Adam(amsgrad=True)."""
    from librecommender_amd.nets.graph_nets import ShardedLightGCNNet

    out = tempfile.mkdtemp()
    for world in (1, 2, 4):
        mp.spawn(run_rank_dropout, args=(world, free_port(), out), nprocs=world, join=True)
    r1, r2, r4 = (torch.load(os.path.join(out, f"drop_w{w}.pt")) for w in (1, 2, 4))
    for r in (r2, r4):
        torch.testing.assert_close(r["E"], r1["E"], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(r["vmax"], r1["vmax"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(r["losses"], r1["losses"], rtol=1e-6)
    g = np.load(GOLD)
    nu, ni, L = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    n = nu + ni
    A = r1["A"].double()
    torch.testing.assert_close(A, A.t())                              # the static Laplacian is symmetric
    E = r1["E0"].double().clone().requires_grad_(True)
    opt = torch.optim.Adam([E], lr=1e-2, eps=1e-8, amsgrad=True)
    rr, cc = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
    users, pos, neg = (torch.from_numpy(np.asarray(g[k])).long() for k in ("users", "pos", "neg"))
    kept = []
    for step in range(1, 4):
        M = ShardedLightGCNNet._keep_mask(rr.reshape(-1), cc.reshape(-1), n, r1["seed"] + step, 0.7).view(n, n).double()
        kept.append(float(M[A != 0].mean()))
        Ad = A * M / 0.7
        layers, cur = [E], E
        for _ in range(L):
            cur = Ad @ cur
            layers.append(cur)
        mean = torch.stack(layers).mean(0)
        u, p, q = mean[users], mean[nu + pos], mean[nu + neg]
        loss = torch.nn.functional.softplus(-((u * p).sum(1) - (u * q).sum(1))).mean()      # torchops/loss.py: bpr = -log sigmoid(pos - neg)
        assert abs(float(loss.detach()) - r1["losses"][step - 1]) < 1e-6
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.testing.assert_close(r1["E"].double(), E.detach(), rtol=1e-4, atol=2e-6)
    assert 0.55 < np.mean(kept) < 0.85                                # about 70 % of the stored entries survive a step
