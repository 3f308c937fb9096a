"""Multi-process path with the HIP kernels: two ranks SHARING cuda:0 (the GPU box
has one device; RCCL refuses two ranks on one GPU, so the collectives run over gloo, staged
through host memory by `parallel._a2a_single`).  Everything else — segment build, row cache,
fused FM kernels in rows mode, owner-side scatter-Adam, item-sharded top-k + merge — is the
product path.  Checks 2 ranks == 1 rank on the concatenated batch, and the unsharded net."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_sharded_cpu import BL, FS, HID, K, NI, NU, V, free_port, make_data

pytestmark = pytest.mark.gpu


def run_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDeepFMNet
    from librecommender_amd.parallel import HipKernels, sharded_score_topk

    full, lin, batches = make_data()
    net = ShardedDeepFMNet(V, FS, embed_size=K, hidden_units=HID, use_bn=True, lr=1e-2,
                           device=dev, seed=42)
    assert isinstance(net.kern, HipKernels)
    net.tables.load_full(torch.from_numpy(full), torch.from_numpy(lin))
    per = 2 * BL // world
    losses = []
    for idx, labels in batches:
        sl = slice(rank * per, (rank + 1) * per)
        losses.append(float(net.train_step(torch.from_numpy(idx[sl]).to(dev), torch.from_numpy(labels[sl]).to(dev))))
    emb, l = net.tables.gather_full()
    rng = np.random.default_rng(5)
    U = torch.from_numpy(rng.standard_normal((6, 8)).astype(np.float32)).to(dev)
    I = torch.from_numpy(rng.standard_normal((1001, 8)).astype(np.float32)).to(dev)
    bounds = np.linspace(0, 1001, world + 1).astype(int)
    s, i = sharded_score_topk(net.kern, U, I[bounds[rank]:bounds[rank + 1]].contiguous(), 9, int(bounds[rank]))
    if rank == 0:
        torch.save({"emb": emb, "lin": l, "losses": losses, "topk_s": s.cpu(), "topk_i": i.cpu(),
                    "dense": {k_: p.detach().cpu() for k_, p in net.P.params.items()}},
                   os.path.join(out_dir, f"w{world}.pt"))
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs(dev):
    out = tempfile.mkdtemp()
    for world in (1, 2):
        mp.spawn(run_rank, args=(world, free_port(), out), nprocs=world, join=True)
    return out


def test_two_ranks_equal_one_rank_hip(runs):
    """BatchNorm runs over the GLOBAL batch (partial sums averaged over the ranks: the `sync` hooks of the fused kernels),
    so two ranks compute the step one rank computes on the concatenated batch: embedding tables after 3 steps within
    rounding, exact ids on top-k."""
    a = torch.load(os.path.join(runs, "w1.pt"))
    b = torch.load(os.path.join(runs, "w2.pt"))
    assert torch.equal(a["topk_i"], b["topk_i"])
    torch.testing.assert_close(a["topk_s"], b["topk_s"])
    assert np.isfinite(a["losses"]).all() and np.isfinite(b["losses"]).all()
    # rows never touched stay identical; touched rows moved by Adam steps of size lr in both runs
    assert (a["emb"] - b["emb"]).abs().max() < 5e-4
    for k_ in a["dense"]:
        torch.testing.assert_close(a["dense"][k_], b["dense"][k_], rtol=1e-3, atol=5e-4)


def run_rank_nobn(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDeepFMNet

    full, lin, batches = make_data()
    net = ShardedDeepFMNet(V, FS, embed_size=K, hidden_units=HID, use_bn=False, lr=1e-2, device=dev, seed=42)
    net.tables.load_full(torch.from_numpy(full), torch.from_numpy(lin))
    per = 2 * BL // world
    for idx, labels in batches:
        sl = slice(rank * per, (rank + 1) * per)
        net.train_step(torch.from_numpy(idx[sl]).to(dev), torch.from_numpy(labels[sl]).to(dev))
    logits = net.forward(torch.from_numpy(batches[0][0][rank * per:(rank + 1) * per]).to(dev)).cpu()
    emb, l = net.tables.gather_full()
    if rank == 0:
        torch.save({"emb": emb, "lin": l, "dense": {k_: p.detach().cpu() for k_, p in net.P.params.items()}},
                   os.path.join(out_dir, f"nobn_w{world}.pt"))
    torch.save(logits, os.path.join(out_dir, f"nobn_w{world}_r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_hip_nobn(dev):
    out = tempfile.mkdtemp()
    for world in (1, 2):
        mp.spawn(run_rank_nobn, args=(world, free_port(), out), nprocs=world, join=True)
    a = torch.load(os.path.join(out, "nobn_w1.pt"))
    b = torch.load(os.path.join(out, "nobn_w2.pt"))
    torch.testing.assert_close(a["emb"], b["emb"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(a["lin"], b["lin"], rtol=1e-4, atol=2e-6)
    for k_ in a["dense"]:
        torch.testing.assert_close(a["dense"][k_], b["dense"][k_], rtol=1e-3, atol=5e-6)
    l1 = torch.load(os.path.join(out, "nobn_w1_r0.pt"))
    l2 = torch.cat([torch.load(os.path.join(out, f"nobn_w2_r{r}.pt")) for r in range(2)])
    torch.testing.assert_close(l1, l2, rtol=1e-3, atol=1e-5)

    # and the unsharded fused path (lr_fm_embed_bwd_adam_f32) gives the same tables
    from librecommender_amd.nets import DeepFMNet
    full, lin, batches = make_data()
    net = DeepFMNet(NU, NI, V - NU - NI - 2, FS, embed_size=K, hidden_units=HID, use_bn=False, lr=1e-2,
                    seed=42, device=dev)
    net.tables.embed.copy_(torch.from_numpy(full))
    net.tables.lin.copy_(torch.from_numpy(lin))
    for idx, labels in batches:
        net.train_step(torch.from_numpy(idx).to(dev), torch.from_numpy(labels).to(dev))
    torch.testing.assert_close(net.tables.embed.cpu(), a["emb"], rtol=1e-4, atol=2e-6)


def run_rank_lightgcn(rank, world, port, out_dir, fuse=False, chunks=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets.graph_nets import ShardedLightGCNNet
    from tests.golden_util import unflatten

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lightgcn.npz"))
    nu, ni, L = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    net = ShardedLightGCNNet(nu, ni, 16, L, unflatten(g["user_consumed_flat"]), dev, seed=42, lr=1e-2, epsilon=1e-8, chunks=chunks)
    assert net.chunks == (chunks if chunks is not None else (1 if world == 1 else 4))
    net.fuse_adam = fuse          # True: the optimiser step is the last backward product's epilogue (no gradient table to look at)
    ue, ie = net.embeddings()
    B = len(g["users"])
    sl = slice(rank * B // world, (rank + 1) * B // world)
    loss, G = net.train_step("bpr", g["users"][sl], g["pos"][sl], items_neg=g["neg"][sl])
    losses = [None] * world
    dist.all_gather_object(losses, float(loss))
    assert (G is None) == (fuse and L >= 2)
    G_full = None if G is None else net._all_gather_rows(G)[: nu + ni].cpu()
    E_full = net._all_gather_rows(net.E)[: nu + ni].cpu()
    if rank == 0:
        torch.save({"ue": ue.cpu(), "ie": ie.cpu(), "loss": float(np.mean(losses)), "G": G_full, "E": E_full},
                   os.path.join(out_dir, f"lgcn_w{world}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("world,chunks", [(1, None), (2, None), (1, 3), (2, 1)])
def test_sharded_lightgcn_hip_matches_reference_fixture(dev, world, chunks, fuse):
    """Row-partitioned LightGCN with the HIP SpMM / scatter / Adam kernels against the fixture
    generated by the reference module (propagation, BPR loss, gradients, one torch-Adam step).  The last forward product
    computes the rows the peers asked for only, the first backward product takes the owners' compact gradient lists instead of
    an all-gathered table; `fuse`: the optimiser step as the last backward product's epilogue.  `chunks` (default 4 under two
    ranks): the layer inputs travel in pieces, the slice is multiplied column block by column block (`acc += A_c X_c`)."""
    out = tempfile.mkdtemp()
    mp.spawn(run_rank_lightgcn, args=(world, free_port(), out, fuse, chunks), nprocs=world, join=True)
    r = torch.load(os.path.join(out, f"lgcn_w{world}.pt"))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lightgcn.npz"))
    nu = int(g["n_users"])
    np.testing.assert_allclose(r["ue"].numpy(), g["user_embeds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["ie"].numpy(), g["item_embeds"], rtol=1e-5, atol=1e-6)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    if r["G"] is not None:
        np.testing.assert_allclose(r["G"][:nu].numpy(), g["gU"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(r["G"][nu:].numpy(), g["gI"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(r["E"][:nu].numpy(), g["U1"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["E"][nu:].numpy(), g["I1"], rtol=1e-4, atol=2e-6)


# ---- row-sharded DeepFM on the FUSED kernels (lookup fused with the first Dense layer on the row cache) --------
NU2, NI2, VOC2, FS2, K2, B2 = 300, 200, 37, 9, 64, 256
HID2 = (128, 64, 32)
V2 = NU2 + 1 + NI2 + 1 + FS2 * (VOC2 + 1)
FRS2 = np.concatenate([[0, NU2 + 1, NU2 + 1 + NI2 + 1], NU2 + NI2 + 2 + (np.arange(FS2) + 1) * (VOC2 + 1)]).astype(np.int64)


def make_data2(seed=3, steps=3):
    rng = np.random.default_rng(seed)
    full = (rng.standard_normal((V2, K2)) * 0.05).astype(np.float32)
    lin = (rng.standard_normal((V2, 1)) * 0.05).astype(np.float32)
    batches = []
    for _ in range(steps):
        cols = [rng.integers(0, NU2 + 1, B2), NU2 + 1 + rng.integers(0, NI2 + 1, B2)]
        cols += [FRS2[2 + j] + rng.integers(0, VOC2 + 1, B2) for j in range(FS2)]
        batches.append((np.stack(cols, axis=1).astype(np.int32), rng.integers(0, 2, B2).astype(np.float32)))
    return full, lin, batches


def run_rank_fused(rank, world, port, out_dir, use_bn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDeepFMNet

    full, lin, batches = make_data2()
    net = ShardedDeepFMNet(V2, FS2, embed_size=K2, hidden_units=HID2, use_bn=use_bn, lr=1e-2, device=dev, seed=42,
                           field_row_start=FRS2)
    assert net.field_row_start is not None, "the fused row-sharded step is not active"
    net.tables.load_full(torch.from_numpy(full), torch.from_numpy(lin))
    per = B2 // world
    losses = []
    for s, (idx, labels) in enumerate(batches):
        sl = slice(rank * per, (rank + 1) * per)
        nxt = torch.from_numpy(batches[s + 1][0][sl]).to(dev) if s + 1 < len(batches) else None
        cur = torch.from_numpy(idx[sl]).to(dev) if s == 0 else cur_next
        losses.append(float(net.train_step(cur, torch.from_numpy(labels[sl]).to(dev), next_idx=nxt)))
        cur_next = nxt
    emb, l = net.tables.gather_full()
    if rank == 0:
        torch.save({"emb": emb, "lin": l, "losses": losses,
                    "dense": {k_: p.detach().cpu() for k_, p in net.P.params.items()}},
                   os.path.join(out_dir, f"fused_bn{int(use_bn)}_w{world}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_bn", [False, True])
def test_fused_row_sharded_step_matches_unsharded_fused_step(dev, use_bn):
    """`ShardedDeepFMNet` with `field_row_start` runs the fused kernels on the step's row cache
    (`lr_fm_field_stats_slots_f32`, `lr_deepfm_l1_*` over cache slots, `lr_fm_rows_grad_f32`, owner-side
    scatter-Adam).  One rank == the unsharded fused `DeepFMNet` step; two ranks == one rank on the concatenated
    batch when BatchNorm (per-replica statistics) is off."""
    out = tempfile.mkdtemp()
    worlds = (1, 2) if not use_bn else (1,)
    for world in worlds:
        mp.spawn(run_rank_fused, args=(world, free_port(), out, use_bn), nprocs=world, join=True)
    a = torch.load(os.path.join(out, f"fused_bn{int(use_bn)}_w1.pt"))
    from librecommender_amd.nets import DeepFMNet

    full, lin, batches = make_data2()
    net = DeepFMNet(NU2, NI2, FS2 * (VOC2 + 1), FS2, embed_size=K2, hidden_units=HID2, use_bn=use_bn, lr=1e-2, seed=42,
                    device=dev, sparse_offsets=np.arange(FS2) * (VOC2 + 1))
    assert net.fused_l1 and net.hip_tail
    net.tables.embed.copy_(torch.from_numpy(full))
    net.tables.lin.copy_(torch.from_numpy(lin))
    ref_losses = [float(net.train_step(torch.from_numpy(idx).to(dev), torch.from_numpy(labels).to(dev)))
                  for idx, labels in batches]
    np.testing.assert_allclose(a["losses"], ref_losses, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a["emb"], net.tables.embed.cpu(), rtol=1e-4, atol=5e-6)
    torch.testing.assert_close(a["lin"], net.tables.lin.cpu(), rtol=1e-4, atol=5e-6)
    for k_, p in net.P.params.items():
        torch.testing.assert_close(a["dense"][k_], p.detach().cpu(), rtol=1e-3, atol=1e-5)
    if not use_bn:
        b = torch.load(os.path.join(out, f"fused_bn{int(use_bn)}_w2.pt"))
        torch.testing.assert_close(a["emb"], b["emb"], rtol=1e-4, atol=5e-6)
        torch.testing.assert_close(a["lin"], b["lin"], rtol=1e-4, atol=5e-6)
        for k_ in a["dense"]:
            torch.testing.assert_close(a["dense"][k_], b["dense"][k_], rtol=1e-3, atol=1e-5)


def _din_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDINNet
    from librecommender_amd.parallel import HipKernels
    from tests import test_sharded_din_cpu as D

    full, batches = D.make_data()
    net = ShardedDINNet(D.V, 16, D.HID, use_bn=False, max_seq_len=D.L, lr=1e-2, device=dev, seed=42)
    assert isinstance(net.kern, HipKernels)
    full16 = np.concatenate([full, full[:, ::-1]], axis=1).copy()          # K = 16: a width the attention kernels take
    net.tables.load_full(torch.from_numpy(full16))
    per = 2 * D.BL // world
    sl = slice(rank * per, (rank + 1) * per)
    for u, i, s, n, y in batches:
        net.train_step(D.global_rows(u[sl], i[sl], s[sl]).to(dev), torch.from_numpy(n[sl]).to(dev), torch.from_numpy(y[sl]).to(dev))
    emb, _ = net.tables.gather_full()
    if rank == 0:
        torch.save({"emb": emb.cpu(), "dense": net.P.flat.detach().cpu().clone(), "full16": torch.from_numpy(full16),
                    "params": {k: p.detach().cpu().clone() for k, p in net.P.params.items()}}, os.path.join(out_dir, f"din_w{world}.pt"))
    dist.destroy_process_group()


def test_sharded_din_hip_two_ranks_equal_one_rank(dev):
    """`ShardedDINNet` with the HIP kernels (dense-form MFMA attention on the fetched rows): 2 ranks sharing the GPU
    == 1 rank on the concatenated batch (the oracle comparison of the same net runs on CPU, tests/test_sharded_din_cpu.py)."""
    out = tempfile.mkdtemp()
    for world in (1, 2):
        mp.spawn(_din_rank, args=(world, free_port(), out), nprocs=world, join=True)
    a, b = torch.load(os.path.join(out, "din_w1.pt")), torch.load(os.path.join(out, "din_w2.pt"))
    torch.testing.assert_close(a["emb"], b["emb"], rtol=1e-4, atol=5e-6)
    torch.testing.assert_close(a["dense"], b["dense"], rtol=1e-3, atol=5e-6)
    assert (a["emb"] != a["full16"]).any()


def _prefetch_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDeepFMNet

    rng = np.random.default_rng(3)
    Fs, voc, nu, ni, K_, B = 10, 3000, 5000, 4000, 64, 8192
    frs = np.concatenate([[0, nu + 1, nu + 1 + ni + 1], nu + 1 + ni + 1 + (np.arange(Fs) + 1) * (voc + 1)]).astype(np.int64)
    Vt = int(frs[-1])
    batches = []
    for _ in range(6):
        cols = [rng.zipf(1.2, B) % (nu + 1), nu + 1 + rng.zipf(1.2, B) % (ni + 1)]
        cols += [frs[2 + f] + rng.zipf(1.3, B) % (voc + 1) for f in range(Fs)]
        batches.append((torch.from_numpy(np.stack(cols, 1).astype(np.int32)).to(dev),
                        torch.from_numpy(rng.integers(0, 2, B).astype(np.float32)).to(dev)))
    outs = {}
    for mode in ("prefetch_async", "plain_synced"):
        net = ShardedDeepFMNet(Vt, Fs, embed_size=K_, hidden_units=(128, 64, 32), use_bn=True, lr=1e-3, device=dev, seed=42,
                               field_row_start=frs)
        assert net.field_row_start is not None
        losses = []
        for s in range(40):
            idx, lab = batches[s % len(batches)]
            if mode == "prefetch_async":    # exchange plans built one step ahead on the side stream, NO host sync in the loop
                losses.append(net.train_step(idx, lab, next_idx=batches[(s + 1) % len(batches)][0]))
            else:
                losses.append(net.train_step(idx, lab))
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        outs[mode] = (net.tables.embed.clone(), net.tables.m.clone(), net.tables.lin.clone(), net.P.flat.detach().clone(),
                      torch.stack([l.reshape(()) for l in losses]))
    for a, b in zip(outs["prefetch_async"], outs["plain_synced"]):
        assert torch.equal(a, b)
    torch.save({"ok": True}, os.path.join(out_dir, "prefetch.pt"))
    dist.destroy_process_group()


def test_prefetched_plans_without_host_sync_equal_synced_steps(dev):
    """`ShardedFieldTables.prefetch` builds the NEXT plan on a side stream into one of two alternating
    segment workspaces; the side stream now waits for the end-of-step event of the step that last used that workspace.
    40 steps with prefetch and no host synchronisation must equal 40 synchronised steps bit for bit."""
    out = tempfile.mkdtemp()
    mp.spawn(_prefetch_rank, args=(1, free_port(), out), nprocs=1, join=True)
    assert torch.load(os.path.join(out, "prefetch.pt"))["ok"]


def _bad_ids_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from librecommender_amd.nets import ShardedDeepFMNet

    Fs, voc, nu, ni, B = 4, 100, 50, 40, 256
    frs = np.concatenate([[0, nu + 1, nu + 1 + ni + 1], nu + 1 + ni + 1 + (np.arange(Fs) + 1) * (voc + 1)]).astype(np.int64)
    net = ShardedDeepFMNet(int(frs[-1]), Fs, embed_size=64, hidden_units=(128, 64, 32), device=dev, seed=1, field_row_start=frs)
    rng = np.random.default_rng(0)
    cols = [rng.integers(0, nu + 1, B), nu + 1 + rng.integers(0, ni + 1, B)] + [frs[2 + f] + rng.integers(0, voc + 1, B) for f in range(Fs)]
    good = torch.from_numpy(np.stack(cols, 1).astype(np.int32)).to(dev)
    lab = torch.from_numpy(rng.integers(0, 2, B).astype(np.float32)).to(dev)
    net.train_step(good, lab)
    bad = good.clone()
    bad[7, 3] = int(frs[5])                 # a row of the LAST sparse field in the column of the second one
    try:
        net.train_step(bad, lab)
        verdict = "no error"
    except ValueError as e:
        verdict = str(e)
    torch.save({"verdict": verdict}, os.path.join(out_dir, f"bad_w{world}_r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_field_plans_refuse_ids_outside_their_columns_field(dev, world):
    """Exchange plans off the field-wise sort drop an id that lies outside its column's row range (it would have no cache
    row): the plan counts the positions it kept and `resolve()` raises on the host instead of letting a kernel read slot -1."""
    out = tempfile.mkdtemp()
    mp.spawn(_bad_ids_rank, args=(world, free_port(), out), nprocs=world, join=True)
    for r in range(world):
        v = torch.load(os.path.join(out, f"bad_w{world}_r{r}.pt"))["verdict"]
        assert "outside the row range" in v, v
