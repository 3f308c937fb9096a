"""Multi-process path on CPU: world_size-2 gloo, oracle kernels injected.

Checks that (i) the row-sharded DeepFM step over 2 ranks reproduces a 1-rank run of the same
semantics on the concatenated batch, weights and all, over several steps; (ii) its first step
matches the reference-graph oracle (TF1 Adam: identical on step 1 from zero moments);
(iii) item-sharded top-k + merge equals the unsharded ranking."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.models_torch import DeepFMOracle
from tests.oracle_kernels import OracleKernels

NU, NI, VOC, FS, K, BL, STEPS = 30, 40, 7, 5, 16, 24, 3
HID = (16, 8)
V = NU + 1 + NI + 1 + FS * (VOC + 1)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def make_data(seed=0):
    rng = np.random.default_rng(seed)
    full = (rng.standard_normal((V, K)) * 0.1).astype(np.float32)
    lin = (rng.standard_normal((V, 1)) * 0.1).astype(np.float32)
    batches = []
    for _ in range(STEPS):
        users = rng.integers(0, NU, 2 * BL)
        items = rng.integers(0, NI, 2 * BL) + NU + 1
        sp = rng.integers(0, VOC, (2 * BL, FS)) + np.arange(FS) * (VOC + 1) + NU + 1 + NI + 1
        idx = np.concatenate([users[:, None], items[:, None], sp], axis=1).astype(np.int32)
        labels = rng.integers(0, 2, 2 * BL).astype(np.float32)
        batches.append((idx, labels))
    return full, lin, batches


def run_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    from librecommender_amd.nets import ShardedDeepFMNet

    full, lin, batches = make_data()
    net = ShardedDeepFMNet(V, FS, embed_size=K, hidden_units=HID, use_bn=False, lr=1e-2,
                           device=torch.device("cpu"), kern=OracleKernels(), seed=42)
    net.tables.load_full(torch.from_numpy(full), torch.from_numpy(lin))
    per = 2 * BL // world
    losses = []
    sl = slice(rank * per, (rank + 1) * per)
    tens = [(torch.from_numpy(idx[sl]), torch.from_numpy(labels[sl])) for idx, labels in batches]
    for j, (ti, tl) in enumerate(tens):        # the next batch's exchange plan is prefetched (as bench.py does)
        losses.append(float(net.train_step(ti, tl, next_idx=tens[j + 1][0] if j + 1 < len(tens) else None)))
    logits = net.forward(torch.from_numpy(batches[0][0][rank * per:(rank + 1) * per]))
    emb, l = net.tables.gather_full()
    # item-sharded scoring
    from librecommender_amd.parallel import sharded_score_topk
    rng = np.random.default_rng(5)
    U = torch.from_numpy(rng.standard_normal((6, 8)).astype(np.float32))
    I = torch.from_numpy(rng.standard_normal((101, 8)).astype(np.float32))
    bounds = np.linspace(0, 101, world + 1).astype(int)
    s, i = sharded_score_topk(OracleKernels(), U, I[bounds[rank]:bounds[rank + 1]].contiguous(), 9, int(bounds[rank]))
    if rank == 0:
        dense = {k_: p.detach().clone() for k_, p in net.P.params.items()}
        torch.save({"emb": emb, "lin": l, "dense": dense, "losses": losses, "topk_s": s, "topk_i": i,
                    "U": U, "I": I}, os.path.join(out_dir, f"w{world}.pt"))
    torch.save({"logits": logits}, os.path.join(out_dir, f"w{world}_r{rank}.pt"))
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs():
    out = tempfile.mkdtemp()
    for world in (1, 2):
        mp.spawn(run_rank, args=(world, free_port(), out), nprocs=world, join=True)
    return out


def test_two_ranks_equal_one_rank(runs):
    a = torch.load(os.path.join(runs, "w1.pt"))
    b = torch.load(os.path.join(runs, "w2.pt"))
    # global-batch loss = mean of the two local means
    torch.testing.assert_close(a["emb"], b["emb"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a["lin"], b["lin"], rtol=1e-5, atol=1e-6)
    for k_ in a["dense"]:
        torch.testing.assert_close(a["dense"][k_], b["dense"][k_], rtol=1e-4, atol=1e-6)
    l1 = torch.load(os.path.join(runs, "w1_r0.pt"))["logits"]
    l2 = torch.cat([torch.load(os.path.join(runs, f"w2_r{r}.pt"))["logits"] for r in range(2)])
    torch.testing.assert_close(l1, l2, rtol=1e-4, atol=1e-5)


def test_eight_ranks_equal_one_rank(runs):
    """The driver's largest scaling point: 8 ranks (6 samples per rank and step, uneven row shards, peers that own none
    of a rank's rows) train the same tables, dense parameters and logits as one rank; item-sharded top-k identical."""
    mp.spawn(run_rank, args=(8, free_port(), runs), nprocs=8, join=True)
    a = torch.load(os.path.join(runs, "w1.pt"))
    b = torch.load(os.path.join(runs, "w8.pt"))
    torch.testing.assert_close(a["emb"], b["emb"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a["lin"], b["lin"], rtol=1e-5, atol=1e-6)
    for k_ in a["dense"]:
        torch.testing.assert_close(a["dense"][k_], b["dense"][k_], rtol=1e-4, atol=1e-6)
    l1 = torch.load(os.path.join(runs, "w1_r0.pt"))["logits"]
    l8 = torch.cat([torch.load(os.path.join(runs, f"w8_r{r}.pt"))["logits"] for r in range(8)])
    torch.testing.assert_close(l1, l8, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a["topk_s"], b["topk_s"])
    assert torch.equal(a["topk_i"], b["topk_i"])


def test_first_step_matches_reference_graph_oracle(runs):
    """From zero moments one lazy-Adam step == one TF1 dense-Adam step (untouched rows get 0)."""
    full, lin, batches = make_data()
    u_rows, i_rows = NU + 1, NI + 1
    b = torch.load(os.path.join(runs, "w2.pt"))
    # rebuild the dense weights the sharded run started from: same seed -> same DenseParams init
    from librecommender_amd.layers import DenseParams, DenseStack, TFDense
    P = DenseParams(torch.device("cpu"), 42)
    TFDense(P, "linear", FS + 2, 1); DenseStack(P, "mlp", (FS + 2) * K, HID, False, 0.0); TFDense(P, "out", 1 + K + HID[-1], 1)
    P.finalize()
    W = {"user_embeds_var": torch.from_numpy(full[:u_rows]), "item_embeds_var": torch.from_numpy(full[u_rows:u_rows + i_rows]),
         "sparse_embeds_var": torch.from_numpy(full[u_rows + i_rows:]), "user_linear_var": torch.from_numpy(lin[:u_rows]),
         "item_linear_var": torch.from_numpy(lin[u_rows:u_rows + i_rows]), "sparse_linear_var": torch.from_numpy(lin[u_rows + i_rows:, 0])}
    W.update({k_: p.detach().clone() for k_, p in P.params.items()})
    o = DeepFMOracle(W, HID, use_bn=False, lr=1e-2, dtype=torch.float64)
    idx, labels = batches[0]
    li = torch.from_numpy(idx).long()
    loss = o.train_step(li[:, 0], li[:, 1] - u_rows, li[:, 2:] - u_rows - i_rows, torch.from_numpy(labels))
    # run ONE sharded step in-process (world 1) to compare weights after exactly one step
    port = free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from librecommender_amd.nets import ShardedDeepFMNet
        net = ShardedDeepFMNet(V, FS, embed_size=K, hidden_units=HID, use_bn=False, lr=1e-2,
                               device=torch.device("cpu"), kern=OracleKernels(), seed=42)
        net.tables.load_full(torch.from_numpy(full), torch.from_numpy(lin))
        l_sh = float(net.train_step(torch.from_numpy(idx), torch.from_numpy(labels)))
        assert abs(l_sh - float(loss)) < 1e-5
        emb = net.tables.embed
        ref = torch.cat([o.V.v["user_embeds_var"], o.V.v["item_embeds_var"], o.V.v["sparse_embeds_var"]]).detach()
        torch.testing.assert_close(emb.double(), ref, rtol=1e-4, atol=2e-6)
        torch.testing.assert_close(net.P["out/kernel"].detach().double(), o.V.v["out/kernel"].detach(), rtol=1e-4, atol=2e-6)
    finally:
        dist.destroy_process_group()


def test_item_sharded_topk_equals_unsharded(runs):
    b = torch.load(os.path.join(runs, "w2.pt"))
    a = torch.load(os.path.join(runs, "w1.pt"))
    assert torch.equal(a["topk_i"], b["topk_i"])
    torch.testing.assert_close(a["topk_s"], b["topk_s"])
    from oracle import ops_np
    ids, _ = ops_np.recommend_from_embedding(b["U"].numpy(), b["I"].numpy(), list(range(6)), 9, 101, {}, False)
    np.testing.assert_array_equal(b["topk_i"].numpy(), ids)


def run_rank_edge(rank, world, port, out_dir):
    """World 3, V not divisible by 3, and batches whose rows all live on ONE owner (rank 1): the
    other owners receive empty requests / gradient lists, rank 1 sees every duplicate."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.parallel import ShardedFieldTables

    Vv, Kk = 50, 8                                            # 50 = 17 + 17 + 16 rows
    rng = np.random.default_rng(3)
    full = torch.from_numpy(rng.standard_normal((Vv, Kk)).astype(np.float32))
    lin = torch.from_numpy(rng.standard_normal((Vv, 1)).astype(np.float32))
    kern = OracleKernels()
    tabs = ShardedFieldTables(Vv, Kk, torch.device("cpu"), kern)
    tabs.load_full(full, lin)
    assert tabs.embed.shape[0] == [17, 17, 16][rank]
    ids = torch.tensor([[1, 4, 7], [4, 4, 49], [1, 46, 7]], dtype=torch.int32)      # all = 1 (mod 3)
    if rank == 2:
        ids = torch.tensor([[4, 4, 4], [4, 4, 4], [4, 4, 4]], dtype=torch.int32)   # one hot row
    ctx = tabs.lookup(ids)
    assert ctx.send_counts[0] == 0 and ctx.send_counts[2] == 0
    torch.testing.assert_close(ctx.cache[ctx.slots.long()], full[ids.long()])          # rows arrive, in run order
    torch.testing.assert_close(ctx.lin_cache[ctx.slots.long()], lin[ids.long()])
    if rank != 1:
        assert ctx.recv_ids.numel() == 0
    g = torch.ones((ctx.n_rows, Kk)) * (rank + 1)
    gl = torch.ones(ctx.n_rows) * (rank + 1)
    tabs.apply_gradients(ctx, g, gl, kern.adam_hp(1e-2, 1, 1e-5))
    emb, _ = tabs.gather_full()
    if rank == 0:
        torch.save({"emb": emb, "full": full}, os.path.join(out_dir, "edge.pt"))
    dist.destroy_process_group()


def test_sharded_tables_uneven_world3_empty_peers_and_hot_row():
    out = tempfile.mkdtemp()
    mp.spawn(run_rank_edge, args=(3, free_port(), out), nprocs=3, join=True)
    r = torch.load(os.path.join(out, "edge.pt"))
    moved = (r["emb"] != r["full"]).any(dim=1).nonzero().flatten().tolist()
    assert moved == [1, 4, 7, 46, 49]                       # exactly the requested rows, wherever they live
    # first Adam step from zero moments moves every touched element by ~lr against the gradient sign
    torch.testing.assert_close(r["emb"][moved], r["full"][moved] - 1e-2, rtol=0, atol=2e-4)


def _rank_save_load(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.parallel import ShardedFieldTables
    from tests.oracle_kernels import OracleKernels

    t = ShardedFieldTables(V, K, torch.device("cpu"), OracleKernels(), seed=7)
    full, lin, _ = make_data()
    if mode == "save":
        t.load_full(torch.from_numpy(full), torch.from_numpy(lin))
        t.m.copy_(t.embed * 2)
        t.v.copy_(t.embed * 3)
        t.lin_m.copy_(t.lin * 4)
        t.lin_v.copy_(t.lin * 5)
        t.save_shard(out_dir)
    else:
        if mode == "same":
            t.load_shard(out_dir)
        else:
            t.load_shards_resharded(out_dir)
        e, l = t.gather_full()
        assert torch.equal(e, torch.from_numpy(full)) and torch.equal(l, torch.from_numpy(lin))
        rows = torch.arange(rank, V, world)
        assert torch.equal(t.m.cpu(), torch.from_numpy(full)[rows] * 2) and torch.equal(t.v.cpu(), torch.from_numpy(full)[rows] * 3)
        assert torch.equal(t.lin_m.cpu(), torch.from_numpy(lin)[rows] * 4) and torch.equal(t.lin_v.cpu(), torch.from_numpy(lin)[rows] * 5)
    dist.destroy_process_group()


def test_sharded_tables_save_load_per_shard_and_resharded():
    """Row-sharded tables are checkpointed per rank (`utils/save_load.py:70-115` semantics without a gather);
    the files restore the tables and the Adam moments on the same world size (2 -> 2) and re-distributed onto
    another one (2 -> 3, 2 -> 1)."""
    import tempfile

    out = tempfile.mkdtemp()
    mp.spawn(_rank_save_load, args=(2, free_port(), out, "save"), nprocs=2, join=True)
    mp.spawn(_rank_save_load, args=(2, free_port(), out, "same"), nprocs=2, join=True)
    mp.spawn(_rank_save_load, args=(3, free_port(), out, "reshard"), nprocs=3, join=True)
    mp.spawn(_rank_save_load, args=(1, free_port(), out, "reshard"), nprocs=1, join=True)
