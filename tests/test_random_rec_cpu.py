"""`random_rec=True` without the [B, N] score matrix (`recommendation/ranking.py:65-73`): the streaming
Gumbel-top-k draw follows the same distribution as the materialised `torch.multinomial` draw (both: n_rec items
without replacement, weights softmax(score)^0.75 + 1e-8, consumed items excluded), is chunk-size independent in
law, returns distinct un-consumed ids ordered by score."""
import numpy as np
import torch

from librecommender_amd.recommendation.recommend import random_select_device, random_select_streaming


def test_streaming_random_rec_matches_the_materialised_draw_in_distribution():
    torch.manual_seed(0)
    N, D, n_rec, B = 23, 4, 3, 40_000
    I = torch.randn(N, D)
    u = torch.randn(1, D) * 1.5
    U = u.repeat(B, 1)
    consumed = torch.tensor([2, 7], dtype=torch.int32)
    ptr = torch.arange(B + 1, dtype=torch.int64) * 2
    cidx = consumed.repeat(B)
    flag = torch.ones(B, dtype=torch.uint8)
    got = random_select_streaming(U, I, ptr, cidx, flag, n_rec, max_elems=B * 7)      # 4 chunks of 7 items
    scores = U @ I.T
    banned = torch.zeros_like(scores, dtype=torch.bool)
    banned[:, consumed.long()] = True
    ref = random_select_device(scores, banned, n_rec)
    for picks in (got, ref):
        assert picks.shape == (B, n_rec)
        assert not bool(((picks == 2) | (picks == 7)).any())
        srt = torch.sort(picks, dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())                               # distinct
        sc = torch.gather(scores, 1, picks)
        assert bool((sc[:, :-1] >= sc[:, 1:]).all())                                 # ordered by score
    f_got = np.bincount(got.reshape(-1).numpy(), minlength=N) / B
    f_ref = np.bincount(ref.reshape(-1).numpy(), minlength=N) / B
    np.testing.assert_allclose(f_got, f_ref, atol=0.012)                             # inclusion probabilities
    # the FIRST pick in draw order is not observable (results are re-ordered by score); the joint law of pairs is:
    pair = lambda p: np.bincount((torch.sort(p, 1).values[:, 0] * N + torch.sort(p, 1).values[:, 1]).numpy(), minlength=N * N) / B  # noqa: E731
    np.testing.assert_allclose(pair(got), pair(ref), atol=0.012)


def test_streaming_random_rec_without_consumed_lists():
    torch.manual_seed(1)
    U, I = torch.randn(5, 8), torch.randn(100, 8)
    ptr = torch.zeros(6, dtype=torch.int64)
    picks = random_select_streaming(U, I, ptr, torch.zeros(1, dtype=torch.int32), torch.zeros(5, dtype=torch.uint8), 10, max_elems=5 * 16)
    assert picks.shape == (5, 10) and all(len(set(r.tolist())) == 10 for r in picks)


def _run_sharded_random(rank, world, port, out_dir):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from librecommender_amd.distributed import ShardedItemEmbeds
    from tests.oracle_kernels import OracleKernels

    torch.manual_seed(0)                                   # the same seed on every rank: streams must still be independent
    N, D, n_rec, B = 23, 4, 3, 30_000
    I = torch.randn(N, D)
    U = (torch.randn(1, D) * 1.5).repeat(B, 1)
    per = -(-N // world)
    base = rank * per
    n_local = max(0, min(per, N - base))
    local = torch.zeros((per, D))
    local[:n_local] = I[base: base + n_local]
    emb = ShardedItemEmbeds(local, N, base, n_local, kern=OracleKernels())
    consumed = torch.tensor([2, 7], dtype=torch.int32)
    ptr = torch.arange(B + 1, dtype=torch.int64) * 2
    picks = emb.random_topk(U, n_rec, ptr, consumed.repeat(B), torch.ones(B, dtype=torch.uint8))
    torch.save({"picks": picks, "U": U, "I": I}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_random_rec_matches_the_materialised_draw_in_distribution():
    """`random_rec` on item embeddings sharded over 3 ranks (`ShardedItemEmbeds.random_topk`, round 4): same law as the
    materialised draw on the gathered matrix, identical lists on every rank."""
    import os
    import tempfile

    import torch.multiprocessing as mp

    from tests.test_dist_api_cpu import free_port

    out = tempfile.mkdtemp()
    mp.spawn(_run_sharded_random, args=(3, free_port(), out), nprocs=3, join=True)
    r = [torch.load(os.path.join(out, f"r{k}.pt"), weights_only=False) for k in range(3)]
    got = r[0]["picks"]
    assert torch.equal(got, r[1]["picks"]) and torch.equal(got, r[2]["picks"])
    U, I = r[0]["U"], r[0]["I"]
    B, N = U.shape[0], I.shape[0]
    scores = U @ I.T
    banned = torch.zeros_like(scores, dtype=torch.bool)
    banned[:, [2, 7]] = True
    ref = random_select_device(scores, banned, 3)
    assert not bool(((got == 2) | (got == 7)).any())
    srt = torch.sort(got, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    sc = torch.gather(scores, 1, got)
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    f_got = np.bincount(got.reshape(-1).numpy(), minlength=N) / B
    f_ref = np.bincount(ref.reshape(-1).numpy(), minlength=N) / B
    np.testing.assert_allclose(f_got, f_ref, atol=0.014)
    pair = lambda p: np.bincount((torch.sort(p, 1).values[:, 0] * N + torch.sort(p, 1).values[:, 1]).numpy(), minlength=N * N) / B  # noqa: E731
    np.testing.assert_allclose(pair(got), pair(ref), atol=0.014)
